// VALU issue-rate microbenchmark for gfx950: cycles per wave64 instruction for f32 add/mul/fma, packed variants,
// DPP moves and LDS gathers, at 1/2/4 waves per SIMD.   hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));

template<int OP>
__global__ __launch_bounds__(256) void k(float *out, int iters, float s) {
    float a[8];
    v2f p[8];
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = threadIdx.x * 0.001f + i, p[i] = v2f{a[i], a[i] + 1.0f};
    v2f s2 = {s, s};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (OP == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(s));
                if (OP == 1) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(s));
                if (OP == 2) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(s));
                if (OP == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(s2));
                if (OP == 4) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(s2));
                if (OP == 5) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(s2));
                if (OP == 6) asm volatile("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
                if (OP == 7) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(a[i]));
                if (OP == 8) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(s));
            }
        }
    }
    float acc = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) acc += a[i] + p[i].x + p[i].y;
    if (acc == 12345.0f) out[0] = acc;
}

template<int OP>
double run(int waves_per_simd, int iters, float *d) {
    int blocks = 256 * waves_per_simd;  // 256 CUs, 4 waves per block = 1 per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    k<OP><<<blocks, 256>>>(d, 10, 1.0001f);
    hipEventRecord(e0);
    k<OP><<<blocks, 256>>>(d, iters, 1.0001f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double instr_per_wave = (double)iters * 64;
    return ms * 1e-3 / (instr_per_wave * waves_per_simd);  // seconds per wave-instruction per SIMD
}

int main() {
    float *d;
    hipMalloc(&d, 4);
    const char *names[] = {"v_add_f32", "v_mul_f32", "v_fma_f32", "v_pk_add_f32", "v_pk_mul_f32", "v_pk_fma_f32",
                           "v_mov_dpp wave_shr", "v_cvt_f32_u32", "v_cndmask"};
    for (int w : {1, 2, 4}) {
        double t[9] = {run<0>(w, 4000, d), run<1>(w, 4000, d), run<2>(w, 4000, d), run<3>(w, 4000, d), run<4>(w, 4000, d),
                       run<5>(w, 4000, d), run<6>(w, 4000, d), run<7>(w, 4000, d), run<8>(w, 4000, d)};
        for (int i = 0; i < 9; i++) printf("waves/SIMD=%d %-20s %.3f ns/instr  (%.2f cycles @2.4GHz)\n", w, names[i], t[i] * 1e9, t[i] * 2.4e9);
    }
    return 0;
}
