// Dependent-issue latency of gfx950 VALU ops: ns per wave-instruction for C independent chains of v_add_f32 / v_fma_f32 /
// add+DPP at 1, 2, 4 waves per SIMD.   hipcc --offload-arch=gfx950 -O3 valu_dep.hip -o valu_dep
#include <hip/hip_runtime.h>
#include <stdio.h>

template<int OP, int C>
__global__ __launch_bounds__(256) void k(float *out, int iters, float s) {
    float a[8];
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = threadIdx.x * 0.001f + i;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 64 / C; r++) {
#pragma unroll
            for (int i = 0; i < C; i++) {
                if (OP == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(s));
                if (OP == 1) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(s));
                if (OP == 2) asm volatile("v_add_f32_dpp %0, %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(s));
                if (OP == 3) asm volatile("v_add_f32 %0, %0, %1\n\tv_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(s));
                if (OP == 4) asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[i]) : "s"(s));
                if (OP == 5) asm volatile("v_mul_f32 %0, 0x40400000, %0" : "+v"(a[i]));
                if (OP == 6) asm volatile("v_mul_f32 %0, 0.5, %0" : "+v"(a[i]));
                if (OP == 7) asm volatile("v_add_f32 %0, %0, %1\n\ts_add_u32 s20, s20, 1" : "+v"(a[i]) : "v"(s) : "s20");
                if (OP == 8) asm volatile("v_add_f32 %0, %0, %1\n\ts_nop 0" : "+v"(a[i]) : "v"(s));
            }
        }
    }
    float acc = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) acc += a[i];
    if (acc == 12345.0f) out[0] = acc;
}

template<int OP, int C>
double run(int w, int iters, float *d) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    k<OP, C><<<256 * w, 256>>>(d, 10, 1.0001f);
    hipEventRecord(e0);
    k<OP, C><<<256 * w, 256>>>(d, iters, 1.0001f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e-3 / ((double)iters * 64 * (OP == 3 ? 2 : 1) * w) * 1e9;   // OP 7, 8: per (VALU + scalar) pair
}

int main() {
    float *d;
    hipMalloc(&d, 4);
    const char *names[] = {"v_add_f32", "v_fma_f32", "v_add_f32_dpp", "add,mul pairs", "v_add_f32 sgpr", "v_mul_f32 literal", "v_mul_f32 inline", "add + s_add pair", "add + s_nop pair"};
    for (int w : {1, 2, 4}) {
        printf("waves/SIMD=%d  chains:      1      2      4      8   (ns per wave-instruction per SIMD)\n", w);
        printf("  %-16s %6.3f %6.3f %6.3f %6.3f\n", names[0], run<0, 1>(w, 3000, d), run<0, 2>(w, 3000, d), run<0, 4>(w, 3000, d), run<0, 8>(w, 3000, d));
        printf("  %-16s %6.3f %6.3f %6.3f %6.3f\n", names[1], run<1, 1>(w, 3000, d), run<1, 2>(w, 3000, d), run<1, 4>(w, 3000, d), run<1, 8>(w, 3000, d));
        printf("  %-16s %6.3f %6.3f %6.3f %6.3f\n", names[2], run<2, 1>(w, 3000, d), run<2, 2>(w, 3000, d), run<2, 4>(w, 3000, d), run<2, 8>(w, 3000, d));
        printf("  %-16s %6.3f %6.3f %6.3f %6.3f\n", names[3], run<3, 1>(w, 3000, d), run<3, 2>(w, 3000, d), run<3, 4>(w, 3000, d), run<3, 8>(w, 3000, d));
        printf("  %-16s %6.3f %6.3f %6.3f %6.3f\n", names[4], run<4, 1>(w, 3000, d), run<4, 2>(w, 3000, d), run<4, 4>(w, 3000, d), run<4, 8>(w, 3000, d));
        printf("  %-16s %6.3f %6.3f %6.3f %6.3f\n", names[5], run<5, 1>(w, 3000, d), run<5, 2>(w, 3000, d), run<5, 4>(w, 3000, d), run<5, 8>(w, 3000, d));
        printf("  %-16s %6.3f %6.3f %6.3f %6.3f\n", names[6], run<6, 1>(w, 3000, d), run<6, 2>(w, 3000, d), run<6, 4>(w, 3000, d), run<6, 8>(w, 3000, d));
        printf("  %-16s %6.3f %6.3f %6.3f %6.3f\n", names[7], run<7, 1>(w, 3000, d), run<7, 2>(w, 3000, d), run<7, 4>(w, 3000, d), run<7, 8>(w, 3000, d));
        printf("  %-16s %6.3f %6.3f %6.3f %6.3f\n", names[8], run<8, 1>(w, 3000, d), run<8, 2>(w, 3000, d), run<8, 4>(w, 3000, d), run<8, 8>(w, 3000, d));
    }
    return 0;
}
