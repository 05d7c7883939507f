// Issue rate of gfx950 packed-f32 VALU ops (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32) against their scalar forms at 1, 2, 4
// waves per SIMD and 1..8 independent chains: ns per wave-INSTRUCTION per SIMD (a packed instruction produces two results).
//   hipcc --offload-arch=gfx950 -O3 valu_pk.hip -o valu_pk
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));

template<int OP, int C>
__global__ __launch_bounds__(256) void k(float *out, int iters, float s) {
    f2 a[8];
    f2 s2 = {s, s};
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = f2{threadIdx.x * 0.001f + i, threadIdx.x * 0.002f + i};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 64 / C; r++) {
#pragma unroll
            for (int i = 0; i < C; i++) {
                if (OP == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i].x) : "v"(s));
                if (OP == 1) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(s2));
                if (OP == 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(s2));
                if (OP == 3) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(s2));
                if (OP == 4) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i].x) : "v"(s));
                if (OP == 5) asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel_hi:[1,0]" : "+v"(a[i]) : "v"(s2));   // broadcast the low half of src1
            }
        }
    }
    float acc = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) acc += a[i].x + a[i].y;
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
    return ms * 1e-3 / ((double)iters * 64 * w) * 1e9;
}

int main() {
    float *d;
    hipMalloc(&d, 4);
    const char *names[] = {"v_add_f32", "v_pk_add_f32", "v_pk_mul_f32", "v_pk_fma_f32", "v_fma_f32", "v_pk_mul bcast"};
    for (int w : {1, 2, 4}) {
        printf("waves/SIMD=%d  chains:      1      2      4      8   (ns per wave-instruction per SIMD)\n", w);
#define ROW(OP) printf("  %-16s %6.3f %6.3f %6.3f %6.3f\n", names[OP], run<OP, 1>(w, 3000, d), run<OP, 2>(w, 3000, d), run<OP, 4>(w, 3000, d), run<OP, 8>(w, 3000, d));
        ROW(0) ROW(1) ROW(2) ROW(3) ROW(4) ROW(5)
    }
    return 0;
}
