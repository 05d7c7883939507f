// Dependent v_mul/v_add chain (the iir_blur recurrence) on FEW wavefronts: does a lightly loaded chip clock lower?
// hipcc --offload-arch=gfx950 -O3 valu_lowload.hip -o valu_lowload
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(64) void k(float *out, int iters, float c1, float u) {
    float b = threadIdx.x * 0.001f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 64; r++) asm volatile("v_mul_f32 %0, %1, %0\n\tv_add_f32 %0, %2, %0" : "+v"(b) : "v"(c1), "v"(u));
    }
    if (b == 12345.0f) out[0] = b;
}
int main() {
    float *d;
    hipMalloc(&d, 4);
    for (int blocks : {1, 72, 256, 1024, 4096}) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0), hipEventCreate(&e1);
        k<<<blocks, 64>>>(d, 10, 0.9f, 0.1f);
        hipEventRecord(e0);
        k<<<blocks, 64>>>(d, 2000, 0.9f, 0.1f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("%5d waves: %.2f ns per (mul, add) step\n", blocks, ms * 1e6 / (2000.0 * 64));
    }
    return 0;
}
