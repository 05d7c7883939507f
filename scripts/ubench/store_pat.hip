// Store-pattern microbenchmark: 25.7 MB written by 421 workgroups x 4 waves (the epilogue of conv3x3_bf16_lin), as
//   0: dword per lane, half-waves on rows 4 apart (the MFMA C/D layout as it falls out of the accumulators)
//   1: dword per lane, one 256-byte run per instruction (after v_permlane32_swap)
//   2: dwordx4 per lane, 1 KB = two whole pixel rows per instruction (after a transpose through LDS)
// hipcc --offload-arch=gfx950 -O3 store_pat.hip -o store_pat
#include <hip/hip_runtime.h>
#include <stdio.h>
template<int MODE>
__global__ __launch_bounds__(256) void k(float *out, int CO) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1;
    const long p0 = (long)blockIdx.x * 128;
    if (MODE == 0) {
        for (int a = 0; a < 2; a++)
            for (int r = 0; r < 16; r++) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const long p = p0 + 64 * wm + 32 * a + row;
                for (int b = 0; b < 2; b++) out[p * CO + 64 * wn + 32 * b + (lane & 31)] = (float)(r + b);
            }
    } else if (MODE == 1) {
        for (int a = 0; a < 2; a++)
            for (int r = 0; r < 32; r++) {
                const long p = p0 + 64 * wm + 32 * a + r;
                out[p * CO + 64 * wn + lane] = (float)r;
            }
    } else {
        for (int i = 0; i < 16; i++) {   // wave writes pixel rows 32 wave + 2 i, + 1: 128 floats each = 32 lanes x float4
            const long p = p0 + 32 * wave + 2 * i + (lane >> 5);
            *reinterpret_cast<float4 *>(out + p * CO + 4 * (lane & 31)) = make_float4(i, 1, 2, 3);
        }
    }
}
template<int MODE>
float run(float *d, int nwg) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    k<MODE><<<nwg, 256>>>(d, 128);
    hipEventRecord(e0);
    for (int i = 0; i < 20; i++) k<MODE><<<nwg, 256>>>(d, 128);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / 20 * 1e3f;
}
int main() {
    float *d;
    const int nwg = 421;
    hipMalloc(&d, (size_t)nwg * 128 * 128 * 4);
    for (int rep = 0; rep < 2; rep++)
        printf("dword split rows %.1f us   dword 256 B runs %.1f us   dwordx4 whole rows %.1f us   (%.1f MB)\n", run<0>(d, nwg), run<1>(d, nwg),
               run<2>(d, nwg), nwg * 65536 / 1e6);
    return 0;
}
