// LDS read throughput by instruction width (gfx950): ns per wave-instruction and bytes per clock per CU, conflict-free streaming
// addresses, 16 reads in flight per wave, 4 / 8 / 16 waves per CU.   hipcc --offload-arch=gfx950 -O3 lds_read_width.hip -o lds_read_width
//   0 ds_read_b32   lane l -> dword l                 1 ds_read2_b32  lane l -> dwords l, l + 64
//   2 ds_read_b64   lane l -> 8 bytes at 8 l          3 ds_read2_b64  lane l -> 8 bytes at 8 l and 8 l + 512
//   4 ds_read_b128  lane l -> 16 bytes at 16 l
#include <hip/hip_runtime.h>
#include <stdio.h>

template<int OP>
__global__ __launch_bounds__(1024) void k(unsigned *out, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned s[16384];
    const int t = threadIdx.x, l = t & 63;
    for (int i = t; i < 16384; i += blockDim.x) s[i] = i * 2654435761u;
    __syncthreads();
    const unsigned wb = ((t >> 6) & 3) * 4096;   // byte base of the wave's window
    unsigned a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
            if (OP == 0) { unsigned v; asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(wb + 4 * l), "n"(u * 256)); a0 ^= v; }
            if (OP == 1) { unsigned long long v; asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(v) : "v"(wb + 4 * l), "n"(u * 2), "n"(u * 2 + 64)); a0 ^= (unsigned)v, a1 ^= (unsigned)(v >> 32); }
            if (OP == 2) { unsigned long long v; asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(wb + 8 * l), "n"(u * 512)); a0 ^= (unsigned)v, a1 ^= (unsigned)(v >> 32); }
            if (OP == 3) { uint4 v; asm volatile("ds_read2_b64 %0, %1 offset0:%2 offset1:%3" : "=v"(v) : "v"(wb + 8 * l), "n"(u * 2), "n"(u * 2 + 64)); a0 ^= v.x, a1 ^= v.y, a2 ^= v.z, a3 ^= v.w; }
            if (OP == 4) { uint4 v; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(wb + 16 * l), "n"(u * 1024)); a0 ^= v.x, a1 ^= v.y, a2 ^= v.z, a3 ^= v.w; }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
    }
    if ((a0 ^ a1 ^ a2 ^ a3) == 0x12345678u) out[0] = a0;
}
template<int OP>
void run(unsigned *d, const char *name, int bytes) {
    for (int waves : {4, 8, 16}) {
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
        const int iters = 4000;
        k<OP><<<256, waves * 64>>>(d, 10);
        (void)hipEventRecord(e0);
        k<OP><<<256, waves * 64>>>(d, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double ns = ms * 1e6 / ((double)iters * 16 * waves);   // per wave-instruction of one CU
        printf("%-14s %2d waves per CU: %5.2f ns per wave-instruction, %6.1f B/ns/CU\n", name, waves, ns, 64.0 * bytes / ns);
    }
}
int main() {
    unsigned *d;
    (void)hipMalloc(&d, 4);
    run<0>(d, "ds_read_b32", 4), run<1>(d, "ds_read2_b32", 8), run<2>(d, "ds_read_b64", 8), run<3>(d, "ds_read2_b64", 16), run<4>(d, "ds_read_b128", 16);
    return 0;
}
