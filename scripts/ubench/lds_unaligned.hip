// Does gfx950 serve ds_read_b128 / ds_read_b64 at addresses that are only 4-byte aligned, and at what cost?  (The compiler splits a
// 4-byte-aligned 16-byte LDS load into four dword reads; local_laplacian's remap-table gathers want 8 consecutive words at a
// data-dependent word offset.)  Part 1: correctness at every word offset mod 4.  Part 2: ns per wave-instruction for RANDOM word
// addresses (the noise-frame pattern), b32 x 8 vs b128 x 2 vs b64 x 4, 8 waves per CU.
//   hipcc --offload-arch=gfx950 -O3 lds_unaligned.hip -o lds_unaligned
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__global__ void check(unsigned *out, int shift) {
    __shared__ __attribute__((aligned(16))) unsigned s[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) s[i] = i;
    __syncthreads();
    const unsigned a = 4u * (threadIdx.x * 5 + shift);   // word offset 5 l + shift: every residue mod 4 among the lanes
    uint4 v;
    unsigned long long w;
    asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a));
    asm volatile("ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(w) : "v"(a));
    unsigned *o = out + threadIdx.x * 6;
    o[0] = v.x, o[1] = v.y, o[2] = v.z, o[3] = v.w, o[4] = (unsigned)w, o[5] = (unsigned)(w >> 32);
}

template<int OP>
__global__ __launch_bounds__(512) void rate(const unsigned *addr, unsigned *out, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned s[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) s[i] = i * 2654435761u;
    __syncthreads();
    unsigned a[4];
    for (int j = 0; j < 4; j++) a[j] = addr[(blockIdx.x * 4 + j) * 512 + threadIdx.x];   // byte addresses, word aligned, random rows
    unsigned acc = 0;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (OP == 0) {
#pragma unroll
                for (int q = 0; q < 8; q++) { unsigned v; asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(a[j]), "n"(q * 4)); acc ^= v; }
            }
            if (OP == 1) {
#pragma unroll
                for (int q = 0; q < 2; q++) { uint4 v; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(a[j]), "n"(q * 16)); acc ^= v.x ^ v.y ^ v.z ^ v.w; }
            }
            if (OP == 2) {
#pragma unroll
                for (int q = 0; q < 4; q++) { unsigned long long v; asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(a[j]), "n"(q * 8)); acc ^= (unsigned)v ^ (unsigned)(v >> 32); }
            }
            if (OP == 3) {   // what the kernel does today: the 8 words at stride 1024 bytes (one plane each)
#pragma unroll
                for (int q = 0; q < 8; q++) { unsigned v; asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(a[j] & 1023u), "n"(q * 1024)); acc ^= v; }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(acc));
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template<int OP>
void run(const unsigned *da, unsigned *d, const char *name) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
    const int iters = 2000;
    rate<OP><<<256, 512>>>(da, d, 10);
    (void)hipEventRecord(e0);
    rate<OP><<<256, 512>>>(da, d, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    // per CU: 8 waves x iters x 4 gathers of 8 words
    printf("%-34s %7.2f ns per 8-word gather of a wave (8 waves per CU share the LDS)\n", name, ms * 1e6 / (iters * 4.0 * 8));
}

int main() {
    unsigned *d, h[64 * 6];
    (void)hipMalloc(&d, sizeof h);
    int bad = 0;
    for (int shift = 0; shift < 4; shift++) {
        check<<<1, 64>>>(d, shift);
        (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        for (int l = 0; l < 64; l++) {
            const unsigned w0 = l * 5 + shift;
            for (int j = 0; j < 4; j++) bad += h[l * 6 + j] != w0 + j;
            bad += h[l * 6 + 4] != w0 || h[l * 6 + 5] != w0 + 1;
        }
    }
    printf("ds_read_b128 / ds_read_b64 at 4-byte-aligned addresses: %s (%d wrong words)\n", bad ? "WRONG" : "correct", bad);
    const int n = 256 * 4 * 512;
    unsigned *ha = (unsigned *)malloc(n * 4), *da;
    srand(1);
    for (int i = 0; i < n; i++) ha[i] = 4u * (unsigned)(rand() % (4096 - 8));   // any word offset
    (void)hipMalloc(&da, n * 4);
    (void)hipMemcpy(da, ha, n * 4, hipMemcpyHostToDevice);
    run<3>(da, d, "8 x ds_read_b32, stride 1 KB (today)");
    run<0>(da, d, "8 x ds_read_b32, consecutive words");
    run<2>(da, d, "4 x ds_read_b64, 4-byte aligned");
    run<1>(da, d, "2 x ds_read_b128, 4-byte aligned");
    return bad ? 1 : 0;
}
