// LDS cost of the access patterns cp_fused_tile's staging uses (gfx950): ns per wave-instruction, one workgroup of 256 threads per CU,
// the loop body is the LDS instruction only.  hipcc --offload-arch=gfx950 -O3 lds_subdword.hip -o lds_subdword
//   0  ds_write_b8   lane l -> byte 2 l + 3        (two lanes share a dword: the quad stage's curved bytes)
//   1  ds_write_b8   lane l -> byte 4 l            (one lane per dword)
//   2  ds_write_b16  lane l -> byte 2 l + 2        (two lanes share a dword, halves)
//   3  ds_write_b32  lane l -> dword l             (reference)
//   4  ds_read_u8    lane l -> byte of a 1 KB table at a RANDOM index       (the curve look-up on a noise image)
//   5  ds_read_u8    lane l -> byte of a 1 KB table at index base + (l >> 3) (neighbouring pixels alike: a real image)
//   6  ds_read_b32   lane l -> dword (random index >> 2)                     (same banks as 4, dword access)
#include <hip/hip_runtime.h>
#include <stdio.h>

template<int OP>
__global__ __launch_bounds__(256) void k(unsigned *out, int iters, unsigned seed) {
    __shared__ unsigned char s[4096];
    const int t = threadIdx.x, l = t & 63;
    for (int i = t; i < 1024; i += 256) reinterpret_cast<unsigned *>(s)[i] = i * 2654435761u;
    __syncthreads();
    unsigned r = (t * 2654435761u + seed) >> 7;
    unsigned acc = 0;
    const unsigned wbase = (t >> 6) * 512;
    for (int it = 0; it < iters; it++) {
        r = r * 1664525u + 1013904223u;
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const unsigned idx = (r >> (u * 3)) & 1023u;
            if (OP == 0) asm volatile("ds_write_b8 %0, %1" ::"v"(wbase + 2 * l + 3), "v"(acc + u) : "memory");
            if (OP == 1) asm volatile("ds_write_b8 %0, %1" ::"v"(wbase + 4 * l), "v"(acc + u) : "memory");
            if (OP == 2) asm volatile("ds_write_b16 %0, %1" ::"v"(wbase + 2 * l + 2), "v"(acc + u) : "memory");
            if (OP == 3) asm volatile("ds_write_b32 %0, %1" ::"v"(wbase + 4 * l), "v"(acc + u) : "memory");
            if (OP == 4) { unsigned v; asm volatile("ds_read_u8 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(idx) : "memory"); acc += v; }
            if (OP == 5) { unsigned v; asm volatile("ds_read_u8 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((idx & 960u) + (l >> 3)) : "memory"); acc += v; }
            if (OP == 6) { unsigned v; asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(idx & ~3u) : "memory"); acc += v; }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (acc == 0x12345678u) out[0] = acc + s[t];
}
template<int OP>
double run(unsigned *d) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
    const int iters = 2000;
    k<OP><<<256, 256>>>(d, 10, 1);
    (void)hipEventRecord(e0);
    k<OP><<<256, 256>>>(d, iters, 2);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e6 / ((double)iters * 8 * 4);   // ns per wave-instruction of one CU's LDS (4 waves of a workgroup take turns)
}
int main() {
    unsigned *d;
    (void)hipMalloc(&d, 4);
    const char *names[] = {"ds_write_b8, two lanes per dword", "ds_write_b8, one lane per dword", "ds_write_b16, two lanes per dword", "ds_write_b32",
                           "ds_read_u8 random of 1 KB", "ds_read_u8 eight lanes alike", "ds_read_b32 random of 1 KB"};
    const double r[] = {run<0>(d), run<1>(d), run<2>(d), run<3>(d), run<4>(d), run<5>(d), run<6>(d)};
    for (int i = 0; i < 7; i++) printf("%-36s %6.2f ns per wave-instruction per CU\n", names[i], r[i]);
    return 0;
}
