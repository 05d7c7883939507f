// What device-wide synchronisation inside one launch costs on gfx950 (the numbers behind the ll_coarse decision, profiles/NOTES.md):
//   1. same-address atomic rate: N workgroups x K returning agent-scope atomicAdds on ONE counter -> ns per atomic (serialised)
//   2. load latency: a dependent pointer chase through a 64 MB buffer with plain loads and with agent-coherent (sc1) loads
//   3. a dependent launch: K empty kernels back to back on one stream -> us per launch
// hipcc --offload-arch=gfx950 -O3 sync_cost.hip -o sync_cost
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

__global__ void atomics(unsigned *ctr, int k, unsigned *sink) {
    unsigned acc = 0;
    if (threadIdx.x == 0)
        for (int i = 0; i < k; i++) acc += __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (acc == 0xffffffffu) *sink = acc;
}
template<bool COH>
__global__ void chase(const unsigned *next, int steps, unsigned *sink) {
    unsigned p = threadIdx.x * 9973u;
    for (int i = 0; i < steps; i++) p = COH ? __hip_atomic_load(next + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : next[p];
    if (p == 0xffffffffu) *sink = p;
}
__global__ void empty(unsigned *sink) {
    if (sink == nullptr) *sink = 0;
}
static double ms_of(hipEvent_t a, hipEvent_t b) {
    float ms;
    hipEventSynchronize(b);
    hipEventElapsedTime(&ms, a, b);
    return ms;
}
int main() {
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    unsigned *ctr, *sink;
    hipMalloc(&ctr, 256), hipMalloc(&sink, 4);
    hipMemset(ctr, 0, 256);
    for (int nwg : {1, 64, 256, 512, 2048}) {
        const int k = nwg == 1 ? 4000 : 64;
        atomics<<<nwg, 64>>>(ctr, 4, sink);
        hipEventRecord(e0);
        atomics<<<nwg, 64>>>(ctr, k, sink);
        hipEventRecord(e1);
        const double ms = ms_of(e0, e1);
        printf("atomics: %4d workgroups x %4d returning atomicAdd on one address: %8.1f us total, %6.1f ns per atomic (%s)\n", nwg, k, ms * 1e3,
               ms * 1e6 / ((double)nwg * k), nwg == 1 ? "latency of a dependent chain" : "throughput");
    }
    const size_t n = 16u << 20;   // 64 MB of indices: beyond the L2s, inside the Infinity Cache
    std::vector<unsigned> h(n);
    unsigned x = 12345u;
    for (size_t i = 0; i < n; i++) x = x * 1664525u + 1013904223u, h[i] = (x >> 8) % n;
    unsigned *d;
    hipMalloc(&d, n * 4);
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    for (int coh = 0; coh < 2; coh++) {
        const int steps = 2000;
        if (coh) chase<true><<<1, 64>>>(d, 10, sink); else chase<false><<<1, 64>>>(d, 10, sink);
        hipEventRecord(e0);
        if (coh) chase<true><<<1, 64>>>(d, steps, sink); else chase<false><<<1, 64>>>(d, steps, sink);
        hipEventRecord(e1);
        printf("dependent %s loads over 64 MB (one wave, 64 chains): %6.1f ns per step\n", coh ? "agent-coherent (sc1)" : "plain", ms_of(e0, e1) * 1e6 / steps);
    }
    // small working set (256 KB: L2-resident): what a re-read of a just-written small level costs
    for (size_t i = 0; i < (64u << 10); i++) h[i] = h[i] % (64u << 10);
    hipMemcpy(d, h.data(), (64u << 10) * 4, hipMemcpyHostToDevice);
    for (int coh = 0; coh < 2; coh++) {
        const int steps = 4000;
        if (coh) chase<true><<<1, 64>>>(d, steps, sink); else chase<false><<<1, 64>>>(d, steps, sink);
        hipEventRecord(e0);
        if (coh) chase<true><<<1, 64>>>(d, steps, sink); else chase<false><<<1, 64>>>(d, steps, sink);
        hipEventRecord(e1);
        printf("dependent %s loads over 256 KB (one wave, 64 chains): %6.1f ns per step\n", coh ? "agent-coherent (sc1)" : "plain", ms_of(e0, e1) * 1e6 / steps);
    }
    for (int k : {1, 16, 256}) {
        empty<<<1, 64>>>(sink);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int i = 0; i < k; i++) empty<<<256, 256>>>(sink);
        hipEventRecord(e1);
        printf("%3d dependent launches of an empty 256 x 256 kernel on one stream: %6.2f us per launch\n", k, ms_of(e0, e1) * 1e3 / k);
    }
    return 0;
}
