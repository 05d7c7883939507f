// Which compute units does a stream created by hipExtStreamCreateWithCUMask really use?  The library builds its four "64-CU partitions"
// with mask bit b set for b = part, part + 4, part + 8, ... < 256 (runtime.cpp, halide_hip_partition_stream).  Every workgroup of a
// long-running kernel records the hardware ids it runs on (HW_REG_HW_ID: CU / SH / SE, HW_REG_XCC_ID); the host counts distinct CUs per
// stream and times a pure-ALU kernel on 1 .. 4 partitions at once.
//   hipcc --offload-arch=gfx950 -O3 cu_mask_probe.hip -o cu_mask_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <set>
#include <vector>

__global__ void ids(unsigned *out, int spin) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    float v = threadIdx.x;
    for (int i = 0; i < spin; i++) v = v * 1.0001f + 0.5f;   // keep the workgroup resident so that the launch spreads
    if (threadIdx.x == 0) out[blockIdx.x * 2] = hw, out[blockIdx.x * 2 + 1] = xcc;
    if (v == 12345.0f) out[0] = 0;
}
__global__ void alu(float *out, int iters) {
    float a = threadIdx.x, b = blockIdx.x, c = 1.0f, d = 2.0f;
    for (int i = 0; i < iters; i++) {
        a = a * 1.0001f + 0.5f, b = b * 0.9999f + 0.25f, c = c * 1.0002f + 0.125f, d = d * 0.9998f + 0.0625f;
    }
    if (a + b + c + d == 12345.0f) out[0] = a;
}

int main() {
    int ncu = 0;
    (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
    printf("multiprocessor count %d\n", ncu);
    hipStream_t part[4];
    for (int p = 0; p < 4; p++) {
        std::vector<uint32_t> mask((ncu + 31) / 32, 0u);
        for (int b = p; b < ncu; b += 4) mask[b / 32] |= 1u << (b % 32);
        if (hipExtStreamCreateWithCUMask(&part[p], (uint32_t)mask.size(), mask.data()) != hipSuccess) { printf("mask refused\n"); return 1; }
    }
    const int NB = 4096;
    unsigned *d, *h = new unsigned[NB * 2];
    (void)hipMalloc(&d, NB * 2 * sizeof(unsigned));
    auto count = [&](hipStream_t s, const char *name) {
        ids<<<NB, 256, 0, s>>>(d, 20000);
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(h, d, NB * 2 * sizeof(unsigned), hipMemcpyDeviceToHost);
        std::set<unsigned> cus, xccs;
        for (int i = 0; i < NB; i++) {
            const unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 0xf;
            const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 0x1, se = (hw >> 13) & 0x7;   // gfx9 HW_ID: CU_ID [11:8], SH_ID [12], SE_ID [15:13]
            cus.insert((xcc << 16) | (se << 8) | (sh << 4) | cu), xccs.insert(xcc);
        }
        printf("%-22s distinct (xcc, se, sh, cu): %3zu on %zu XCCs\n", name, cus.size(), xccs.size());
    };
    count(0, "null stream");
    for (int p = 0; p < 4; p++) { char n[32]; snprintf(n, sizeof n, "partition %d of 4", p); count(part[p], n); }

    // other mask shapes: the first 64 bits, the first 32 bits, every 8th bit (one bit per XCC stripe if bits are dealt round-robin)
    {
        struct { const char *name; int lo, hi, step; } shapes[] = {{"bits 0..63", 0, 64, 1}, {"bits 0..31", 0, 32, 1}, {"bits 0,8,16,..", 0, ncu, 8}, {"bit 0 only", 0, 1, 1}};
        for (auto &sh : shapes) {
            std::vector<uint32_t> mask((ncu + 31) / 32, 0u);
            for (int b = sh.lo; b < sh.hi; b += sh.step) mask[b / 32] |= 1u << (b % 32);
            hipStream_t s;
            hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data());
            if (e != hipSuccess) { printf("%-22s refused (%d)\n", sh.name, (int)e); continue; }
            uint32_t back[16] = {0};
            hipError_t g = hipExtStreamGetCUMask(s, (uint32_t)mask.size(), back);
            count(s, sh.name);
            printf("    hipExtStreamGetCUMask -> %d, words %08x %08x %08x %08x %08x %08x %08x %08x\n", (int)g, back[0], back[1], back[2], back[3], back[4], back[5], back[6], back[7]);
            (void)hipStreamDestroy(s);
        }
        const char *env = getenv("HSA_CU_MASK"); printf("HSA_CU_MASK=%s ROC_GLOBAL_CU_MASK=%s\n", env ? env : "(unset)", getenv("ROC_GLOBAL_CU_MASK") ? getenv("ROC_GLOBAL_CU_MASK") : "(unset)");
    }
    float *fo;
    (void)hipMalloc(&fo, 64);
    auto timed = [&](int nparts, bool null_stream) {
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0, 0);
        (void)hipStreamSynchronize(0);
        if (null_stream) alu<<<256 * 8, 256, 0, 0>>>(fo, 200000);
        else for (int p = 0; p < nparts; p++) alu<<<256 * 8 / 4, 256, 0, part[p]>>>(fo, 200000);   // a quarter of the work per partition
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        return ms;
    };
    timed(1, true);
    printf("pure ALU, 2048 workgroups x 256 threads on the null stream: %.2f ms\n", timed(1, true));
    for (int k = 1; k <= 4; k++) printf("pure ALU, 512 workgroups on each of %d partition stream(s) at once: %.2f ms\n", k, timed(k, false));
    return 0;
}
