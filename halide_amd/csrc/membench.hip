// membench.hip — the practical HBM ceiling of the device the pipelines run on ("verify with a device copy
// microbenchmark; record achieved-copy GB/s as the practical ceiling").  Measurement tooling behind bench.py / bench_apps.py,
// not a pipeline.
//
// Two entry points:
//  * hlmi_membench: the round-1/2 figure — three naive grid-stride float4 kernels (copy / read-only / write-only), 4096 blocks.
//  * hlmi_membench_sweep: the SAME traffic through every combination of
//        - loads in flight per thread (1, 2, 4, 8 float4 issued before the first store),
//        - workgroups per CU of a persistent grid (1, 2, 4, 8) or one chunk per workgroup (a grid as large as the buffer),
//        - default / non-temporal stores / non-temporal loads and stores,
//    plus hipMemcpyDtoDAsync, each timed with HIP events over `iters` launches on buffers far larger than the 256 MB
//    Infinity Cache.  The best copy rate is what `roofline.hbm_copy_ceiling_gbs` reports; the whole table is returned as
//    JSON so that the run that quotes a ceiling also shows how it was found (MI355X_MICROARCH.md quotes 6.29 TB/s for a
//    float4 copy; the naive kernel reaches 4.5–5.1 TB/s on this pool).
#include "hlmi_internal.h"

#include <string>

using namespace hlmi;

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void mb_copy(const f32x4 *__restrict__ src, f32x4 *__restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
__global__ __launch_bounds__(256) void mb_read(const f32x4 *__restrict__ src, float *__restrict__ sink, size_t n) {
    f32x4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += src[i];
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[0] = acc.x;   // never true for the zero-filled source
}
__global__ __launch_bounds__(256) void mb_write(f32x4 *__restrict__ dst, size_t n) {
    const f32x4 v = {1.0f, 2.0f, 3.0f, 4.0f};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = v;
}

// MODE 0: copy, 1: read-only, 2: write-only.  A workgroup moves chunks of 256 * U float4 (U * 4 KB): every thread issues its U
// loads (stride 256 float4: each wave-instruction is one 1 KB contiguous run) before the first store.  NT: 0 default policy,
// 1 non-temporal stores, 2 non-temporal loads and stores.
template<int U, int NT, int MODE>
__global__ __launch_bounds__(256) void mb_stream(const f32x4 *__restrict__ src, f32x4 *__restrict__ dst, size_t nchunks) {
    f32x4 acc = {0, 0, 0, 0};
    for (size_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const size_t base = c * (size_t)(256 * U) + threadIdx.x;
        f32x4 v[U];
        if (MODE != 2) {
#pragma unroll
            for (int u = 0; u < U; u++) v[u] = NT == 2 ? __builtin_nontemporal_load(src + base + u * 256) : src[base + u * 256];
        }
        if (MODE == 1) {
#pragma unroll
            for (int u = 0; u < U; u++) acc += v[u];
        } else {
#pragma unroll
            for (int u = 0; u < U; u++) {
                const f32x4 w = MODE == 2 ? f32x4{1.0f, 2.0f, 3.0f, 4.0f} : v[u];
                if (NT >= 1) __builtin_nontemporal_store(w, dst + base + u * 256);
                else dst[base + u * 256] = w;
            }
        }
    }
    if (MODE == 1 && acc.x + acc.y + acc.z + acc.w == 12345.678f) dst[0] = acc;   // never true for the zero-filled source
}

// ---- calibration of the PMC byte counters by access width (MI355X_MICROARCH.md §HBM: FETCH_SIZE is calibrated only for 16-byte
// streaming reads).  Each kernel reads every byte of the buffer exactly once from HBM with one access shape:
//   mb_ld4 / mb_ld8 / mb_ld16: aligned 4 / 8 / 16-byte loads, consecutive lanes consecutive addresses
//   mb_ld8u: 8-byte loads at 4-byte misalignment (what a packed, 4-aligned float2 gather compiles to)
//   mb_ld8o: OVERLAPPING 8-byte loads at a 4-byte lane stride — ll_up0f's plane gathers (lane i reads columns c+i, c+i+1)
// so that rocprofv3's FETCH_SIZE (and the TCC request counters) of each can be divided by the known byte count.
struct __attribute__((packed, aligned(4))) F2P { float x, y; };
template<int MODE>
__global__ __launch_bounds__(256) void mb_width(const char *__restrict__ src, float *__restrict__ sink, size_t bytes) {
    float acc = 0.0f;
    const size_t per_wave_iter = MODE == 0 ? 256 : MODE == 1 ? 512 : MODE == 2 ? 1024 : MODE == 3 ? 512 : 256;
    const size_t lane = threadIdx.x & 63, wave = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = (size_t)gridDim.x * 4;
    for (size_t base = wave * per_wave_iter; base + per_wave_iter + 16 <= bytes; base += nwaves * per_wave_iter) {
        if (MODE == 0) acc += *reinterpret_cast<const float *>(src + base + 4 * lane);
        if (MODE == 1) { const float2 v = *reinterpret_cast<const float2 *>(src + base + 8 * lane); acc += v.x + v.y; }
        if (MODE == 2) { const f32x4 v = *reinterpret_cast<const f32x4 *>(src + base + 16 * lane); acc += v.x + v.y + v.z + v.w; }
        if (MODE == 3) { const F2P v = *reinterpret_cast<const F2P *>(src + base + 4 + 8 * lane); acc += v.x + v.y; }
        if (MODE == 4) { const F2P v = *reinterpret_cast<const F2P *>(src + base + 4 * lane); acc += v.x + v.y; }
    }
    if (acc == 12345.678f) sink[0] = acc;   // never true for the zero-filled source
}

struct Timer {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipStream_t st;
    explicit Timer(hipStream_t s) : st(s) {
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
    }
    ~Timer() {
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
    }
    template<typename F>
    double ms_per_iter(int iters, F &&launch) {   // two untimed launches first
        for (int it = -2; it < iters; it++) {
            if (it == 0) (void)hipEventRecord(e0, event_stream(st));
            launch();
        }
        (void)hipEventRecord(e1, event_stream(st));
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        return (double)ms / iters;
    }
};

template<int U, int NT, int MODE>
double run_stream(Timer &t, int iters, int blocks, const void *a, void *b, size_t n16) {
    const size_t nchunks = n16 / (size_t)(256 * U);
    const unsigned grid = blocks > 0 ? (unsigned)blocks : (unsigned)(nchunks < 0x7fffffffull ? nchunks : 0x7fffffffull);
    return t.ms_per_iter(iters, [&] {
        hipLaunchKernelGGL((mb_stream<U, NT, MODE>), dim3(grid), dim3(256), 0, t.st, (const f32x4 *)a, (f32x4 *)b, nchunks);
    });
}

template<int MODE>
double dispatch(Timer &t, int iters, int U, int NT, int blocks, const void *a, void *b, size_t n16) {
#define MB_CASE(u, nt) if (U == u && NT == nt) return run_stream<u, nt, MODE>(t, iters, blocks, a, b, n16);
    MB_CASE(1, 0) MB_CASE(2, 0) MB_CASE(4, 0) MB_CASE(8, 0)
    MB_CASE(1, 1) MB_CASE(2, 1) MB_CASE(4, 1) MB_CASE(8, 1)
    MB_CASE(1, 2) MB_CASE(2, 2) MB_CASE(4, 2) MB_CASE(8, 2)
#undef MB_CASE
    return -1.0;
}

struct Buffers {
    void *a = nullptr, *b = nullptr;
    size_t n16 = 0;
    int alloc(size_t bytes, hipStream_t st) {
        n16 = (bytes / 16) & ~(size_t)(256 * 8 - 1);   // a whole number of the largest chunks
        if (n16 == 0) return halide_error_code_bad_dimensions;
        if (hipMalloc(&a, n16 * 16) != hipSuccess || hipMalloc(&b, n16 * 16) != hipSuccess) {
            (void)hipFree(a);
            a = nullptr;
            return halide_error_code_device_malloc_failed;
        }
        (void)hipMemsetAsync(a, 0, n16 * 16, st);
        (void)hipMemsetAsync(b, 0, n16 * 16, st);
        return 0;
    }
    ~Buffers() {
        (void)hipFree(a);
        (void)hipFree(b);
    }
};

}  // namespace

// bytes: size of each buffer (rounded down to 16).  out_gbs[3] = {copy: 2*bytes per launch, read: bytes, write: bytes}, GB/s.
extern "C" int hlmi_membench(size_t bytes, int iters, int blocks, double *out_gbs) {
    DeviceCtx ctx;
    int r = acquire_device(nullptr, &ctx);
    if (r) return r;
    const size_t n = bytes / 16;
    if (n == 0 || iters < 1 || !out_gbs) return halide_error_code_bad_dimensions;
    if (blocks < 1) blocks = 256 * 16;
    void *a = nullptr, *b = nullptr;
    if (hipMalloc(&a, n * 16) != hipSuccess || hipMalloc(&b, n * 16) != hipSuccess) {
        (void)hipFree(a);
        return halide_error_code_device_malloc_failed;
    }
    (void)hipMemsetAsync(a, 0, n * 16, ctx.stream);
    (void)hipMemsetAsync(b, 0, n * 16, ctx.stream);
    {
        Timer t(ctx.stream);
        for (int which = 0; which < 3; which++) {
            const double ms = t.ms_per_iter(iters, [&] {
                if (which == 0) hipLaunchKernelGGL(mb_copy, dim3(blocks), dim3(256), 0, ctx.stream, (const f32x4 *)a, (f32x4 *)b, n);
                if (which == 1) hipLaunchKernelGGL(mb_read, dim3(blocks), dim3(256), 0, ctx.stream, (const f32x4 *)a, (float *)b, n);
                if (which == 2) hipLaunchKernelGGL(mb_write, dim3(blocks), dim3(256), 0, ctx.stream, (f32x4 *)b, n);
            });
            out_gbs[which] = (which == 0 ? 2.0 : 1.0) * (double)(n * 16) / (ms * 1e-3) / 1e9;
        }
    }
    (void)hipFree(a);
    (void)hipFree(b);
    return hipGetLastError() == hipSuccess ? 0 : halide_error_code_device_run_failed;
}

// Launches the five width-calibration kernels (mb_width<0..4>: ld4, ld8, ld16, ld8u, ld8o) `iters` times each over a buffer of
// `bytes`; out_gbs[5] = achieved GB/s.  Meant to be run under `rocprofv3 --pmc FETCH_SIZE` (scripts/gpu_r3_pmc.sh).
extern "C" int hlmi_membench_widths(size_t bytes, int iters, double *out_gbs) {
    DeviceCtx ctx;
    int r = acquire_device(nullptr, &ctx);
    if (r) return r;
    if (iters < 1 || !out_gbs) return halide_error_code_bad_dimensions;
    Buffers buf;
    if ((r = buf.alloc(bytes, ctx.stream))) return r;
    const size_t nbytes = buf.n16 * 16;
    const int cus = stream_cu_count(ctx.device, nullptr);
    Timer t(ctx.stream);
    for (int mode = 0; mode < 5; mode++) {
        const double ms = t.ms_per_iter(iters, [&] {
            const dim3 grid(cus * 8), block(256);
            switch (mode) {
                case 0: hipLaunchKernelGGL(mb_width<0>, grid, block, 0, ctx.stream, (const char *)buf.a, (float *)buf.b, nbytes); break;
                case 1: hipLaunchKernelGGL(mb_width<1>, grid, block, 0, ctx.stream, (const char *)buf.a, (float *)buf.b, nbytes); break;
                case 2: hipLaunchKernelGGL(mb_width<2>, grid, block, 0, ctx.stream, (const char *)buf.a, (float *)buf.b, nbytes); break;
                case 3: hipLaunchKernelGGL(mb_width<3>, grid, block, 0, ctx.stream, (const char *)buf.a, (float *)buf.b, nbytes); break;
                default: hipLaunchKernelGGL(mb_width<4>, grid, block, 0, ctx.stream, (const char *)buf.a, (float *)buf.b, nbytes); break;
            }
        });
        out_gbs[mode] = (double)nbytes / (ms * 1e-3) / 1e9;
    }
    return hipGetLastError() == hipSuccess ? 0 : halide_error_code_device_run_failed;
}

// The sweep described in the header.  Writes a JSON object
//   {"bytes": n, "iters": k, "cus": c, "copy": [{"u":..,"nt":..,"wg_per_cu":..,"gbs":..}, ...], "read": [...], "write": [...],
//    "memcpy_d2d_gbs": x, "best": {"copy": {...}, "read": {...}, "write": {...}}}
// into `json` (NUL-terminated, truncated to `cap`); returns the length the full text needs (call with cap = 0 to size), or a
// negative halide error code.
extern "C" long hlmi_membench_sweep(size_t bytes, int iters, char *json, size_t cap) {
    DeviceCtx ctx;
    int r = acquire_device(nullptr, &ctx);
    if (r) return r;
    if (iters < 1) return halide_error_code_bad_dimensions;
    static thread_local std::string text;   // sized by a first call with cap = 0, fetched by the second
    if (json == nullptr || cap == 0 || text.empty()) {
        Buffers buf;
        if ((r = buf.alloc(bytes, ctx.stream))) return r;
        const int cus = stream_cu_count(ctx.device, nullptr);
        Timer t(ctx.stream);
        const double nbytes = (double)buf.n16 * 16.0;
        std::string out = "{\"bytes\": " + std::to_string((unsigned long long)nbytes) + ", \"iters\": " + std::to_string(iters) +
                          ", \"cus\": " + std::to_string(cus);
        const char *names[3] = {"copy", "read", "write"};
        std::string best_txt[3];
        for (int mode = 0; mode < 3; mode++) {
            out += std::string(", \"") + names[mode] + "\": [";
            double best = 0;
            bool first = true;
            for (int nt = 0; nt <= 2; nt++) {
                if (mode == 1 && nt == 1) continue;   // read-only: nt 1 (stores only) is nt 0
                if (mode == 2 && nt == 2) continue;   // write-only: nt 2 is nt 1
                for (int u : {1, 2, 4, 8}) {
                    for (int wpc : {0, 1, 2, 4, 8}) {   // 0: one chunk per workgroup
                        const double ms = mode == 0 ? dispatch<0>(t, iters, u, nt, wpc * cus, buf.a, buf.b, buf.n16)
                                        : mode == 1 ? dispatch<1>(t, iters, u, nt, wpc * cus, buf.a, buf.b, buf.n16)
                                                    : dispatch<2>(t, iters, u, nt, wpc * cus, buf.a, buf.b, buf.n16);
                        const double gbs = (mode == 0 ? 2.0 : 1.0) * nbytes / (ms * 1e-3) / 1e9;
                        char row[160];
                        snprintf(row, sizeof row, "{\"u\": %d, \"nt\": %d, \"wg_per_cu\": %d, \"gbs\": %.1f}", u, nt, wpc, gbs);
                        out += (first ? "" : ", ") + std::string(row);
                        first = false;
                        if (gbs > best) best = gbs, best_txt[mode] = row;
                    }
                }
            }
            out += "]";
        }
        const double ms = t.ms_per_iter(iters, [&] { (void)hipMemcpyDtoDAsync((hipDeviceptr_t)buf.b, (hipDeviceptr_t)buf.a, (size_t)nbytes, ctx.stream); });
        char tail[96];
        snprintf(tail, sizeof tail, ", \"memcpy_d2d_gbs\": %.1f", 2.0 * nbytes / (ms * 1e-3) / 1e9);
        out += tail;
        out += ", \"best\": {\"copy\": " + best_txt[0] + ", \"read\": " + best_txt[1] + ", \"write\": " + best_txt[2] + "}}";
        if (hipGetLastError() != hipSuccess) return halide_error_code_device_run_failed;
        text = out;
    }
    if (json && cap) {
        snprintf(json, cap, "%s", text.c_str());
        const long need = (long)text.size();
        if (cap > text.size()) text.clear();   // delivered in full: the next call measures again
        return need;
    }
    return (long)text.size();
}
