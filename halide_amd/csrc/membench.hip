// membench.hip — the practical HBM ceiling of the device the pipelines run on ("verify with a device copy
// microbenchmark; record achieved-copy GB/s as the practical ceiling").  Measurement tooling behind bench_apps.py, not a
// pipeline: three grid-stride kernels over buffers far larger than the 256 MB MALL — copy (read + write), read-only
// (sum kept alive by a never-true store) and write-only — each timed with HIP events over `iters` launches.
#include "hlmi_internal.h"

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
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int which = 0; which < 3; which++) {
        for (int it = -2; it < iters; it++) {   // two untimed launches first
            if (it == 0) (void)hipEventRecord(e0, event_stream(ctx.stream));
            if (which == 0) hipLaunchKernelGGL(mb_copy, dim3(blocks), dim3(256), 0, ctx.stream, (const f32x4 *)a, (f32x4 *)b, n);
            if (which == 1) hipLaunchKernelGGL(mb_read, dim3(blocks), dim3(256), 0, ctx.stream, (const f32x4 *)a, (float *)b, n);
            if (which == 2) hipLaunchKernelGGL(mb_write, dim3(blocks), dim3(256), 0, ctx.stream, (f32x4 *)b, n);
        }
        (void)hipEventRecord(e1, event_stream(ctx.stream));
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        out_gbs[which] = (which == 0 ? 2.0 : 1.0) * (double)(n * 16) * iters / (ms * 1e-3) / 1e9;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(a);
    (void)hipFree(b);
    return hipGetLastError() == hipSuccess ? 0 : halide_error_code_device_run_failed;
}
