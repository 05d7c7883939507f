// debug_math.hip — test hook: the device-side math primitives of hlmi_device_math.h evaluated directly on caller-supplied
// operands, so that tests/test_device_math.py can sweep them against the oracle's restatement (oracle/oracle_common.h) over the
// whole float range — in the pipelines they only ever see the LUT's, the tone curve's and nl_means' operand ranges.
// Reference semantics: /root/reference/src/IROperator.cpp:847-966 (halide_log, halide_exp), :1616-1643 (fast_exp),
// src/CodeGen_LLVM.cpp:3925-3941 (pow's select chain).  Not part of the reference ABI.
#include "hlmi_device_math.h"
#include "hlmi_internal.h"

using namespace hlmi;

namespace {

// fn: 0 halide_exp(x), 1 halide_log(x), 2 halide_pow(x, y), 3 fast_exp(x), 4 lerpf(x, y, w = z)
__global__ __launch_bounds__(256) void dbg_math(int fn, const float *__restrict__ x, const float *__restrict__ y, const float *__restrict__ z,
                                                float *__restrict__ out, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float r;
    switch (fn) {
        case 0: r = dev::halide_exp(x[i]); break;
        case 1: r = dev::halide_log(x[i]); break;
        case 2: r = dev::halide_pow(x[i], y[i]); break;
        case 3: r = dev::fast_exp(x[i]); break;
        default: r = dev::lerpf(x[i], y[i], z[i]); break;
    }
    out[i] = r;
}

}  // namespace

// x, y, z, out: host pointers to n floats (y / z may be NULL where the function does not read them).  Returns 0, < 0 = HIP error.
extern "C" int hlmi_debug_math(int fn, const float *x, const float *y, const float *z, float *out, size_t n) {
    if (fn < 0 || fn > 4 || !x || !out || n == 0) return -1;
    float *d[4] = {nullptr, nullptr, nullptr, nullptr};
    const float *h[3] = {x, y, z};
    int rc = 0;
    for (int i = 0; i < 4 && !rc; i++) {
        if (i < 3 && !h[i]) continue;
        if (hipMalloc(&d[i], sizeof(float) * n) != hipSuccess) rc = -2;
        else if (i < 3 && hipMemcpy(d[i], h[i], sizeof(float) * n, hipMemcpyHostToDevice) != hipSuccess) rc = -3;
    }
    if (!rc) {
        hipLaunchKernelGGL(dbg_math, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, fn, d[0], d[1] ? d[1] : d[0], d[2] ? d[2] : d[0], d[3], n);
        if (hipGetLastError() != hipSuccess || hipMemcpy(out, d[3], sizeof(float) * n, hipMemcpyDeviceToHost) != hipSuccess) rc = -4;
    }
    for (int i = 0; i < 4; i++) {
        if (d[i]) (void)hipFree(d[i]);
    }
    return rc;
}
