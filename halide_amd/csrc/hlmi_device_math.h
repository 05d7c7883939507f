// hlmi_device_math.h — device-side arithmetic primitives shared by the gfx950 kernels.
//
// These restate, for the GPU, the scalar semantics the reference's code generator gives the same
// operators on its CPU targets (paths relative to /root/reference); the kernels must be compiled with
// -ffp-contract=off and without fast-math so that every expression below rounds exactly once per
// operator, IEEE binary32 (division is the correctly-rounded `/`).  A fused multiply-add is issued ONLY where the
// canonical form asks for one, through the helpers below:
//   HLMI_CANON_FMA = 1 (default)  a multiply with one use that feeds an add or a subtract is contracted with it, the way LLVM's
//                                 DAG combiner contracts it under the `contract` flag the reference sets on every float operation
//                                 (src/CodeGen_LLVM.cpp:483-500, src/CodeGen_Internal.cpp:614): mad / mad2 / msub / mulsub
//   HLMI_CANON_FMA = 0            no contraction: one rounding per operator (`make VARIANT=_nofma EXTRA=-DHLMI_CANON_FMA=0`)
// oracle/oracle_common.h defines the two forms for the checker; hlmi_canon_fma() reports which one a library was built for.
//   integer division / modulo by a positive constant round toward -inf       src/IR.h:145-166
//   clamp(a, lo, hi) = max(min(a, hi), lo)                                    src/IROperator.h
//   float lerp(zero, one, w) = zero*(1-w) + one*w                             src/Lerp.cpp:82-83,127-128
//   evaluate_polynomial (even/odd Horner on x^2)                              src/IROperator.cpp:33-65
//   exp  -> halide_exp                                                        src/IROperator.cpp:921-966
//   log  -> halide_log + range_reduce_log                                     src/IROperator.cpp:847-919
//   pow  -> exp(log|x|*y) + select chain                                      src/CodeGen_LLVM.cpp:3925-3941
//   fast_exp                                                                  src/IROperator.cpp:1616-1643
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#ifndef HLMI_CANON_FMA
#define HLMI_CANON_FMA 1
#endif

namespace hlmi {
namespace dev {

constexpr bool CANON_FMA = HLMI_CANON_FMA != 0;
// a * b + c  (fadd(fmul(a, b), c) and fadd(c, fmul(a, b)))
__device__ __forceinline__ float mad(float a, float b, float c) { return CANON_FMA ? __builtin_fmaf(a, b, c) : a * b + c; }
// a * b + c * d: the first product is the one contracted
__device__ __forceinline__ float mad2(float a, float b, float c, float d) { return CANON_FMA ? __builtin_fmaf(a, b, c * d) : a * b + c * d; }
// c - a * b
__device__ __forceinline__ float msub(float c, float a, float b) { return CANON_FMA ? __builtin_fmaf(-a, b, c) : c - a * b; }
// a * b - c
__device__ __forceinline__ float mulsub(float a, float b, float c) { return CANON_FMA ? __builtin_fmaf(a, b, -c) : a * b - c; }

__device__ __forceinline__ int fdiv2(int a) { return a >> 1; }           // floor(a / 2)
__device__ __forceinline__ int fmod2(int a) { return a & 1; }            // a mod 2, always in {0,1}
__device__ __forceinline__ int fdiv8(int a) { return a >> 3; }
__device__ __forceinline__ int fmod8(int a) { return a & 7; }
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ float clampf(float v, float lo, float hi) {
    float m = v < hi ? v : hi;
    return m > lo ? m : lo;
}
__device__ __forceinline__ float lerpf(float zero, float one, float w) { return mad2(zero, 1.0f - w, one, w); }

template<int N>
__device__ __forceinline__ float poly(float x, const float (&c)[N]) {
    float x2 = x * x;
    float even = c[0], odd = c[1];
#pragma unroll
    for (int i = 2; i < N; i++) {
        if ((i & 1) == 0) {
            even = (c[i] == 0.0f) ? even * x2 : mad(even, x2, c[i]);
        } else {
            odd = (c[i] == 0.0f) ? odd * x2 : mad(odd, x2, c[i]);
        }
    }
    return ((N & 1) == 0) ? mad(even, x, odd) : mad(odd, x, even);
}

__device__ __forceinline__ float halide_exp(float x_full) {
    const float ln2_part1 = 0.6931457519f, ln2_part2 = 1.4286067653e-6f;
    const float one_over_ln2 = __uint_as_float(0x3fb8aa3bu);  // 1.0f / logf(2.0f)
    const float coeff[8] = {0.00031965933071842413f, 0.00119156835564003744f, 0.00848988645943932717f,
                            0.04160188091348320655f, 0.16667983794100929562f, 0.49999899033463041098f,
                            1.0f, 1.0f};
    float scaled = x_full * one_over_ln2;
    float k_real = floorf(scaled);
    int k = (int)k_real;
    float x = msub(x_full, k_real, ln2_part1);
    x = msub(x, k_real, ln2_part2);
    float result = poly<8>(x, coeff);
    int biased = k + 127;
    result = result * __uint_as_float((uint32_t)biased << 23);
    if (!(biased < 255)) result = __uint_as_float(0x7f800000u);
    if (!(biased > 0)) result = 0.0f;
    return result;
}

__device__ __forceinline__ float halide_log(float x_full) {
    const float coeff[10] = {0.05111976432738144643f, -0.11793923497136414580f, 0.14971993724699017569f,
                             -0.16862004708254804686f, 0.19980668101718729313f, -0.24991211576292837737f,
                             0.33333435275479328386f, -0.50000106292873236491f, 1.0f, 0.0f};
    bool use_nan = x_full < 0.0f, use_neg_inf = x_full == 0.0f;
    float patched = (use_nan || use_neg_inf) ? 1.0f : x_full;
    int32_t iv = (int32_t)__float_as_uint(patched);
    int32_t no_exponent = iv & (int32_t)0x807fffff;
    int32_t new_biased = 127 - (no_exponent >> 22);
    int32_t exponent = (iv >> 23) - new_biased;
    float reduced = __uint_as_float((uint32_t)(no_exponent | (new_biased << 23)));
    float result = poly<10>(reduced - 1.0f, coeff);
    result = mad((float)exponent, __uint_as_float(0x3f317218u) /* logf(2.0) */, result);
    if (use_nan) return __uint_as_float(0x7fc00000u);
    if (use_neg_inf) return __uint_as_float(0xff800000u);
    return result;
}

// pow_f32 as the LLVM back ends lower it (src/CodeGen_LLVM.cpp:3925-3941): exp(log(abs(x)) * y) under a select chain; abs clears the
// sign bit (of a NaN too), `iy % 2` is the float modulo a - b floor(a / b) with a / 2 folded to a * 0.5f
__device__ __forceinline__ float halide_pow(float x, float y) {
    const float ax = __uint_as_float(__float_as_uint(x) & 0x7fffffffu);
    const float e = halide_exp(halide_log(ax) * y);
    if (x > 0.0f) return e;                                   // strictly positive x
    if (y == 0.0f) return 1.0f;                               // x^0 == 1
    if (x == 0.0f) return 0.0f;                               // 0^y == 0
    const float iy = floorf(y);
    if (y != iy) return __uint_as_float(0x7fc00000u);         // negative x to a non-integer power
    const float r = iy - 2.0f * floorf(iy * 0.5f);            // 2 * floor(..) is exact: the same with or without contraction
    return (r == 0.0f) ? e : -e;                              // negative x to an even / odd power
}

__device__ __forceinline__ float fast_exp(float x_full) {
    const float ln2 = __uint_as_float(0x3f317218u);           // logf(2.0)
    const float inv_ln2 = __uint_as_float(0x3fb8aa3bu);       // fold(1 / logf(2.0))
    const float coeff[6] = {0.01314350012789660196f, 0.03668965196652099192f, 0.16873890085469545053f,
                            0.49970514590562437052f, 1.0f, 1.0f};
    float k_real = floorf(x_full * inv_ln2);
    float x = msub(x_full, k_real, ln2);
    float result = poly<6>(x, coeff);
    int biased = clampi((int)k_real + 127, 0, 255);
    return result * __uint_as_float((uint32_t)biased << 23);
}

}  // namespace dev
}  // namespace hlmi
