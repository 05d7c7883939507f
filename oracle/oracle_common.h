/* oracle_common.h — TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the arithmetic primitives every pipeline oracle uses.  Nothing under oracle/ is
 * linked into, imported by, or executed from the product (halide_amd/, libhlmi.so); only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may call it, and only as the checker /
 * the timed CPU baseline.
 *
 * All paths below are relative to /root/reference.  Build with -ffp-contract=off: every rounding below is
 * written out.  The evaluation order is "expression order as written in the generator, IEEE binary32, round
 * to nearest even", with the simplifier's deterministic float rewrite x / c -> x * (1/c)
 * (src/Simplify_Div.cpp:204) applied, in one of TWO canonical forms selected at run time
 * (oracle_set_canon, canon_oracle.c):
 *   canon 0  no contraction: one rounding per operator of the expression.
 *   canon 1  FMA contraction, modelled on what LLVM's DAG combiner does with the `contract` flag the
 *            reference sets on every float operation (src/CodeGen_LLVM.cpp:483-500 setAllowContract,
 *            src/CodeGen_Internal.cpp:614 AllowFPOpFusion = Fast), non-aggressive fusion (x86):
 *              fadd(fmul(a, b), c)          -> fma(a, b, c)        o_mad
 *              fadd(c, fmul(a, b))          -> fma(a, b, c)        o_mad
 *              fadd(fmul(a, b), fmul(c, d)) -> fma(a, b, c * d)    o_mad2 (the first product is the one fused)
 *              fsub(fmul(a, b), c)          -> fma(a, b, -c)       o_mulsub
 *              fsub(c, fmul(a, b))          -> fma(-a, b, c)       o_msub
 *            for multiplies with ONE use; a product that is used twice stays a multiply.  Reassociation
 *            (the `reassoc` flag) is not part of either form: scripts/oracle_variants.py bounds it.
 * The library is built for one of the two (halide_amd/csrc/hlmi_device_math.h, HLMI_CANON_FMA; it reports which
 * through hlmi_canon_fma()) and the parity tests select the matching form here.
 */
#ifndef ORACLE_COMMON_H
#define ORACLE_COMMON_H

#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* the canonical form in force (canon_oracle.c); read by every helper below */
extern int o_canon_fma;
void oracle_set_canon(int fma);
int oracle_get_canon(void);
/* Variant study only (scripts/oracle_variants.py): 1 = the sums LLVM may re-associate under the `reassoc` flag
 * (src/CodeGen_LLVM.cpp:495) are evaluated as balanced trees instead of left to right — the four taps of local_laplacian's
 * down-sampling, the seven-term patch sums of nl_means.  Never a canonical form: no library build follows it. */
extern int o_reassoc;
void oracle_set_reassoc(int on);

/* a * b + c */
static inline float o_mad(float a, float b, float c) { return o_canon_fma ? fmaf(a, b, c) : a * b + c; }
/* a * b + c * d: the first product is contracted, the second stays a multiply */
static inline float o_mad2(float a, float b, float c, float d) { return o_canon_fma ? fmaf(a, b, c * d) : a * b + c * d; }
/* c - a * b */
static inline float o_msub(float c, float a, float b) { return o_canon_fma ? fmaf(-a, b, c) : c - a * b; }
/* a * b - c */
static inline float o_mulsub(float a, float b, float c) { return o_canon_fma ? fmaf(a, b, -c) : a * b - c; }

/* Halide integer division / modulo round toward -inf for positive divisors (src/IR.h:145-166). */
static inline int o_fdiv(int a, int b) {
    int q = a / b, r = a % b;
    return (r < 0) ? q - 1 : q;
}
static inline int o_fmod(int a, int b) {
    int r = a % b;
    return (r < 0) ? r + b : r;
}
static inline int o_clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
/* clamp(a, lo, hi) = max(min(a, hi), lo)  (src/IROperator.cpp clamp) */
static inline float o_clampf(float v, float lo, float hi) {
    float m = v < hi ? v : hi;
    return m > lo ? m : lo;
}
static inline float o_bits2f(uint32_t u) {
    float f;
    memcpy(&f, &u, 4);
    return f;
}
static inline uint32_t o_f2bits(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
}

/* lerp(zero, one, w) for floats = zero*(1-w) + one*w  (src/Lerp.cpp:82-83,127-128) */
static inline float o_lerp(float zero, float one, float w) { return o_mad2(zero, 1.0f - w, one, w); }

/* evaluate_polynomial (src/IROperator.cpp:33-65): even/odd Horner split on x^2, high order first.
 * A zero coefficient multiplies by x2 without the add. */
static inline float o_poly(float x, const float *c, int n) {
    float x2 = x * x;
    float even = c[0], odd = c[1];
    for (int i = 2; i < n; i++) {
        if ((i & 1) == 0) {
            even = (c[i] == 0.0f) ? even * x2 : o_mad(even, x2, c[i]);
        } else {
            odd = (c[i] == 0.0f) ? odd * x2 : o_mad(odd, x2, c[i]);
        }
    }
    return ((n & 1) == 0) ? o_mad(even, x, odd) : o_mad(odd, x, even);
}

/* halide_exp (src/IROperator.cpp:921-966).  one_over_ln2 = 1.0f / logf(2.0f) evaluated in float when
 * the compiler was built (:927) = 0x3fb8aa3b; ln2_part1/2 as written (:924-925). */
static inline float o_halide_exp(float x_full) {
    const float ln2_part1 = 0.6931457519f, ln2_part2 = 1.4286067653e-6f;
    const float one_over_ln2 = 1.0f / 0.693147182464599609375f; /* logf(2.0f) */
    static const float coeff[8] = {0.00031965933071842413f, 0.00119156835564003744f, 0.00848988645943932717f,
                                   0.04160188091348320655f, 0.16667983794100929562f, 0.49999899033463041098f,
                                   1.0f, 1.0f};
    float scaled = x_full * one_over_ln2;
    float k_real = floorf(scaled);
    int k = (int)k_real;
    float x = o_msub(x_full, k_real, ln2_part1);
    x = o_msub(x, k_real, ln2_part2);
    float result = o_poly(x, coeff, 8);
    int biased = k + 127;
    float two_to_the_n = o_bits2f((uint32_t)biased << 23);
    result = result * two_to_the_n;
    if (!(biased < 255)) result = INFINITY;
    if (!(biased > 0)) result = 0.0f;
    return result;
}

/* range_reduce_log + halide_log (src/IROperator.cpp:847-919) */
static inline float o_halide_log(float x_full) {
    static const float coeff[10] = {0.05111976432738144643f, -0.11793923497136414580f, 0.14971993724699017569f,
                                    -0.16862004708254804686f, 0.19980668101718729313f, -0.24991211576292837737f,
                                    0.33333435275479328386f, -0.50000106292873236491f, 1.0f, 0.0f};
    int use_nan = x_full < 0.0f, use_neg_inf = x_full == 0.0f;
    float patched = (use_nan || use_neg_inf) ? 1.0f : x_full;
    int32_t iv = (int32_t)o_f2bits(patched);
    int32_t no_exponent = iv & (int32_t)0x807fffff;
    int32_t new_exponent = no_exponent >> 22;
    int32_t new_biased = 127 - new_exponent;
    int32_t old_biased = iv >> 23;
    int32_t exponent = old_biased - new_biased;
    float reduced = o_bits2f((uint32_t)(no_exponent | (new_biased << 23)));
    float x1 = reduced - 1.0f;
    float result = o_poly(x1, coeff, 10);
    result = o_mad((float)exponent, 0.693147182464599609375f /* logf(2.0) */, result);
    if (use_nan) return NAN;
    if (use_neg_inf) return -INFINITY;
    return result;
}

/* pow(x, y) for non-constant y on LLVM CPU targets (src/CodeGen_LLVM.cpp:3925-3941): exp(log(abs(x)) * y) under the select
 * chain written there; abs clears the sign bit (of a NaN too); `iy % 2` on floats is a - b * floor(a / b) (src/CodeGen_LLVM.cpp
 * visit(Mod)) with a / 2 folded to a * 0.5f. */
static inline float o_halide_pow(float x, float y) {
    const float ax = o_bits2f(o_f2bits(x) & 0x7fffffffu);
    const float e = o_halide_exp(o_halide_log(ax) * y);
    if (x > 0.0f) return e;
    if (y == 0.0f) return 1.0f;
    if (x == 0.0f) return 0.0f;
    const float iy = floorf(y);
    if (y != iy) return NAN;
    const float r = iy - 2.0f * floorf(iy * 0.5f);
    return (r == 0.0f) ? e : -e;
}

/* fast_exp (src/IROperator.cpp:1616-1643) */
static inline float o_fast_exp(float x_full) {
    const float ln2 = 0.693147182464599609375f; /* logf(2.0) */
    static const float coeff[6] = {0.01314350012789660196f, 0.03668965196652099192f, 0.16873890085469545053f,
                                   0.49970514590562437052f, 1.0f, 1.0f};
    float scaled = x_full * (1.0f / ln2); /* x / logf(2) -> x * fold(1/c) */
    float k_real = floorf(scaled);
    float x = o_msub(x_full, k_real, ln2);
    float result = o_poly(x, coeff, 6);
    int k = (int)k_real;
    int biased = o_clampi(k + 127, 0, 255);
    float two_to_the_n = o_bits2f((uint32_t)biased << 23);
    return result * two_to_the_n;
}

#endif /* ORACLE_COMMON_H */
