/* local_laplacian_oracle.c — TEST INFRASTRUCTURE ONLY (see oracle_common.h).
 *
 * CPU restatement of apps/local_laplacian/local_laplacian_generator.cpp:18-87 ("THE ALGORITHM") with
 * downsample :267-273 and upsample :276-282 of /root/reference.  PARITY UNPINNED for the float stages:
 * the reference ships no golden image for this pipeline (its only test is that the driver prints
 * "Success!", apps/local_laplacian/CMakeLists.txt:50-60) and the Halide compiler cannot be built in
 * this environment (no LLVM), so this file *defines* the canonical evaluation order (oracle_common.h).
 *
 * Every Func is a total function on Z^2; only the input is edge-clamped (:28).  Level j is therefore
 * evaluated on the region the levels above/below demand of it (bounds inference), NOT on ceil(W/2^j):
 *   R_0 = [0, W-1];  R_{j+1} = [fdiv(minR_j - 1, 2), fdiv(maxR_j + 1, 2)]         (source of `upsample`)
 *   G_{J-1} = R_{J-1};  G_j = R_j U [2*minG_{j+1} - 1, 2*maxG_{j+1} + 2]          (source of `downsample`)
 */
#include "oracle_common.h"

#define LL_MAXJ 20

/* ---- canonicalisation variants ------------------------------------------------------------------------------
 * What the reference's object computes for the float stages is the generator's expression AFTER Halide's
 * simplifier, compiled by LLVM with fast-math contraction allowed.  The simplifier part is deterministic and is
 * restated here as the CANONICAL form (variant 0):
 *   - x / c0 -> x * fold(1 / c0)                     src/Simplify_Div.cpp:204 (constants fold in double, then round
 *                                                      to float32: src/IRMatch.h:1014-1016)
 *   - (x * c0) * c1 -> x * fold(c0 * c1)              src/Simplify_Mul.cpp:70 (not gated on the type) — so
 *     gray = 0.299f * (u / 65535.0f) + ... is evaluated as u * C0 + ... with C_c = float(double(r) * double(coef_c)),
 *     r = float(1.0 / 65535.0)                        (generator :32, :36)
 *   - power-of-two scalings (/8.0f, /4.0f, /256.0f, /2.0f) move freely: exact.
 * What LLVM may add on top (contraction of mul+add into fma, src/CodeGen_LLVM.cpp:483-500,
 * src/CodeGen_Internal.cpp:614) is NOT deterministic across targets/LLVM versions; the other variants exist to MEASURE
 * how far the u16 output moves under the plausible alternatives (tests/test_oracle_variants.py, scripts/
 * oracle_variants.py):
 *   LL_VAR_SOURCE  gray in source order, (u * r) * coef (no constant folding)            — round 1's canonical form
 *   LL_VAR_FMA     every mul that feeds an add is contracted the way LLVM's DAG combiner does it
 *                  (fadd(fmul(a,b), c) -> fma(a,b,c), first operand first; single-use multiplies only)
 *   LL_VAR_DIV     floating = u / 65535.0f as a true division (no reciprocal rewrite; implies source order)
 * Bits combine (FMA | SOURCE etc.). */
enum { LL_VAR_SOURCE = 1, LL_VAR_FMA = 2, LL_VAR_DIV = 4 };
static int ll_var = 0;
void oracle_ll_set_variant(int v) { ll_var = v; }
int oracle_ll_get_variant(void) { return ll_var; }

/* contraction is in force under canon 1 (oracle_common.h) or when the variant asks for it on top of canon 0 (the Python
 * side sets canon 1 for the call then, so that halide_exp's polynomial is contracted with the rest) */
static inline int ll_fma(void) { return (ll_var & LL_VAR_FMA) || o_canon_fma; }
/* a * b + c, contracted or not */
static inline float v_mad(float a, float b, float c) { return ll_fma() ? fmaf(a, b, c) : a * b + c; }
/* a * b + c * d: the DAG combiner contracts the first multiply and keeps the second */
static inline float v_mad2(float a, float b, float c, float d) { return ll_fma() ? fmaf(a, b, c * d) : a * b + c * d; }
/* (a + 3 (b + c) + d) / 8 (:270-271), left to right; under the re-association study (oracle_common.h) as the balanced tree
 * (a + d) + 3 (b + c) */
static inline float ll_down4(float a, float b, float c, float d) {
    if (o_reassoc) return v_mad(3.0f, b + c, a + d) * 0.125f;
    return (v_mad(3.0f, b + c, a) + d) * 0.125f;
}
static inline float v_lerp(float zero, float one, float w) { return v_mad2(zero, 1.0f - w, one, w); }

static inline float ll_gray(float u0, float u1, float u2) {
    if (ll_var & LL_VAR_DIV) {
        float f0 = u0 / 65535.0f, f1 = u1 / 65535.0f, f2 = u2 / 65535.0f;
        return v_mad(0.114f, f2, v_mad2(0.299f, f0, 0.587f, f1));
    }
    const float r = (float)(1.0 / 65535.0);
    if (ll_var & LL_VAR_SOURCE) {
        float f0 = u0 * r, f1 = u1 * r, f2 = u2 * r;
        return v_mad(0.114f, f2, v_mad2(0.299f, f0, 0.587f, f1));
    }
    const float C0 = (float)((double)r * (double)0.299f), C1 = (float)((double)r * (double)0.587f),
                C2 = (float)((double)r * (double)0.114f);
    return v_mad(u2, C2, v_mad2(u0, C0, u1, C1));
}
/* the folded constants, for the product-side unit test (tests/test_local_laplacian.py) */
void oracle_ll_gray_constants(float *c) {
    const float r = (float)(1.0 / 65535.0);
    c[0] = (float)((double)r * (double)0.299f), c[1] = (float)((double)r * (double)0.587f), c[2] = (float)((double)r * (double)0.114f);
}

typedef struct {
    int x0, x1, y0, y1; /* inclusive */
    int w, h;
    float *p;
} plane_t;

static void plane_alloc(plane_t *pl, int x0, int x1, int y0, int y1) {
    pl->x0 = x0, pl->x1 = x1, pl->y0 = y0, pl->y1 = y1;
    pl->w = x1 - x0 + 1, pl->h = y1 - y0 + 1;
    pl->p = (float *)malloc(sizeof(float) * (size_t)pl->w * (size_t)pl->h);
}
static inline float P(const plane_t *pl, int x, int y) {
    /* callers only touch coordinates inside the region by construction; assert via abort in debug */
    return pl->p[(size_t)(y - pl->y0) * (size_t)pl->w + (size_t)(x - pl->x0)];
}
static inline float *PP(plane_t *pl, int x, int y) {
    return &pl->p[(size_t)(y - pl->y0) * (size_t)pl->w + (size_t)(x - pl->x0)];
}

/* downsample (:267-273): downy first, then downx; "/ 8.0f" -> "* 0.125f" (exact either way). */
static void downsample(const plane_t *f, plane_t *out) {
    /* downy needed on x in [2*out.x0-1, 2*out.x1+2], y in [out.y0, out.y1] */
    plane_t dy;
    plane_alloc(&dy, 2 * out->x0 - 1, 2 * out->x1 + 2, out->y0, out->y1);
#pragma omp parallel for schedule(static)
    for (int y = dy.y0; y <= dy.y1; y++) {
        for (int x = dy.x0; x <= dy.x1; x++) {
            float a = P(f, x, 2 * y - 1), b = P(f, x, 2 * y), c = P(f, x, 2 * y + 1), d = P(f, x, 2 * y + 2);
            *PP(&dy, x, y) = ll_down4(a, b, c, d);
        }
    }
#pragma omp parallel for schedule(static)
    for (int y = out->y0; y <= out->y1; y++) {
        for (int x = out->x0; x <= out->x1; x++) {
            float a = P(&dy, 2 * x - 1, y), b = P(&dy, 2 * x, y), c = P(&dy, 2 * x + 1, y), d = P(&dy, 2 * x + 2, y);
            *PP(out, x, y) = ll_down4(a, b, c, d);
        }
    }
    free(dy.p);
}

/* upsample (:276-282) evaluated at one point */
static inline float upx_at(const plane_t *f, int x, int y) {
    float w = (float)(o_fmod(x, 2) * 2 + 1) * 0.25f;
    return v_lerp(P(f, o_fdiv(x + 1, 2), y), P(f, o_fdiv(x - 1, 2), y), w);
}
static inline float up_at(const plane_t *f, int x, int y) {
    float w = (float)(o_fmod(y, 2) * 2 + 1) * 0.25f;
    return v_lerp(upx_at(f, x, o_fdiv(y + 1, 2)), upx_at(f, x, o_fdiv(y - 1, 2)), w);
}

/* remap LUT (:23-25): remap(i) = alpha * fx * exp(-fx * fx / 2.0f), fx = float(i) / 256.0f */
static float remap_at(int i, float alpha) {
    float fx = (float)i * (1.0f / 256.0f);
    return (alpha * fx) * o_halide_exp(((-fx) * fx) * 0.5f);
}

/* Exported helper: the LUT itself, for unit tests of the GPU LUT kernel. n = 2*(levels-1)*256+1. */
void oracle_ll_remap_lut(int levels, float alpha, float *lut) {
    int half = (levels - 1) * 256;
    for (int i = -half; i <= half; i++) lut[i + half] = remap_at(i, alpha);
}

float oracle_halide_exp(float x) { return o_halide_exp(x); }
float oracle_halide_log(float x) { return o_halide_log(x); }
float oracle_halide_pow(float x, float y) { return o_halide_pow(x, y); }
float oracle_fast_exp(float x) { return o_fast_exp(x); }
/* array forms, for the direct sweeps of the device-side primitives (tests/test_device_math.py) */
void oracle_halide_exp_v(const float *x, float *out, size_t n) {
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) out[i] = o_halide_exp(x[i]);
}
void oracle_halide_log_v(const float *x, float *out, size_t n) {
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) out[i] = o_halide_log(x[i]);
}
void oracle_fast_exp_v(const float *x, float *out, size_t n) {
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) out[i] = o_fast_exp(x[i]);
}
void oracle_halide_pow_v(const float *x, const float *y, float *out, size_t n) {
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) out[i] = o_halide_pow(x[i], y[i]);
}
void oracle_lerp_v(const float *a, const float *b, const float *w, float *out, size_t n) {
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) out[i] = o_lerp(a[i], b[i], w[i]);
}

/* Full pipeline.  in/out: planar u16 [3][H][W] with row stride `in_sy`/`out_sy` and plane stride
 * `in_sc`/`out_sc` (elements); (X0, Y0) = dim[0].min, dim[1].min of both buffers, i.e. the absolute
 * coordinate of element [0][0] (the pyramids are functions of absolute coordinates).  J = pyramid_levels (8 in the reference build, :10), levels = K.
 * If dbg_level >= 0, additionally copies outGPyramid[dbg_level] restricted to R_j into dbg (row-major,
 * width = R_j extent) — used by the tests to localise a GPU mismatch; pass -1/NULL otherwise.
 * Returns 0, or -1 on bad arguments. */
int oracle_local_laplacian(const uint16_t *in, int W, int H, int in_sy, int in_sc, int X0, int Y0, int J, int levels, float alpha,
                           float beta, uint16_t *out, int out_sy, int out_sc, int dbg_level, float *dbg) {
    if (J < 1 || J > LL_MAXJ || levels < 2 || W < 1 || H < 1) return -1;
    const int K = levels;
    int Rx0[LL_MAXJ], Rx1[LL_MAXJ], Ry0[LL_MAXJ], Ry1[LL_MAXJ];
    int Gx0[LL_MAXJ], Gx1[LL_MAXJ], Gy0[LL_MAXJ], Gy1[LL_MAXJ];
    Rx0[0] = X0, Rx1[0] = X0 + W - 1, Ry0[0] = Y0, Ry1[0] = Y0 + H - 1;
    for (int j = 0; j + 1 < J; j++) {
        Rx0[j + 1] = o_fdiv(Rx0[j] - 1, 2), Rx1[j + 1] = o_fdiv(Rx1[j] + 1, 2);
        Ry0[j + 1] = o_fdiv(Ry0[j] - 1, 2), Ry1[j + 1] = o_fdiv(Ry1[j] + 1, 2);
    }
    Gx0[J - 1] = Rx0[J - 1], Gx1[J - 1] = Rx1[J - 1], Gy0[J - 1] = Ry0[J - 1], Gy1[J - 1] = Ry1[J - 1];
    for (int j = J - 2; j >= 0; j--) {
        int a = 2 * Gx0[j + 1] - 1, b = 2 * Gx1[j + 1] + 2, c = 2 * Gy0[j + 1] - 1, d = 2 * Gy1[j + 1] + 2;
        Gx0[j] = a < Rx0[j] ? a : Rx0[j], Gx1[j] = b > Rx1[j] ? b : Rx1[j];
        Gy0[j] = c < Ry0[j] ? c : Ry0[j], Gy1[j] = d > Ry1[j] ? d : Ry1[j];
    }

    /* remap LUT */
    const int half = (K - 1) * 256;
    float *lut = (float *)malloc(sizeof(float) * (size_t)(2 * half + 1));
    oracle_ll_remap_lut(K, alpha, lut);

    /* gray on G_0 with clamped input coordinates (:28-36); see ll_gray for the canonical constants */
    plane_t gray;
    plane_alloc(&gray, Gx0[0], Gx1[0], Gy0[0], Gy1[0]);
#pragma omp parallel for schedule(static)
    for (int y = gray.y0; y <= gray.y1; y++) {
        int yc = o_clampi(y, Y0, Y0 + H - 1) - Y0;
        for (int x = gray.x0; x <= gray.x1; x++) {
            int xc = o_clampi(x, X0, X0 + W - 1) - X0;
            size_t o = (size_t)yc * (size_t)in_sy + (size_t)xc;
            *PP(&gray, x, y) = ll_gray((float)in[o], (float)in[o + (size_t)in_sc], (float)in[o + 2 * (size_t)in_sc]);
        }
    }

    /* processed Gaussian pyramid (:38-47): g[j][k], j >= 1 materialised; g0 is pointwise */
    plane_t(*g)[LL_MAXJ] = (plane_t(*)[LL_MAXJ])calloc((size_t)K, sizeof(plane_t[LL_MAXJ]));
    const float Km1 = (float)(K - 1);
    const float inv_Km1 = 1.0f / Km1; /* (1.0f / (levels - 1)) :41 */
#define G0_AT(gr, k) \
    (v_mad(beta, (gr) - (float)(k) * inv_Km1, (float)(k) * inv_Km1) + \
     lut[o_clampi((int)(((gr) * Km1) * 256.0f), 0, half) - 256 * (k) + half])
    for (int k = 0; k < K; k++) {
        plane_t g0;
        plane_alloc(&g0, Gx0[0], Gx1[0], Gy0[0], Gy1[0]);
#pragma omp parallel for schedule(static)
        for (int y = g0.y0; y <= g0.y1; y++) {
            for (int x = g0.x0; x <= g0.x1; x++) {
                float gr = P(&gray, x, y);
                *PP(&g0, x, y) = G0_AT(gr, k);
            }
        }
        const plane_t *prev = &g0;
        for (int j = 1; j < J; j++) {
            plane_alloc(&g[k][j], Gx0[j], Gx1[j], Gy0[j], Gy1[j]);
            downsample(prev, &g[k][j]);
            prev = &g[k][j];
        }
        free(g0.p);
    }

    /* Gaussian pyramid of the input (:57-61) */
    plane_t inG[LL_MAXJ];
    inG[0] = gray;
    for (int j = 1; j < J; j++) {
        plane_alloc(&inG[j], Gx0[j], Gx1[j], Gy0[j], Gy1[j]);
        downsample(&inG[j - 1], &inG[j]);
    }

    /* output pyramids (:50-54, :63-79), coarse to fine, each on R_j */
    plane_t outG[LL_MAXJ];
    for (int j = J - 1; j >= 0; j--) {
        plane_alloc(&outG[j], Rx0[j], Rx1[j], Ry0[j], Ry1[j]);
#pragma omp parallel for schedule(static)
        for (int y = Ry0[j]; y <= Ry1[j]; y++) {
            for (int x = Rx0[j]; x <= Rx1[j]; x++) {
                float level = P(&inG[j], x, y) * Km1;
                int li = o_clampi((int)level, 0, K - 2);
                float lf = level - (float)li;
                float l0, l1;
                if (j == 0) {
                    float gr = P(&gray, x, y);
                    l0 = G0_AT(gr, li);
                    l1 = G0_AT(gr, li + 1);
                } else {
                    l0 = P(&g[li][j], x, y);
                    l1 = P(&g[li + 1][j], x, y);
                }
                if (j < J - 1) { /* lPyramid[j] = gPyramid[j] - upsample(gPyramid[j+1]) */
                    l0 = l0 - up_at(&g[li][j + 1], x, y);
                    l1 = l1 - up_at(&g[li + 1][j + 1], x, y);
                }
                float outL = v_mad2(1.0f - lf, l0, lf, l1);
                *PP(&outG[j], x, y) = (j == J - 1) ? outL : up_at(&outG[j + 1], x, y) + outL;
            }
        }
    }
    if (dbg && dbg_level >= 0 && dbg_level < J) {
        memcpy(dbg, outG[dbg_level].p, sizeof(float) * (size_t)outG[dbg_level].w * (size_t)outG[dbg_level].h);
    }

    /* colour + u16 (:82-87); `input` here is the UNclamped input, read inside the image only */
    const float eps = 0.01f;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < H; y++) {
        for (int x = 0; x < W; x++) {
            float og = P(&outG[0], X0 + x, Y0 + y) + eps, gr = P(&gray, X0 + x, Y0 + y) + eps;
            for (int c = 0; c < 3; c++) {
                float v = ((float)in[(size_t)y * (size_t)in_sy + (size_t)x + (size_t)c * (size_t)in_sc] * og) / gr;
                out[(size_t)y * (size_t)out_sy + (size_t)x + (size_t)c * (size_t)out_sc] =
                    (uint16_t)o_clampf(v, 0.0f, 65535.0f);
            }
        }
    }

    for (int j = 0; j < J; j++) free(outG[j].p);
    for (int j = 1; j < J; j++) free(inG[j].p);
    for (int k = 0; k < K; k++) {
        for (int j = 1; j < J; j++) free(g[k][j].p);
    }
    free(g);
    free(gray.p);
    free(lut);
    return 0;
}
