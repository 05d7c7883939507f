/* camera_pipe_oracle.c — TEST INFRASTRUCTURE ONLY (see oracle_common.h).
 *
 * CPU restatement of /root/reference/apps/camera_pipe/camera_pipe_generator.cpp:
 *   shift (16,12) :406-413, hot_pixel_suppression :240-250, deinterleave :252-263, Demosaic :47-152,
 *   color_correct :265-299, apply_curve :301-366 (lutResample = 1), sharpen :368-404.
 * Integer stages follow src/IR.h:29-108,145-166 (u8/u16/i16 wrap, i32 no-overflow, floor division) and
 * src/IROperator.cpp:769-816 (int16 (x) uint8 -> int16) and are exact with respect to the reference by
 * construction.  The two float set-up computations (3x4 colour matrix, 1024-entry tone curve) use the canonical
 * float order of oracle_common.h (pow -> halide_exp(halide_log(x)*y), x/c -> x*(1/c)); PARITY UNPINNED for those
 * (no golden output in the reference) — a 1-ulp difference there could move a LUT entry or a matrix coefficient by 1.
 * Canon 1 (oracle_common.h) contracts the set-up's product sums (the matrix blend, the two branches of the contrast curve,
 * z * 255 + 0.5) and the polynomials inside pow.
 * Every Func is evaluated as a pure function straight from raw input; no boundary condition exists in the
 * pipeline: the caller's input must cover x in [10, W+21], y in [6, H+17] for a W x H output.
 */
#include "oracle_common.h"

typedef struct {
    const uint16_t *in;
    int in_sy, iw, ih; /* input, origin (0,0) */
    int16_t matrix[3][4];
    uint8_t curve[1024];
    uint8_t strength_x32;
} cp_t;

static inline uint16_t avg16(uint16_t a, uint16_t b) { return (uint16_t)(((uint32_t)a + b + 1) / 2); }
static inline uint8_t avg8(uint8_t a, uint8_t b) { return (uint8_t)(((uint16_t)a + b + 1) / 2); }
static inline uint16_t absd16(uint16_t a, uint16_t b) { return a > b ? (uint16_t)(a - b) : (uint16_t)(b - a); }
static inline uint16_t max16(uint16_t a, uint16_t b) { return a > b ? a : b; }

static inline uint16_t raw(const cp_t *p, int x, int y) { /* shifted(x,y) = input(x+16, y+12) */
    return p->in[(size_t)(y + 12) * p->in_sy + (x + 16)];
}
static inline uint16_t denoised(const cp_t *p, int x, int y) {
    uint16_t a = max16(max16(raw(p, x - 2, y), raw(p, x + 2, y)), max16(raw(p, x, y - 2), raw(p, x, y + 2)));
    uint16_t v = raw(p, x, y);
    v = v < a ? v : a; /* clamp(v, 0, a) = max(min(v, a), 0) */
    return v;
}
/* deinterleaved channels */
static inline uint16_t g_gr(const cp_t *p, int x, int y) { return denoised(p, 2 * x, 2 * y); }
static inline uint16_t r_r(const cp_t *p, int x, int y) { return denoised(p, 2 * x + 1, 2 * y); }
static inline uint16_t b_b(const cp_t *p, int x, int y) { return denoised(p, 2 * x, 2 * y + 1); }
static inline uint16_t g_gb(const cp_t *p, int x, int y) { return denoised(p, 2 * x + 1, 2 * y + 1); }

static inline uint16_t g_r(const cp_t *p, int x, int y) {
    uint16_t gv = avg16(g_gb(p, x, y - 1), g_gb(p, x, y)), gvd = absd16(g_gb(p, x, y - 1), g_gb(p, x, y));
    uint16_t gh = avg16(g_gr(p, x + 1, y), g_gr(p, x, y)), ghd = absd16(g_gr(p, x + 1, y), g_gr(p, x, y));
    return ghd < gvd ? gh : gv;
}
static inline uint16_t g_b(const cp_t *p, int x, int y) {
    uint16_t gv = avg16(g_gr(p, x, y + 1), g_gr(p, x, y)), gvd = absd16(g_gr(p, x, y + 1), g_gr(p, x, y));
    uint16_t gh = avg16(g_gb(p, x - 1, y), g_gb(p, x, y)), ghd = absd16(g_gb(p, x - 1, y), g_gb(p, x, y));
    return ghd < gvd ? gh : gv;
}
#define U16(e) ((uint16_t)(e))
static inline uint16_t r_gr(const cp_t *p, int x, int y) {
    uint16_t corr = U16(g_gr(p, x, y) - avg16(g_r(p, x, y), g_r(p, x - 1, y)));
    return U16(corr + avg16(r_r(p, x - 1, y), r_r(p, x, y)));
}
static inline uint16_t b_gr(const cp_t *p, int x, int y) {
    uint16_t corr = U16(g_gr(p, x, y) - avg16(g_b(p, x, y), g_b(p, x, y - 1)));
    return U16(corr + avg16(b_b(p, x, y), b_b(p, x, y - 1)));
}
static inline uint16_t r_gb(const cp_t *p, int x, int y) {
    uint16_t corr = U16(g_gb(p, x, y) - avg16(g_r(p, x, y), g_r(p, x, y + 1)));
    return U16(corr + avg16(r_r(p, x, y), r_r(p, x, y + 1)));
}
static inline uint16_t b_gb(const cp_t *p, int x, int y) {
    uint16_t corr = U16(g_gb(p, x, y) - avg16(g_b(p, x, y), g_b(p, x + 1, y)));
    return U16(corr + avg16(b_b(p, x, y), b_b(p, x + 1, y)));
}
static inline uint16_t r_b(const cp_t *p, int x, int y) {
    uint16_t corr = U16(g_b(p, x, y) - avg16(g_r(p, x, y), g_r(p, x - 1, y + 1)));
    uint16_t rp = U16(corr + avg16(r_r(p, x, y), r_r(p, x - 1, y + 1)));
    uint16_t rpd = absd16(r_r(p, x, y), r_r(p, x - 1, y + 1));
    corr = U16(g_b(p, x, y) - avg16(g_r(p, x - 1, y), g_r(p, x, y + 1)));
    uint16_t rn = U16(corr + avg16(r_r(p, x - 1, y), r_r(p, x, y + 1)));
    uint16_t rnd = absd16(r_r(p, x - 1, y), r_r(p, x, y + 1));
    return rpd < rnd ? rp : rn;
}
static inline uint16_t b_r(const cp_t *p, int x, int y) {
    uint16_t corr = U16(g_r(p, x, y) - avg16(g_b(p, x, y), g_b(p, x + 1, y - 1)));
    uint16_t bp = U16(corr + avg16(b_b(p, x, y), b_b(p, x + 1, y - 1)));
    uint16_t bpd = absd16(b_b(p, x, y), b_b(p, x + 1, y - 1));
    corr = U16(g_r(p, x, y) - avg16(g_b(p, x + 1, y), g_b(p, x, y - 1)));
    uint16_t bn = U16(corr + avg16(b_b(p, x + 1, y), b_b(p, x, y - 1)));
    uint16_t bnd = absd16(b_b(p, x + 1, y), b_b(p, x, y - 1));
    return bpd < bnd ? bp : bn;
}

/* demosaiced(x, y, c) at full resolution, int16 (:135-143) */
static inline void demosaic(const cp_t *p, int X, int Y, int16_t rgb[3]) {
    int x = o_fdiv(X, 2), y = o_fdiv(Y, 2);
    int ex = o_fmod(X, 2) == 0, ey = o_fmod(Y, 2) == 0;
    uint16_t r, g, b;
    if (ey) {
        if (ex) r = r_gr(p, x, y), g = g_gr(p, x, y), b = b_gr(p, x, y);
        else r = r_r(p, x, y), g = g_r(p, x, y), b = b_r(p, x, y);
    } else {
        if (ex) r = r_b(p, x, y), g = g_b(p, x, y), b = b_b(p, x, y);
        else r = r_gb(p, x, y), g = g_gb(p, x, y), b = b_gb(p, x, y);
    }
    rgb[0] = (int16_t)r, rgb[1] = (int16_t)g, rgb[2] = (int16_t)b;
}

static inline int32_t fdiv256(int32_t v) { return v >> 8; } /* floor division by 256 */

static inline void curved(const cp_t *p, int X, int Y, uint8_t out[3]) {
    int16_t d[3];
    demosaic(p, X, Y, d);
    int32_t ir = d[0], ig = d[1], ib = d[2];
    for (int c = 0; c < 3; c++) {
        const int16_t *m = p->matrix[c];
        int32_t v = ((m[3] + m[0] * ir) + m[1] * ig) + m[2] * ib;
        int16_t cc = (int16_t)fdiv256(v);
        out[c] = p->curve[o_clampi(cc, 0, 1023)];
    }
}

/* set-up: colour matrix (:265-275), tone curve (:301-350), sharpen strength (:370-372) */
void oracle_camera_pipe_setup(const float *m3200, const float *m7000 /* [3][4] row-major: (x=col, y=row) */, float color_temp,
                              float gamma, float contrast, float sharpen_strength, int blackLevel, int whiteLevel,
                              int16_t *matrix, uint8_t *curve, uint8_t *strength_x32) {
    const float k1 = 1.0f / 3200, k2 = 1.0f / 7000; /* C++ float constants in the generator */
    const float inv_den = 1.0f / (k2 - k1);         /* x / c -> x * fold(1 / c) */
    float alpha = (1.0f / color_temp - k1) * inv_den;
    for (int i = 0; i < 12; i++) {
        float val = o_mad2(m3200[i], alpha, m7000[i], 1.0f - alpha);
        matrix[i] = (int16_t)(val * 256.0f);
    }
    int minRaw = 0 + blackLevel, maxRaw = whiteLevel;
    float invRange = 1.0f / (float)(maxRaw - minRaw);
    float b = 2.0f - o_halide_pow(2.0f, contrast * (1.0f / 100.0f));
    float a = 2.0f - 2.0f * b;
    float inv_gamma = 1.0f / gamma;
    for (int x = 0; x < 1024; x++) {
        float xf = o_clampf((float)(x - minRaw) * invRange, 0.0f, 1.0f);
        float g = o_halide_pow(xf, inv_gamma);
        float z = g > 0.5f ? 1.0f - o_mad2(a * (1.0f - g), 1.0f - g, b, 1.0f - g) : o_mad2(a * g, g, b, g);
        uint8_t val = (uint8_t)o_clampf(o_mad(z, 255.0f, 0.5f), 0.0f, 255.0f);
        curve[x] = x <= minRaw ? 0 : (x > maxRaw ? 255 : val);
    }
    float s = o_clampf(sharpen_strength * 32.0f, 0.0f, 255.0f); /* u8_sat */
    *strength_x32 = (uint8_t)s;
}

int oracle_camera_pipe(const uint16_t *in, int in_w, int in_h, int in_sy, const float *m3200, const float *m7000,
                       float color_temp, float gamma, float contrast, float sharpen_strength, int blackLevel, int whiteLevel,
                       uint8_t *out, int W, int H, int out_sy, int out_sc) {
    if (W < 0 || H < 0) return -1;
    if (W > 0 && H > 0 && (W + 21 >= in_w || H + 17 >= in_h)) return -4; /* access out of bounds */
    cp_t p;
    p.in = in, p.in_sy = in_sy, p.iw = in_w, p.ih = in_h;
    oracle_camera_pipe_setup(m3200, m7000, color_temp, gamma, contrast, sharpen_strength, blackLevel, whiteLevel,
                             &p.matrix[0][0], p.curve, &p.strength_x32);
    /* curved on [-1, W] x [-1, H] */
    const int CW = W + 2, CH = H + 2;
    uint8_t *cv = (uint8_t *)malloc((size_t)CW * CH * 3);
#pragma omp parallel for schedule(static)
    for (int y = 0; y < CH; y++) {
        for (int x = 0; x < CW; x++) {
            uint8_t o[3];
            curved(&p, x - 1, y - 1, o);
            for (int c = 0; c < 3; c++) cv[((size_t)c * CH + y) * CW + x] = o[c];
        }
    }
#define CV(x, y, c) cv[((size_t)(c) * CH + ((y) + 1)) * CW + ((x) + 1)]
#pragma omp parallel for schedule(static)
    for (int y = 0; y < H; y++) {
        for (int x = 0; x < W; x++) {
            for (int c = 0; c < 3; c++) {
                uint8_t uy[3];
                for (int d = -1; d <= 1; d++) uy[d + 1] = avg8(avg8(CV(x + d, y - 1, c), CV(x + d, y + 1, c)), CV(x + d, y, c));
                uint8_t unsharp = avg8(avg8(uy[0], uy[2]), uy[1]);
                int16_t mask = (int16_t)((int16_t)CV(x, y, c) - (int16_t)unsharp);
                int16_t prod = (int16_t)(mask * (int16_t)p.strength_x32); /* int16 (x) uint8 -> int16, wraps */
                int16_t q = (int16_t)o_fdiv(prod, 32);
                int16_t s = (int16_t)((int16_t)CV(x, y, c) + q);
                out[(size_t)y * out_sy + x + (size_t)c * out_sc] = (uint8_t)(s < 0 ? 0 : (s > 255 ? 255 : s));
            }
        }
    }
    free(cv);
    return 0;
}
