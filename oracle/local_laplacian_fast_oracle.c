/* local_laplacian_fast_oracle.c — TEST INFRASTRUCTURE ONLY (see oracle_common.h).
 *
 * A second, TUNED CPU evaluation of the local_laplacian pipeline (apps/local_laplacian/local_laplacian_generator.cpp:18-87,
 * downsample :267-273, upsample :276-282 of /root/reference), for the `cpu_baseline` leg of bench.py: the same
 * operations in the same order as local_laplacian_oracle.c (canonical variant; tests/test_local_laplacian.py checks the
 * two bit for bit), scheduled the way a CPU wants them —
 *   - level 0 of the processed pyramid (levels planes of the full image: 9 x 33 MB at 4K) is never stored: level 1 of a
 *     plane is made directly from the gray image, a band of rows per task with a four-row ring of level-0 rows
 *     (the reference's own CPU schedule computes gPyramid[0] at the consumer too: generator :121-131);
 *   - the last level of the output pyramid is fused with the colour stage;
 *   - one arena, kept between calls (the first oracle pays a page fault per 4 KB of every plane, every call);
 *   - every loop nest is one flat OpenMP loop over (plane, row) so that the small levels still feed all threads.
 * It is NOT Halide's autoscheduled x86 code (that needs the Halide compiler), hence cpu_baseline.kind stays "port". */
#ifndef FCANON
#include "oracle_common.h"

#include <omp.h>

#define FJ 20

typedef struct {
    int x0, x1, y0, y1, w, h;
    float *p;
} fpl_t;

static inline float FP(const fpl_t *pl, int x, int y) { return pl->p[(size_t)(y - pl->y0) * (size_t)pl->w + (size_t)(x - pl->x0)]; }
static inline float *FPP(fpl_t *pl, int x, int y) { return &pl->p[(size_t)(y - pl->y0) * (size_t)pl->w + (size_t)(x - pl->x0)]; }

/* cpu_baseline picks the thread count that runs fastest (on a 256-thread host the small levels make 16-32 threads faster
 * than all of them); n <= 0 restores the OpenMP default */
void oracle_set_threads(int n) { omp_set_num_threads(n > 0 ? n : omp_get_num_procs()); }

static float *g_arena = NULL;
static size_t g_arena_cap = 0;

static void set_box(fpl_t *pl, int x0, int x1, int y0, int y1) {
    pl->x0 = x0, pl->x1 = x1, pl->y0 = y0, pl->y1 = y1, pl->w = x1 - x0 + 1, pl->h = y1 - y0 + 1, pl->p = NULL;
}
static size_t box_floats(const fpl_t *pl) { return ((size_t)pl->w * (size_t)pl->h + 15) & ~(size_t)15; }


/* The evaluation itself is compiled twice — once per canonical form, the contraction decided at compile time so that the
 * timed loops carry no test of the switch — by including this file in itself. */
#define FN2(n, c) n##_c##c
#define FN1(n, c) FN2(n, c)
#define FN(n) FN1(n, FCANON)
#define FCANON 0
#define F_MAD(a, b, c) ((a) * (b) + (c))
#define F_MAD2(a, b, c, d) ((a) * (b) + (c) * (d))
#include "local_laplacian_fast_oracle.c"
#undef FCANON
#undef F_MAD
#undef F_MAD2
#define FCANON 1
#define F_MAD(a, b, c) fmaf((a), (b), (c))
#define F_MAD2(a, b, c, d) fmaf((a), (b), (c) * (d))
#include "local_laplacian_fast_oracle.c"
#undef FCANON
#undef F_MAD
#undef F_MAD2

int oracle_local_laplacian_fast(const uint16_t *in, int W, int H, int in_sy, int in_sc, int X0, int Y0, int J, int levels,
                                float alpha, float beta, uint16_t *out, int out_sy, int out_sc) {
    return o_canon_fma ? ll_fast_c1(in, W, H, in_sy, in_sc, X0, Y0, J, levels, alpha, beta, out, out_sy, out_sc)
                       : ll_fast_c0(in, W, H, in_sy, in_sc, X0, Y0, J, levels, alpha, beta, out, out_sy, out_sc);
}
#else  /* ---- the body, for canon FCANON ---- */
/* the canonical forms of local_laplacian_oracle.c (variant 0), for canon FCANON (oracle_common.h): 0 = one rounding per
 * operator, 1 = contracted (v_mad / v_mad2 there) */
static inline float FN(f_gray)(float u0, float u1, float u2) {
    const float r = (float)(1.0 / 65535.0);
    const float C0 = (float)((double)r * (double)0.299f), C1 = (float)((double)r * (double)0.587f), C2 = (float)((double)r * (double)0.114f);
    return F_MAD(u2, C2, F_MAD2(u0, C0, u1, C1));
}
static inline float FN(f_lerp)(float zero, float one, float w) { return F_MAD2(zero, 1.0f - w, one, w); }
static inline float FN(f_upx)(const fpl_t *f, int x, int y) {
    float w = (float)(o_fmod(x, 2) * 2 + 1) * 0.25f;
    return FN(f_lerp)(FP(f, o_fdiv(x + 1, 2), y), FP(f, o_fdiv(x - 1, 2), y), w);
}
static inline float FN(f_up)(const fpl_t *f, int x, int y) {
    float w = (float)(o_fmod(y, 2) * 2 + 1) * 0.25f;
    return FN(f_lerp)(FN(f_upx)(f, x, o_fdiv(y + 1, 2)), FN(f_upx)(f, x, o_fdiv(y - 1, 2)), w);
}
static inline float FN(f_down4)(float a, float b, float c, float d) { return (F_MAD(3.0f, b + c, a) + d) * 0.125f; }

/* one level of one or several planes: dy then dx, flat loops over (plane, row); tmp: the planes' dy images */
static void FN(down_planes)(const fpl_t *src, fpl_t *dst, fpl_t *tmp, int np) {
    const int th = tmp[0].h, dh = dst[0].h;
#pragma omp parallel for schedule(static)
    for (int t = 0; t < np * th; t++) {
        const int k = t / th, y = tmp[k].y0 + (t - k * th);
        float *o = FPP(&tmp[k], tmp[k].x0, y);
        const fpl_t *f = &src[k];
        for (int x = tmp[k].x0; x <= tmp[k].x1; x++)
            *o++ = FN(f_down4)(FP(f, x, 2 * y - 1), FP(f, x, 2 * y), FP(f, x, 2 * y + 1), FP(f, x, 2 * y + 2));
    }
#pragma omp parallel for schedule(static)
    for (int t = 0; t < np * dh; t++) {
        const int k = t / dh, y = dst[k].y0 + (t - k * dh);
        float *o = FPP(&dst[k], dst[k].x0, y);
        const fpl_t *f = &tmp[k];
        for (int x = dst[k].x0; x <= dst[k].x1; x++)
            *o++ = FN(f_down4)(FP(f, 2 * x - 1, y), FP(f, 2 * x, y), FP(f, 2 * x + 1, y), FP(f, 2 * x + 2, y));
    }
}

static int FN(ll_fast)(const uint16_t *in, int W, int H, int in_sy, int in_sc, int X0, int Y0, int J, int levels,
                                float alpha, float beta, uint16_t *out, int out_sy, int out_sc) {
    if (J < 2 || J > FJ || levels < 2 || W < 1 || H < 1) return -1;
    const int K = levels;
    int Rx0[FJ], Rx1[FJ], Ry0[FJ], Ry1[FJ], Gx0[FJ], Gx1[FJ], Gy0[FJ], Gy1[FJ];
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
    const int half = (K - 1) * 256, nlut = 2 * half + 1;

    /* ---- arena layout */
    fpl_t gray, inG[FJ], outG[FJ];
    fpl_t *g = (fpl_t *)malloc(sizeof(fpl_t) * (size_t)K * FJ);   /* g[k * FJ + j], j >= 1 */
    fpl_t *tmp = (fpl_t *)malloc(sizeof(fpl_t) * (size_t)(K + 1));
    size_t total = 0;
    set_box(&gray, Gx0[0], Gx1[0], Gy0[0], Gy1[0]);
    total += box_floats(&gray);
    for (int j = 1; j < J; j++) {
        set_box(&inG[j], Gx0[j], Gx1[j], Gy0[j], Gy1[j]);
        set_box(&outG[j], Rx0[j], Rx1[j], Ry0[j], Ry1[j]);
        total += box_floats(&inG[j]) + box_floats(&outG[j]);
        for (int k = 0; k < K; k++) {
            set_box(&g[k * FJ + j], Gx0[j], Gx1[j], Gy0[j], Gy1[j]);
            total += box_floats(&g[k * FJ + j]);
        }
    }
    /* dy images: the largest is level 1's of the input pyramid (one plane) or level 2's of the K planes */
    size_t tmp_one = 0;
    {
        fpl_t t1;
        set_box(&t1, 2 * Gx0[1] - 1, 2 * Gx1[1] + 2, Gy0[1], Gy1[1]);
        tmp_one = box_floats(&t1);
        if (J > 2) {
            fpl_t t2;
            set_box(&t2, 2 * Gx0[2] - 1, 2 * Gx1[2] + 2, Gy0[2], Gy1[2]);
            if (box_floats(&t2) * (size_t)K > tmp_one) tmp_one = box_floats(&t2) * (size_t)K;
        }
    }
    total += tmp_one + (size_t)((nlut + 15) & ~15);
    if (total > g_arena_cap) {
        free(g_arena);
        g_arena = (float *)malloc(sizeof(float) * total);
        g_arena_cap = g_arena ? total : 0;
        if (!g_arena) {
            free(g), free(tmp);
            return -1;
        }
    }
    float *cur = g_arena;
    gray.p = cur, cur += box_floats(&gray);
    for (int j = 1; j < J; j++) {
        inG[j].p = cur, cur += box_floats(&inG[j]);
        outG[j].p = cur, cur += box_floats(&outG[j]);
        for (int k = 0; k < K; k++) g[k * FJ + j].p = cur, cur += box_floats(&g[k * FJ + j]);
    }
    float *tmp_base = cur;
    cur += tmp_one;
    float *lut = cur;
    for (int i = -half; i <= half; i++) {
        float fx = (float)i * (1.0f / 256.0f);
        lut[i + half] = (alpha * fx) * o_halide_exp(((-fx) * fx) * 0.5f);
    }

    /* ---- gray on G_0 (clamped input coordinates) */
#pragma omp parallel for schedule(static)
    for (int y = gray.y0; y <= gray.y1; y++) {
        const int yc = o_clampi(y, Y0, Y0 + H - 1) - Y0;
        float *o = FPP(&gray, gray.x0, y);
        for (int x = gray.x0; x <= gray.x1; x++) {
            const int xc = o_clampi(x, X0, X0 + W - 1) - X0;
            const size_t i = (size_t)yc * (size_t)in_sy + (size_t)xc;
            *o++ = FN(f_gray)((float)in[i], (float)in[i + (size_t)in_sc], (float)in[i + 2 * (size_t)in_sc]);
        }
    }
    const float Km1 = (float)(K - 1), inv_Km1 = 1.0f / Km1;
#define F_G0(gr, k) \
    (F_MAD(beta, (gr) - (float)(k) * inv_Km1, (float)(k) * inv_Km1) + lut[o_clampi((int)(((gr) * Km1) * 256.0f), 0, half) - 256 * (k) + half])

    /* ---- level 1 of the K processed planes straight from gray: tasks = (plane, band of level-1 rows) */
    {
        const fpl_t *l1 = &g[0 * FJ + 1];
        const int dx0 = 2 * l1->x0 - 1, dx1 = 2 * l1->x1 + 2, dw = dx1 - dx0 + 1, BAND = 8;
        const int nb = (l1->h + BAND - 1) / BAND;
#pragma omp parallel
        {
            float *rows = (float *)malloc(sizeof(float) * (size_t)dw * 5);   /* four level-0 rows + the dy row */
#pragma omp for schedule(dynamic, 1)
            for (int t = 0; t < K * nb; t++) {
                const int k = t / nb, b = t - k * nb;
                fpl_t *dst = &g[k * FJ + 1];
                const int ya = dst->y0 + b * BAND, yb = (ya + BAND - 1 < dst->y1) ? ya + BAND - 1 : dst->y1;
                float *r[4] = {rows, rows + dw, rows + 2 * (size_t)dw, rows + 3 * (size_t)dw}, *dyr = rows + 4 * (size_t)dw;
                for (int y = ya; y <= yb; y++) {
                    /* rows 2y-1 .. 2y+2 of plane k's level 0; the last two of the row before are the first two of this one */
                    const int first = (y == ya) ? 0 : 2;
                    if (first) {
                        float *s0 = r[0], *s1 = r[1];
                        r[0] = r[2], r[1] = r[3], r[2] = s0, r[3] = s1;
                    }
                    for (int q = first; q < 4; q++) {
                        const int yy = 2 * y - 1 + q;
                        const float *gp = &gray.p[(size_t)(yy - gray.y0) * (size_t)gray.w + (size_t)(dx0 - gray.x0)];
                        float *o = r[q];
                        for (int x = 0; x < dw; x++) {
                            const float gr = gp[x];
                            o[x] = F_G0(gr, k);
                        }
                    }
                    for (int x = 0; x < dw; x++) dyr[x] = FN(f_down4)(r[0][x], r[1][x], r[2][x], r[3][x]);
                    float *o = FPP(dst, dst->x0, y);
                    for (int x = dst->x0; x <= dst->x1; x++) {
                        const float *d = dyr + (2 * x - 1 - dx0);
                        *o++ = FN(f_down4)(d[0], d[1], d[2], d[3]);
                    }
                }
            }
            free(rows);
        }
    }
    /* ---- levels 2 .. J-1 of the K planes */
    for (int j = 2; j < J; j++) {
        fpl_t src[64], dst[64];
        if (K > 64) {   /* more planes than the scratch arrays: one at a time */
            for (int k = 0; k < K; k++) {
                set_box(&tmp[0], 2 * Gx0[j] - 1, 2 * Gx1[j] + 2, Gy0[j], Gy1[j]);
                tmp[0].p = tmp_base;
                FN(down_planes)(&g[k * FJ + j - 1], &g[k * FJ + j], tmp, 1);
            }
            continue;
        }
        float *tp = tmp_base;
        for (int k = 0; k < K; k++) {
            src[k] = g[k * FJ + j - 1], dst[k] = g[k * FJ + j];
            set_box(&tmp[k], 2 * Gx0[j] - 1, 2 * Gx1[j] + 2, Gy0[j], Gy1[j]);
            tmp[k].p = tp, tp += box_floats(&tmp[k]);
        }
        FN(down_planes)(src, dst, tmp, K);
        for (int k = 0; k < K; k++) g[k * FJ + j].p = dst[k].p;
    }
    /* ---- Gaussian pyramid of the input */
    inG[0] = gray;
    for (int j = 1; j < J; j++) {
        set_box(&tmp[0], 2 * Gx0[j] - 1, 2 * Gx1[j] + 2, Gy0[j], Gy1[j]);
        tmp[0].p = tmp_base;
        FN(down_planes)(&inG[j - 1], &inG[j], tmp, 1);
    }
    /* ---- output pyramids, coarse to fine; level 0 goes straight into the colour stage */
    for (int j = J - 1; j >= 1; j--) {
#pragma omp parallel for schedule(static)
        for (int y = Ry0[j]; y <= Ry1[j]; y++) {
            float *o = FPP(&outG[j], Rx0[j], y);
            for (int x = Rx0[j]; x <= Rx1[j]; x++) {
                const float level = FP(&inG[j], x, y) * Km1;
                const int li = o_clampi((int)level, 0, K - 2);
                const float lf = level - (float)li;
                float l0 = FP(&g[li * FJ + j], x, y), l1 = FP(&g[(li + 1) * FJ + j], x, y);
                if (j < J - 1) {
                    l0 = l0 - FN(f_up)(&g[li * FJ + j + 1], x, y);
                    l1 = l1 - FN(f_up)(&g[(li + 1) * FJ + j + 1], x, y);
                }
                const float outL = F_MAD2(1.0f - lf, l0, lf, l1);
                *o++ = (j == J - 1) ? outL : FN(f_up)(&outG[j + 1], x, y) + outL;
            }
        }
    }
    const float eps = 0.01f;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < H; y++) {
        for (int x = 0; x < W; x++) {
            const int X = X0 + x, Y = Y0 + y;
            const float gr = FP(&gray, X, Y);
            const float level = gr * Km1;
            const int li = o_clampi((int)level, 0, K - 2);
            const float lf = level - (float)li;
            float l0 = F_G0(gr, li), l1 = F_G0(gr, li + 1);
            l0 = l0 - FN(f_up)(&g[li * FJ + 1], X, Y);
            l1 = l1 - FN(f_up)(&g[(li + 1) * FJ + 1], X, Y);
            const float outL = F_MAD2(1.0f - lf, l0, lf, l1);
            const float og = (FN(f_up)(&outG[1], X, Y) + outL) + eps, gre = gr + eps;
            for (int c = 0; c < 3; c++) {
                const float v = ((float)in[(size_t)y * (size_t)in_sy + (size_t)x + (size_t)c * (size_t)in_sc] * og) / gre;
                out[(size_t)y * (size_t)out_sy + (size_t)x + (size_t)c * (size_t)out_sc] = (uint16_t)o_clampf(v, 0.0f, 65535.0f);
            }
        }
    }
#undef F_G0
    free(g), free(tmp);
    return 0;
}
#endif
