/* bgu_oracle.c — TEST INFRASTRUCTURE ONLY (see oracle_common.h).
 *
 * CPU restatement of apps/bgu/bgu_generator.cpp:268-488 (bilateral-guided upsampling: fit a 3x4 affine colour
 * transform per bilateral-grid cell from a low-res input/output pair, then slice the grid of transforms with the
 * high-res input) of /root/reference.  PARITY UNPINNED: the reference holds no golden output for this app and its
 * compiler cannot be built here; the canonical order is the generator's expression order after the simplifier's
 * deterministic rewrites (oracle_common.h), of which two apply:
 *   - gray: 0.25 a + 0.5 b + 0.25 c  ->  ((a + b*2) + c) * 0.25   (src/Simplify_Add.cpp:124 then :114; powers of
 *     two, so the value is the same outside the subnormal range)
 *   - the centre tap `* t3` with t3 = 1.0f disappears (src/Simplify_Mul.cpp, x * 1 -> x; exact either way)
 * and with one TARGET-DEPENDENT primitive: fast_inverse (:170) is
 *   - 1.0f / x, correctly rounded, on the reference's CUDA path (src/runtime/ptx_dev.ll:61-66, __nv_frcp_rn) and on
 *     wasm / Metal / WebGPU (src/runtime/wasm_math.ll:8-11, src/CodeGen_Metal_Dev.cpp:844)      -> BGU_VAR_CANONICAL
 *   - the 12-bit rcpss estimate on x86 (src/runtime/x86.ll:100-106)                             -> BGU_VAR_X86_RCP
 * The GPU library follows the CUDA definition (it replaces the GPU schedule, :571-669); the x86 variant exists to
 * measure how far the reference's CPU result is from it (tests/test_bgu.py).
 * The histogram's float sums are taken in the serial order of the CPU schedule (:529-532: r.x innermost, then r.y);
 * the reference's CUDA schedule accumulates with atomics (:589-597) and has no defined order at all.
 *
 * Every Func is a total function on Z^n (histogram is 0 wherever no sample lands), so each stage is evaluated on the
 * box its consumers read: line / blurx on the cells [cx0, cx1] x [cy0, cy1] x [0, nb + 1] the output region touches
 * (:441-476), blury 3 cells wider in x, blurz 3 wider in x and y, histogram 3 wider in z as well.
 * Canon 1 (oracle_common.h) contracts: the products of the seven-tap filters, the multiply-subtracts of the elimination and the
 * two substitutions, the three lerps and the affine transform of the slice.  The histogram's update adds one of 22 products
 * chosen by mux() (:307-315) — a select stands between the product and the add, so those stay two operations in both forms.
 */
#include "oracle_common.h"
#include <xmmintrin.h>

enum { BGU_VAR_CANONICAL = 0, BGU_VAR_X86_RCP = 1 };
#define BGU_NC 22

static inline float bgu_inv(float x, int variant) {
    if (variant == BGU_VAR_X86_RCP) return _mm_cvtss_f32(_mm_rcp_ss(_mm_set_ss(x)));
    return 1.0f / x;
}

/* solve_symmetric<4, 3> (:131-238), statement by statement; f = [A | b] (4 x 7).  x[j][k] = solution row j of rhs k */
static void bgu_solve(float f[4][7], int variant) {
    enum { M = 4, N = 3 };
    for (int j = 0; j < M; j++) {
        f[j][j] = bgu_inv(f[j][j], variant);                              /* :170 */
        for (int i = j + 1; i < M; i++) f[i][j] = f[i][j] * f[j][j];     /* :171-173 */
        for (int i = j + 1; i < M; i++) {                                 /* :178-187 */
            for (int k = j + 1; k < M; k++) {
                if (k < i) f[i][k] = f[k][i];
                else f[i][k] = o_msub(f[i][k], f[k][j], f[j][i]);
            }
        }
    }
    for (int k = 0; k < N; k++) {                                         /* :199-229 */
        for (int j = 0; j < M; j++)
            for (int i = 0; i < j; i++) f[j][M + k] = o_msub(f[j][M + k], f[j][i], f[i][M + k]);
        for (int j = 0; j < M; j++) f[j][M + k] = f[j][M + k] * f[j][j];
        for (int j = M - 1; j >= 0; j--)
            for (int i = j + 1; i < M; i++) f[j][M + k] = o_msub(f[j][M + k], f[i][j], f[i][M + k]);
    }
}

/* the seven-tap 1/d^3-like filter (:333-359): ((((((a t0 + b t1) + c t2) + d) + e t2) + f t1) + g t0) */
static inline float bgu_tap7(float a, float b, float c, float d, float e, float f, float g) {
    const float t0 = 1.0f / 64, t1 = 1.0f / 27, t2 = 1.0f / 8;
    return o_mad(g, t0, o_mad(f, t1, o_mad(e, t2, o_mad(c, t2, o_mad2(a, t0, b, t1)) + d)));
}

static inline int bgu_cell(int v, int big) { return (int)floorf((float)v / (float)big); }

/* splat: f32 [lc][lh][lw] planar (the low-res input), values: f32 [vc][vh][vw] (the low-res output), slice: f32
 * [3][H][W] (the high-res input); all with mins 0, each low-res image edge-clamped in EVERY dimension to its own box
 * (BoundaryConditions::repeat_edge, :270-271).  out: f32 [3][oh][ow] = output on [ox0, ox0 + ow) x [oy0, oy0 + oh).
 * line_out (optional): f32 [ncy][ncx][nb + 2][12], the fitted transforms, for stage-wise comparison; dims_out (optional)
 * receives {cx0, cy0, ncx, ncy, nz, big_sigma}.  Returns 0, -1 on allocation failure, -2 on arguments the generator
 * gives no meaning to (s_sigma < 1, r_sigma <= 0, empty low-res images). */
int oracle_bgu(float r_sigma, int s_sigma, const float *splat, int lw, int lh, int lc, const float *values, int vw, int vh, int vc,
               const float *slice, int W, int H, int ox0, int oy0, int ow, int oh, float *out, int variant, float *line_out,
               int *dims_out) {
    if (s_sigma < 1 || !(r_sigma > 0.0f) || lw < 1 || lh < 1 || lc < 1 || vw < 1 || vh < 1 || vc < 1) return -2;
    if (ow <= 0 || oh <= 0) return 0;
    /* :275-279 */
    const int ufx = (int)ceilf((float)W / (float)lw), ufy = (int)ceilf((float)H / (float)lh);
    const int uf = ufx > ufy ? ufx : ufy;
    const int big = s_sigma * uf;                                        /* :436 */
    if (big < 1) return -2;
    const float inv_r = 1.0f / r_sigma;                                  /* :302 */
    const int nb = (int)(1.0f / r_sigma);                                /* :458 */
    const int zmax = (int)rintf(inv_r);                                  /* largest bin a sample can land in (:300-302) */
    const int nz = nb + 2;                                               /* z = 0 .. nb + 1 are sliced (:461-476) */
    const int cx0 = bgu_cell(ox0, big), cx1 = bgu_cell(ox0 + ow - 1, big) + 1;
    const int cy0 = bgu_cell(oy0, big), cy1 = bgu_cell(oy0 + oh - 1, big) + 1;
    const int ncx = cx1 - cx0 + 1, ncy = cy1 - cy0 + 1;
    const int hx0 = cx0 - 3, hy0 = cy0 - 3, nhx = ncx + 6, nhy = ncy + 6, nhz = zmax + 1;
    if (dims_out) dims_out[0] = cx0, dims_out[1] = cy0, dims_out[2] = ncx, dims_out[3] = ncy, dims_out[4] = nz, dims_out[5] = big;

    float *hist = (float *)calloc((size_t)nhy * nhx * nhz * BGU_NC, sizeof(float));
    float *bz = (float *)malloc((size_t)nhy * nhx * nz * BGU_NC * sizeof(float));
    float *by = (float *)malloc((size_t)ncy * nhx * nz * BGU_NC * sizeof(float));
    float *line = (float *)malloc((size_t)ncy * ncx * nz * 12 * sizeof(float));
    if (!hist || !bz || !by || !line) {
        free(hist), free(bz), free(by), free(line);
        return -1;
    }
#define SPLAT(x, y, c) splat[((size_t)o_clampi(c, 0, lc - 1) * lh + o_clampi(y, 0, lh - 1)) * lw + o_clampi(x, 0, lw - 1)]
#define VALS(x, y, c) values[((size_t)o_clampi(c, 0, vc - 1) * vh + o_clampi(y, 0, vh - 1)) * vw + o_clampi(x, 0, vw - 1)]
#define HIST(x, y, z, c) hist[((((size_t)((y) - hy0)) * nhx + ((x) - hx0)) * nhz + (z)) * BGU_NC + (c)]
#define HISTZ(x, y, z, c) (((z) < 0 || (z) > zmax) ? 0.0f : HIST(x, y, z, c))
#define BZ(x, y, z, c) bz[((((size_t)((y) - hy0)) * nhx + ((x) - hx0)) * nz + (z)) * BGU_NC + (c)]
#define BY(x, y, z, c) by[((((size_t)((y) - cy0)) * nhx + ((x) - hx0)) * nz + (z)) * BGU_NC + (c)]
#define LINE(x, y, z, c) line[((((size_t)((y) - cy0)) * ncx + ((x) - cx0)) * nz + (z)) * 12 + (c)]

    /* ---- histogram (:289-318): per cell, the s_sigma x s_sigma samples in r.y-outer, r.x-inner order */
#pragma omp parallel for collapse(2) schedule(static)
    for (int y = hy0; y < hy0 + nhy; y++) {
        for (int x = hx0; x < hx0 + nhx; x++) {
            for (int ry = 0; ry < s_sigma; ry++) {
                for (int rx = 0; rx < s_sigma; rx++) {
                    const int sx = x * s_sigma + rx - s_sigma / 2, sy = y * s_sigma + ry - s_sigma / 2;   /* :294 */
                    const float sr = SPLAT(sx, sy, 0), sg = SPLAT(sx, sy, 1), sb = SPLAT(sx, sy, 2);
                    const float vr = VALS(sx, sy, 0), vg = VALS(sx, sy, 1), vb = VALS(sx, sy, 2);
                    float pos = ((sr + sg * 2.0f) + sb) * 0.25f;                                              /* :281-284, folded */
                    pos = o_clampf(pos, 0.0f, 1.0f);
                    const int zi = (int)rintf(pos * inv_r);                                                   /* :297 (ties to even) */
                    const float t[BGU_NC] = {sr * sr, sr * sg, sr * sb, sr, sg * sg, sg * sb, sg, sb * sb, sb, 1.0f,
                                             vr * sr, vr * sg, vr * sb, vr, vg * sr, vg * sg, vg * sb, vg,
                                             vb * sr, vb * sg, vb * sb, vb};                                 /* :307-315 */
                    float *h = &HIST(x, y, zi, 0);
                    for (int c = 0; c < BGU_NC; c++) h[c] = h[c] + t[c];
                }
            }
        }
    }
    /* ---- blur z, y, x (:333-359) */
#pragma omp parallel for schedule(static)
    for (int y = hy0; y < hy0 + nhy; y++)
        for (int x = hx0; x < hx0 + nhx; x++)
            for (int z = 0; z < nz; z++)
                for (int c = 0; c < BGU_NC; c++)
                    BZ(x, y, z, c) = bgu_tap7(HISTZ(x, y, z - 3, c), HISTZ(x, y, z - 2, c), HISTZ(x, y, z - 1, c), HISTZ(x, y, z, c),
                                              HISTZ(x, y, z + 1, c), HISTZ(x, y, z + 2, c), HISTZ(x, y, z + 3, c));
#pragma omp parallel for schedule(static)
    for (int y = cy0; y <= cy1; y++)
        for (int x = hx0; x < hx0 + nhx; x++)
            for (int z = 0; z < nz; z++)
                for (int c = 0; c < BGU_NC; c++)
                    BY(x, y, z, c) = bgu_tap7(BZ(x, y - 3, z, c), BZ(x, y - 2, z, c), BZ(x, y - 1, z, c), BZ(x, y, z, c), BZ(x, y + 1, z, c),
                                              BZ(x, y + 2, z, c), BZ(x, y + 3, z, c));
    /* ---- blur x + the solve (:363-425) */
#pragma omp parallel for schedule(static)
    for (int y = cy0; y <= cy1; y++) {
        for (int x = cx0; x <= cx1; x++) {
            for (int z = 0; z < nz; z++) {
                float bx[BGU_NC];
                for (int c = 0; c < BGU_NC; c++)
                    bx[c] = bgu_tap7(BY(x - 3, y, z, c), BY(x - 2, y, z, c), BY(x - 1, y, z, c), BY(x, y, z, c), BY(x + 1, y, z, c),
                                     BY(x + 2, y, z, c), BY(x + 3, y, z, c));
                const float lambda = 1e-1f;
                float f[4][7];
                f[0][0] = bx[0] + lambda, f[0][1] = bx[1], f[0][2] = bx[2], f[0][3] = bx[3];
                f[1][0] = bx[1], f[1][1] = bx[4] + lambda, f[1][2] = bx[5], f[1][3] = bx[6];
                f[2][0] = bx[2], f[2][1] = bx[5], f[2][2] = bx[7] + lambda, f[2][3] = bx[8];
                f[3][0] = bx[3], f[3][1] = bx[6], f[3][2] = bx[8], f[3][3] = bx[9] + lambda;
                for (int k = 0; k < 3; k++)
                    for (int j = 0; j < 4; j++) f[j][4 + k] = bx[10 + 4 * k + j];                              /* :388-399 */
                f[0][4] = f[0][4] + lambda, f[1][5] = f[1][5] + lambda, f[2][6] = f[2][6] + lambda;             /* :412-414 */
                bgu_solve(f, variant);
                for (int k = 0; k < 3; k++)
                    for (int j = 0; j < 4; j++) LINE(x, y, z, 4 * k + j) = f[j][4 + k];                         /* :417-433 */
            }
        }
    }
    if (line_out) memcpy(line_out, line, (size_t)ncy * ncx * nz * 12 * sizeof(float));

    /* ---- slice (:441-488) */
#pragma omp parallel for schedule(static)
    for (int yo = 0; yo < oh; yo++) {
        const int y = oy0 + yo;
        float yf = (float)y / (float)big;
        const int yi = (int)floorf(yf);
        yf = yf - (float)yi;
        for (int xo = 0; xo < ow; xo++) {
            const int x = ox0 + xo;
            float xf = (float)x / (float)big;
            const int xi = (int)floorf(xf);
            xf = xf - (float)xi;
            const float s0 = slice[((size_t)0 * H + y) * W + x], s1 = slice[((size_t)1 * H + y) * W + x], s2 = slice[((size_t)2 * H + y) * W + x];
            float val = ((s0 + s1 * 2.0f) + s2) * 0.25f;                                                       /* :286-289, folded */
            val = o_clampf(val, 0.0f, 1.0f);
            const float zv = val * (float)nb;
            const int zi = (int)zv;
            const float zf = zv - (float)zi;
            float m[12];
            for (int c = 0; c < 12; c++) {
                float mz[2];
                for (int dz = 0; dz < 2; dz++) {
                    const float y0 = o_lerp(LINE(xi, yi, zi + dz, c), LINE(xi, yi + 1, zi + dz, c), yf);       /* :444-447 */
                    const float y1 = o_lerp(LINE(xi + 1, yi, zi + dz, c), LINE(xi + 1, yi + 1, zi + dz, c), yf);
                    mz[dz] = o_lerp(y0, y1, xf);                                                                /* :452-455 */
                }
                m[c] = o_lerp(mz[0], mz[1], zf);                                                                /* :467-470 */
            }
            for (int c = 0; c < 3; c++) {
                const float v = o_mad(m[4 * c + 2], s2, o_mad2(m[4 * c], s0, m[4 * c + 1], s1)) + m[4 * c + 3];           /* :473-477 */
                out[((size_t)c * oh + yo) * ow + xo] = o_clampf(v, 0.0f, 1.0f);                                 /* :482 */
            }
        }
    }
    free(hist), free(bz), free(by), free(line);
    return 0;
}
