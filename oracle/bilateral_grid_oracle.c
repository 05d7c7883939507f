/* bilateral_grid_oracle.c — TEST INFRASTRUCTURE ONLY (see oracle_common.h).
 *
 * CPU restatement of /root/reference/apps/bilateral_grid/bilateral_grid_generator.cpp:14-67, s_sigma = 8
 * (GeneratorParam :8).  PARITY UNPINNED for floats (no golden image in the reference; its test only checks
 * "Success!", apps/bilateral_grid/CMakeLists.txt); this file defines the canonical order:
 *   - histogram update in RDom order, r.x fastest (:21-29; src/Func.h update semantics), starting from 0.0f
 *   - 5-tap blurs left to right: (((a + b*4) + c*6) + d*4) + e  (:33-47)
 *   - lerp(a,b,w) = a*(1-w) + b*w nested x -> y -> z (:58-64), out = interp0 / interp1 (:67)
 *   - 1.0f / r_sigma is one float division, shared by both uses (:25, :52)
 *   - x / s_sigma, x % s_sigma are floor division / non-negative modulo (src/IR.h:145-166)
 * Canon 1 (oracle_common.h): val * (1 / r_sigma) + 0.5f is one fma (both uses of 1 / r_sigma are separate multiplies);
 * each `tap * 4` / `tap * 6` of the blurs is contracted with the add it feeds (B5 below); lerp = fma(a, 1 - w, b * w).
 */
#include "oracle_common.h"

#define S_SIGMA 8
/* (((a + b * 4) + c * 6) + d * 4) + e */
static inline float B5(float a, float b, float c, float d, float e) { return o_mad(d, 4.0f, o_mad(c, 6.0f, o_mad(b, 4.0f, a))) + e; }

int oracle_bilateral_grid(const float *in, int W, int H, int in_sy, int X0, int Y0, float r_sigma, float *out, int out_sy) {
    if (W < 1 || H < 1) return -1;
    const float inv_r = 1.0f / r_sigma;
    const int zmax = (int)o_mad(1.0f, inv_r, 0.5f); /* largest bin a clamped value can hit */
    const int ZH = zmax + 1;                     /* histogram bins 0..zmax */
    const int ZD = zmax + 2;                     /* blurred planes 0..zmax+1 */
    const int gx0 = o_fdiv(X0, S_SIGMA), gx1 = o_fdiv(X0 + W - 1, S_SIGMA) + 1;
    const int gy0 = o_fdiv(Y0, S_SIGMA), gy1 = o_fdiv(Y0 + H - 1, S_SIGMA) + 1;
    const int GX = gx1 - gx0 + 1, GY = gy1 - gy0 + 1, HX = GX + 4, HY = GY + 4;
    /* histogram on cells [gx0-2, gx1+2] x [gy0-2, gy1+2], bins [0, zmax], 2 channels */
    float *hist = (float *)calloc((size_t)HX * HY * ZH * 2, sizeof(float));
#define HIST(x, y, z, c) hist[((((size_t)(z)) * HY + (y)) * HX + (x)) * 2 + (c)]
#pragma omp parallel for schedule(static)
    for (int cy = 0; cy < HY; cy++) {
        for (int cx = 0; cx < HX; cx++) {
            int gx = gx0 - 2 + cx, gy = gy0 - 2 + cy;
            for (int ry = 0; ry < S_SIGMA; ry++) {
                for (int rx = 0; rx < S_SIGMA; rx++) {
                    int px = o_clampi(gx * S_SIGMA + rx - S_SIGMA / 2, X0, X0 + W - 1) - X0;
                    int py = o_clampi(gy * S_SIGMA + ry - S_SIGMA / 2, Y0, Y0 + H - 1) - Y0;
                    float val = o_clampf(in[(size_t)py * in_sy + px], 0.0f, 1.0f);
                    int zi = (int)o_mad(val, inv_r, 0.5f);
                    HIST(cx, cy, zi, 0) += val;
                    HIST(cx, cy, zi, 1) += 1.0f;
                }
            }
        }
    }
    /* blurz on the same cells, z in [0, zmax+1]; histogram is 0 outside [0, zmax] */
    float *bz = (float *)malloc(sizeof(float) * (size_t)HX * HY * ZD * 2);
#define BZ(x, y, z, c) bz[((((size_t)(z)) * HY + (y)) * HX + (x)) * 2 + (c)]
#define HZ(x, y, z, c) (((z) >= 0 && (z) < ZH) ? HIST(x, y, z, c) : 0.0f)
#pragma omp parallel for schedule(static)
    for (int z = 0; z < ZD; z++)
        for (int y = 0; y < HY; y++)
            for (int x = 0; x < HX; x++)
                for (int c = 0; c < 2; c++)
                    BZ(x, y, z, c) = B5(HZ(x, y, z - 2, c), HZ(x, y, z - 1, c), HZ(x, y, z, c), HZ(x, y, z + 1, c), HZ(x, y, z + 2, c));
    /* blurx on x in [gx0, gx1] (GX), y still HY */
    float *bx = (float *)malloc(sizeof(float) * (size_t)GX * HY * ZD * 2);
#define BX(x, y, z, c) bx[((((size_t)(z)) * HY + (y)) * GX + (x)) * 2 + (c)]
#pragma omp parallel for schedule(static)
    for (int z = 0; z < ZD; z++)
        for (int y = 0; y < HY; y++)
            for (int x = 0; x < GX; x++)
                for (int c = 0; c < 2; c++)
                    BX(x, y, z, c) = B5(BZ(x, y, z, c), BZ(x + 1, y, z, c), BZ(x + 2, y, z, c), BZ(x + 3, y, z, c), BZ(x + 4, y, z, c));
    /* blury on y in [gy0, gy1] (GY) */
    float *by = (float *)malloc(sizeof(float) * (size_t)GX * GY * ZD * 2);
#define BY(x, y, z, c) by[((((size_t)(z)) * GY + (y)) * GX + (x)) * 2 + (c)]
#pragma omp parallel for schedule(static)
    for (int z = 0; z < ZD; z++)
        for (int y = 0; y < GY; y++)
            for (int x = 0; x < GX; x++)
                for (int c = 0; c < 2; c++)
                    BY(x, y, z, c) = B5(BX(x, y, z, c), BX(x, y + 1, z, c), BX(x, y + 2, z, c), BX(x, y + 3, z, c), BX(x, y + 4, z, c));
    /* trilinear slice + normalise */
#pragma omp parallel for schedule(static)
    for (int y = 0; y < H; y++) {
        for (int x = 0; x < W; x++) {
            int ax = X0 + x, ay = Y0 + y;
            float val = o_clampf(in[(size_t)y * in_sy + x], 0.0f, 1.0f);
            float zv = val * inv_r;
            int zi = (int)zv;
            float zf = zv - (float)zi;
            float xf = (float)o_fmod(ax, S_SIGMA) * 0.125f, yf = (float)o_fmod(ay, S_SIGMA) * 0.125f;
            int xi = o_fdiv(ax, S_SIGMA) - gx0, yi = o_fdiv(ay, S_SIGMA) - gy0;
            float r[2];
            for (int c = 0; c < 2; c++) {
                float a = o_lerp(o_lerp(BY(xi, yi, zi, c), BY(xi + 1, yi, zi, c), xf),
                                 o_lerp(BY(xi, yi + 1, zi, c), BY(xi + 1, yi + 1, zi, c), xf), yf);
                float b = o_lerp(o_lerp(BY(xi, yi, zi + 1, c), BY(xi + 1, yi, zi + 1, c), xf),
                                 o_lerp(BY(xi, yi + 1, zi + 1, c), BY(xi + 1, yi + 1, zi + 1, c), xf), yf);
                r[c] = o_lerp(a, b, zf);
            }
            out[(size_t)y * out_sy + x] = r[0] / r[1];
        }
    }
    free(hist), free(bz), free(bx), free(by);
    return 0;
}
