/* unsharp_oracle.c — TEST INFRASTRUCTURE ONLY (see oracle_common.h).
 *
 * CPU restatement of /root/reference/apps/unsharp/unsharp_generator.cpp:13-52 (sigma = 1.5, a GeneratorParam, :7):
 *   kernel(i) = exp(-i*i / (2 sigma^2)) / (sqrtf(2 pi) sigma)   i = 0..3 — constants: the simplifier folds exp_f32 of a
 *               constant with the HOST's double std::exp and rounds to float (src/Simplify_Call.cpp:767-780)
 *   gray      = 0.299 in(.,0) + 0.587 in(.,1) + 0.114 in(.,2)            input edge-clamped (repeat_edge, :20)
 *   blur_y    = k0 gray(y) + k1 (gray(y-1) + gray(y+1)) + k2 (gray(y-2) + gray(y+2)) + k3 (gray(y-3) + gray(y+3))
 *   blur_x    = the same along x on blur_y
 *   sharpen   = 2 gray - blur_x;  ratio = sharpen / gray;  output(x,y,c) = ratio(x,y) * input(x,y,c)   (input unclamped)
 * Canon 0: every operator rounds once, left to right as written; canon 1 (oracle_common.h): every product of gray and of the two
 * 7-tap sums is fused with the add it feeds (2 gray is exact: sharpen is the same in both).  PARITY UNPINNED, like the other
 * float pipelines.
 * Planar layout: in[c*in_sc + y*in_sy + x]; (ix0, iy0) = absolute coordinates of the input's first element, (W, H) its
 * extents; the output region starts at (ox0, oy0) and must lie inside the input (the final tap is unclamped).
 */
#include "oracle_common.h"

void oracle_unsharp_kernel(float k[4]) {
    const float kPi = 3.14159265358979310000f, sigma = 1.5f;
    const float den = sqrtf(2 * kPi) * sigma;
    for (int i = 0; i < 4; i++) {
        const float arg = (float)(-i * i) / (2 * sigma * sigma);
        k[i] = (float)exp((double)arg) / den;
    }
}

int oracle_unsharp(const float *in, int W, int H, long in_sy, long in_sc, int ix0, int iy0, float *out, int ox0, int oy0, int ow,
                   int oh, long out_sy, long out_sc) {
    if (W < 1 || H < 1 || ow < 0 || oh < 0) return -1;
    float k[4];
    oracle_unsharp_kernel(k);
#define IN(ax, ay, c) in[(long)(c) * in_sc + (long)(o_clampi((ay), iy0, iy0 + H - 1) - iy0) * in_sy + (o_clampi((ax), ix0, ix0 + W - 1) - ix0)]
#pragma omp parallel for schedule(static)
    for (int y = 0; y < oh; y++) {
        const int Y = oy0 + y;
        for (int x = 0; x < ow; x++) {
            const int X = ox0 + x;
            float by[7];
            float g0 = 0;
            for (int dx = -3; dx <= 3; dx++) {
                float g[7];
                for (int dy = -3; dy <= 3; dy++) {
                    g[dy + 3] = o_mad(0.114f, IN(X + dx, Y + dy, 2), o_mad2(0.299f, IN(X + dx, Y + dy, 0), 0.587f, IN(X + dx, Y + dy, 1)));
                }
                by[dx + 3] = o_mad(k[3], g[0] + g[6], o_mad(k[2], g[1] + g[5], o_mad2(k[0], g[3], k[1], g[2] + g[4])));
                if (dx == 0) g0 = g[3];
            }
            const float bx = o_mad(k[3], by[0] + by[6], o_mad(k[2], by[1] + by[5], o_mad2(k[0], by[3], k[1], by[2] + by[4])));
            const float sharpen = 2.0f * g0 - bx;
            const float ratio = sharpen / g0;
            for (int c = 0; c < 3; c++) {
                out[(long)c * out_sc + (long)y * out_sy + x] = ratio * in[(long)c * in_sc + (long)(Y - iy0) * in_sy + (X - ix0)];
            }
        }
    }
#undef IN
    return 0;
}
