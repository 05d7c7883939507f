/* nl_means_oracle.c — TEST INFRASTRUCTURE ONLY (see oracle_common.h).
 *
 * CPU restatement of /root/reference/apps/nl_means/nl_means_generator.cpp:24-63.  PARITY UNPINNED for
 * floats (no golden output in the reference).  Canonical order defined here:
 *   inv_sigma_sq = -1.0f / (((sigma*sigma) * float(patch)) * float(patch))                    (:24)
 *   dc = (a-b)*(a-b)  (pow(.,2) with a constant integer exponent is a multiply, src/IROperator.cpp:2451-2456)
 *   d  = ((0 + dc0) + dc1) + dc2                 inline `sum` starts from 0, RDom order        (:35-37)
 *   blur_d_y(x,y) = sum_{p=-(P/2)}^{P-1-(P/2)} d(x, y+p);  blur_d(x,y) = sum_p blur_d_y(x+p, y)  (:40-45)
 *   w = fast_exp(blur_d * inv_sigma_sq)                                                        (:49)
 *   S(x,y,c) = sum over s_dom (x fastest, then y) of w * clamped_with_alpha(x+sx, y+sy, c), alpha = 1.0f (:52-59)
 *   out = clamp(S_c / S_3, 0, 1)                                                               (:61-62)
 * Every tap goes through repeat_edge(input) (:27), channel index included.
 * Canon 1 (oracle_common.h) contracts the two multiply-adds of the update definitions: d += dc * dc and S += w * in
 * (the alpha lane's w * 1.0f is folded to w by the simplifier in either form).
 */
#include "oracle_common.h"

int oracle_nl_means(const float *in, int W, int H, int in_sy, int in_sc, int patch, int search, float sigma, float *out,
                    int out_sy, int out_sc) {
    if (W < 1 || H < 1 || patch < 1 || search < 1) return -1;
    const float inv = -1.0f / (((sigma * sigma) * (float)patch) * (float)patch);
    const int p0 = -(patch / 2), s0 = -(search / 2);
    const int pl = -p0, ph = patch - 1 + p0; /* taps p0..p0+patch-1 => halo pl below, ph above */
    /* d on [-pl, W-1+ph]^2 ; blur_d_y on x in [-pl, W-1+ph], y in [0,H-1] */
    const int DW = W + pl + ph, DH = H + pl + ph;
    float *d = (float *)malloc(sizeof(float) * (size_t)DW * DH);
    float *bdy = (float *)malloc(sizeof(float) * (size_t)DW * H);
    float *sum = (float *)calloc((size_t)W * H * 4, sizeof(float));
#define IN(x, y, c) in[(size_t)o_clampi((y), 0, H - 1) * in_sy + o_clampi((x), 0, W - 1) + (size_t)(c) * in_sc]
    for (int sy = s0; sy < s0 + search; sy++) {
        for (int sx = s0; sx < s0 + search; sx++) {
#pragma omp parallel for schedule(static)
            for (int y = 0; y < DH; y++) {
                for (int x = 0; x < DW; x++) {
                    int ax = x - pl, ay = y - pl;
                    float acc = 0.0f;
                    for (int c = 0; c < 3; c++) {
                        float t = IN(ax, ay, c) - IN(ax + sx, ay + sy, c);
                        acc = o_mad(t, t, acc);   /* d += (a - b)^2: the square has one use */
                    }
                    d[(size_t)y * DW + x] = acc;
                }
            }
#pragma omp parallel for schedule(static)
            for (int y = 0; y < H; y++) {
                for (int x = 0; x < DW; x++) {
                    float acc = 0.0f;
                    if (o_reassoc && patch == 7) { /* variant study: balanced tree */
                        const float *q = &d[(size_t)y * DW + x];
                        acc = ((q[0] + q[DW]) + (q[2 * (size_t)DW] + q[3 * (size_t)DW])) + ((q[4 * (size_t)DW] + q[5 * (size_t)DW]) + q[6 * (size_t)DW]);
                    } else {
                        for (int p = 0; p < patch; p++) acc = acc + d[(size_t)(y + p) * DW + x]; /* rows y+p0.. in abs coords */
                    }
                    bdy[(size_t)y * DW + x] = acc;
                }
            }
#pragma omp parallel for schedule(static)
            for (int y = 0; y < H; y++) {
                for (int x = 0; x < W; x++) {
                    float acc = 0.0f;
                    if (o_reassoc && patch == 7) {
                        const float *q = &bdy[(size_t)y * DW + x];
                        acc = ((q[0] + q[1]) + (q[2] + q[3])) + ((q[4] + q[5]) + q[6]);
                    } else {
                        for (int p = 0; p < patch; p++) acc = acc + bdy[(size_t)y * DW + x + p];
                    }
                    float w = o_fast_exp(acc * inv);
                    float *s = &sum[((size_t)y * W + x) * 4];
                    for (int c = 0; c < 3; c++) s[c] = o_mad(w, IN(x + sx, y + sy, c), s[c]);
                    s[3] = s[3] + w * 1.0f;
                }
            }
        }
    }
#pragma omp parallel for schedule(static)
    for (int y = 0; y < H; y++) {
        for (int x = 0; x < W; x++) {
            const float *s = &sum[((size_t)y * W + x) * 4];
            for (int c = 0; c < 3; c++) out[(size_t)y * out_sy + x + (size_t)c * out_sc] = o_clampf(s[c] / s[3], 0.0f, 1.0f);
        }
    }
    free(d), free(bdy), free(sum);
    return 0;
}
