/* harris_oracle.c — TEST INFRASTRUCTURE ONLY (see oracle_common.h).
 *
 * CPU restatement of /root/reference/apps/harris/harris_generator.cpp:7-62 (Harris corner response, no boundary
 * condition: the input must cover the output grown by 2 in x and y, filter.cpp:24-26):
 *   gray = 0.299 in(.,0) + 0.587 in(.,1) + 0.114 in(.,2)
 *   Iy   = g(-1,-1) a + g(-1,+1) b + g(0,-1) c + g(0,+1) d + g(+1,-1) a + g(+1,+1) b      a = -1.0f/12, b = 1.0f/12,
 *   Ix   = g(-1,-1) a + g(+1,-1) b + g(-1,0) c + g(+1,0) d + g(-1,+1) a + g(+1,+1) b      c = -2.0f/12, d = 2.0f/12
 *   Sxx, Syy, Sxy = sum3x3 of Ix Ix, Iy Iy, Ix Iy   (order: (x-1,y-1) (x-1,y) (x-1,y+1) (x,y-1) (x,y) (x,y+1) (x+1,..))
 *   out  = (Sxx Syy - Sxy Sxy) - (0.04 trace) trace,  trace = Sxx + Syy
 * All sums left to right as written, one rounding per operator (canon 0).  Canon 1 (oracle_common.h) contracts every product
 * that feeds an add: gray, Ix, Iy are fma chains; Ixx / Iyy / Ixy are INLINE in the reference's CPU schedule (:111-123
 * materialises gray, Ix, Iy only), so the sums of products Sxx, Syy, Sxy are fma chains too (their second term stays a
 * multiply); det = fma(Sxx, Syy, -(Sxy Sxy)); out = fma(-(0.04 trace), trace, det).
 * PARITY UNPINNED (no golden output in the reference).
 * Planar f32: in[c*in_sc + (y - iy0)*in_sy + (x - ix0)]; output region (ox0, oy0) + (ow, oh) in absolute coordinates.
 */
#include "oracle_common.h"

int oracle_harris(const float *in, long in_sy, long in_sc, int ix0, int iy0, float *out, int ox0, int oy0, int ow, int oh,
                  long out_sy) {
    const float a = -1.0f / 12, b = 1.0f / 12, c = -2.0f / 12, d = 2.0f / 12;
#define G(X, Y) o_mad(0.114f, in[2 * in_sc + (long)((Y) - iy0) * in_sy + ((X) - ix0)], \
                      o_mad2(0.299f, in[(long)((Y) - iy0) * in_sy + ((X) - ix0)], 0.587f, in[in_sc + (long)((Y) - iy0) * in_sy + ((X) - ix0)]))
/* ((((g0 k0 + g1 k1) + g2 k2) + g3 k3) + g4 k4) + g5 k5 */
#define D6(g0, k0, g1, k1, g2, k2, g3, k3, g4, k4, g5, k5) o_mad(g5, k5, o_mad(g4, k4, o_mad(g3, k3, o_mad(g2, k2, o_mad2(g0, k0, g1, k1)))))
#pragma omp parallel for schedule(static)
    for (int y = 0; y < oh; y++) {
        for (int x = 0; x < ow; x++) {
            const int X = ox0 + x, Y = oy0 + y;
            float gx[3][3], gy[3][3];
            for (int dx = -1; dx <= 1; dx++) {
                for (int dy = -1; dy <= 1; dy++) {
                    const int px = X + dx, py = Y + dy;
                    gy[dx + 1][dy + 1] = D6(G(px - 1, py - 1), a, G(px - 1, py + 1), b, G(px, py - 1), c, G(px, py + 1), d, G(px + 1, py - 1), a,
                                            G(px + 1, py + 1), b);
                    gx[dx + 1][dy + 1] = D6(G(px - 1, py - 1), a, G(px + 1, py - 1), b, G(px - 1, py), c, G(px + 1, py), d, G(px - 1, py + 1), a,
                                            G(px + 1, py + 1), b);
                }
            }
            /* sum3x3 of the inline product p[i][j] * q[i][j], left to right: the first product is contracted with the second (which
             * stays a multiply), every later one with the running sum */
#define S3(p, q) o_mad(p[2][2], q[2][2], o_mad(p[2][1], q[2][1], o_mad(p[2][0], q[2][0], o_mad(p[1][2], q[1][2], o_mad(p[1][1], q[1][1], \
                 o_mad(p[1][0], q[1][0], o_mad(p[0][2], q[0][2], o_mad2(p[0][0], q[0][0], p[0][1], q[0][1]))))))))
            const float sxx = S3(gx, gx), syy = S3(gy, gy), sxy = S3(gx, gy);
#undef S3
            const float det = o_mulsub(sxx, syy, sxy * sxy), trace = sxx + syy;
            out[(long)y * out_sy + x] = o_msub(det, 0.04f * trace, trace);
        }
    }
#undef G
#undef D6
    return 0;
}
