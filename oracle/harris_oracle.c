/* harris_oracle.c — TEST INFRASTRUCTURE ONLY (see oracle_common.h).
 *
 * CPU restatement of /root/reference/apps/harris/harris_generator.cpp:7-62 (Harris corner response, no boundary
 * condition: the input must cover the output grown by 2 in x and y, filter.cpp:24-26):
 *   gray = 0.299 in(.,0) + 0.587 in(.,1) + 0.114 in(.,2)
 *   Iy   = g(-1,-1) a + g(-1,+1) b + g(0,-1) c + g(0,+1) d + g(+1,-1) a + g(+1,+1) b      a = -1.0f/12, b = 1.0f/12,
 *   Ix   = g(-1,-1) a + g(+1,-1) b + g(-1,0) c + g(+1,0) d + g(-1,+1) a + g(+1,+1) b      c = -2.0f/12, d = 2.0f/12
 *   Sxx, Syy, Sxy = sum3x3 of Ix Ix, Iy Iy, Ix Iy   (order: (x-1,y-1) (x-1,y) (x-1,y+1) (x,y-1) (x,y) (x,y+1) (x+1,..))
 *   out  = (Sxx Syy - Sxy Sxy) - (0.04 trace) trace,  trace = Sxx + Syy
 * All sums left to right as written, one rounding per operator.  PARITY UNPINNED (no golden output in the reference).
 * Planar f32: in[c*in_sc + (y - iy0)*in_sy + (x - ix0)]; output region (ox0, oy0) + (ow, oh) in absolute coordinates.
 */
#include "oracle_common.h"

int oracle_harris(const float *in, long in_sy, long in_sc, int ix0, int iy0, float *out, int ox0, int oy0, int ow, int oh,
                  long out_sy) {
    const float a = -1.0f / 12, b = 1.0f / 12, c = -2.0f / 12, d = 2.0f / 12;
#define G(X, Y) ((0.299f * in[(long)((Y) - iy0) * in_sy + ((X) - ix0)] + 0.587f * in[in_sc + (long)((Y) - iy0) * in_sy + ((X) - ix0)]) + \
                 0.114f * in[2 * in_sc + (long)((Y) - iy0) * in_sy + ((X) - ix0)])
#pragma omp parallel for schedule(static)
    for (int y = 0; y < oh; y++) {
        for (int x = 0; x < ow; x++) {
            const int X = ox0 + x, Y = oy0 + y;
            float ixx[3][3], iyy[3][3], ixy[3][3];
            for (int dx = -1; dx <= 1; dx++) {
                for (int dy = -1; dy <= 1; dy++) {
                    const int px = X + dx, py = Y + dy;
                    const float iy = ((((G(px - 1, py - 1) * a + G(px - 1, py + 1) * b) + G(px, py - 1) * c) + G(px, py + 1) * d) +
                                      G(px + 1, py - 1) * a) + G(px + 1, py + 1) * b;
                    const float ix = ((((G(px - 1, py - 1) * a + G(px + 1, py - 1) * b) + G(px - 1, py) * c) + G(px + 1, py) * d) +
                                      G(px - 1, py + 1) * a) + G(px + 1, py + 1) * b;
                    ixx[dx + 1][dy + 1] = ix * ix, iyy[dx + 1][dy + 1] = iy * iy, ixy[dx + 1][dy + 1] = ix * iy;
                }
            }
#define S3(f) ((((((((f[0][0] + f[0][1]) + f[0][2]) + f[1][0]) + f[1][1]) + f[1][2]) + f[2][0]) + f[2][1]) + f[2][2])
            const float sxx = S3(ixx), syy = S3(iyy), sxy = S3(ixy);
#undef S3
            const float det = sxx * syy - sxy * sxy, trace = sxx + syy;
            out[(long)y * out_sy + x] = det - (0.04f * trace) * trace;
        }
    }
#undef G
    return 0;
}
