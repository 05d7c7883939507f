/* max_filter_oracle.c — TEST INFRASTRUCTURE ONLY (see oracle_common.h).
 *
 * CPU restatement of /root/reference/apps/max_filter/max_filter_generator.cpp:14-53, AS WRITTEN (radius = 26, a
 * GeneratorParam, :10) — the GPU kernel computes the same function from its closed form (a max over a disc-like footprint)
 * for output rows y >= 0, so agreement with this literal evaluation is also the proof of that closed form.  (For output
 * rows -26..-12 — above the image — the construction as written is NOT the footprint max: a sample that starts below the
 * rows the update touches holds row 0 alone.  tests/test_max_filter.py pins that too.)
 *   input      = repeat_edge(input_) in x and y                                                               (:17-19)
 *   slices     = (int)ceilf(logf(radius) / logf(2)) + 1 = 6                                                   (:22)
 *   vert_log(x, y, c, t) = input(x, y, c); then for t = 1..slices-1, for y = -radius .. height-1 (RDom r, :28: min -radius,
 *                extent height + radius — rows outside that range keep the pure definition):
 *                vert_log(x, y, c, t) = max(vert_log(x, y, c, t-1), vert_log(x, y + clamp(1 << (t-1), 0, 2 radius), c, t-1))
 *   slice_for_radius(t) = (int)floor(log(2t + 1) / logf(2))                                                    (:37)
 *   vert(x, y, c, t)    = max(vert_log(x, y - t, c, s), vert_log(x, y + t + 1 - clamp(1 << s, 0, 2 radius), c, s)),
 *                s = clamp(slice_for_radius(t), 0, slices)                                                    (:42-45)
 *   filter_height(dx)   = #{dy in [0, radius] : dx*dx + dy*dy < (radius + 0.25f)^2}                           (:47-49)
 *   output(x, y, c)     = max over dx in [-radius, radius] of vert(x + dx, y, c, clamp(filter_height(dx), 0, radius + 1))
 * max(a, b) of floats = a > b ? a : b.  The only float arithmetic is the table slice_for_radius: 2t + 1 is odd, log2 of it
 * is never within 0.04 of an integer for t = 1..27, so libm's logf gives the same floor as the reference's own log
 * polynomial.  Everything else is comparisons: the result is exact, so parity is pinned by the algorithm alone.
 * Planar layout: in[c*in_sc + y*in_sy + x], first element at absolute (ix0, iy0), extents (W, H); output region (ox0, oy0) +
 * (ow, oh) — any region, inside the input or not.
 */
#include "oracle_common.h"

#define MF_RADIUS 26

void oracle_max_filter_tables(int slice_for_radius[MF_RADIUS + 2], int filter_height[2 * MF_RADIUS + 1]) {
    for (int t = 0; t <= MF_RADIUS + 1; t++) slice_for_radius[t] = (int)floorf(logf((float)(2 * t + 1)) / logf(2));
    const float lim = (MF_RADIUS + 0.25f) * (MF_RADIUS + 0.25f);
    for (int dx = -MF_RADIUS; dx <= MF_RADIUS; dx++) {
        int n = 0;
        for (int dy = 0; dy <= MF_RADIUS; dy++) n += ((float)(dx * dx + dy * dy) < lim) ? 1 : 0;
        filter_height[dx + MF_RADIUS] = n;
    }
}

int oracle_max_filter(const float *in, int W, int H, long in_sy, long in_sc, int ix0, int iy0, float *out, int ox0, int oy0, int ow,
                      int oh, int channels, long out_sy, long out_sc) {
    if (W < 1 || H < 1 || ow < 0 || oh < 0) return -1;
    if (ow == 0 || oh == 0) return 0;
    const int radius = MF_RADIUS;
    const int slices = (int)(ceilf(logf((float)radius) / logf(2))) + 1;
    int sfr[MF_RADIUS + 2], fh[2 * MF_RADIUS + 1];
    oracle_max_filter_tables(sfr, fh);
    /* rows of vert_log that exist: what vert reads for the output rows, what the update writes, and what the update reads */
    const int upd_lo = -radius, upd_hi = H - 1; /* RDom r.x: absolute rows, not offset by the input's min (:28) */
    int lo = oy0 - (radius + 1), hi = oy0 + oh - 1 + radius + 1;
    if (upd_lo < lo) lo = upd_lo;
    if (upd_hi > hi) hi = upd_hi;
    hi += 2 * radius;
    const int rows = hi - lo + 1, cols = ow + 2 * radius, cx0 = ox0 - radius;
    const size_t plane = (size_t)rows * cols;
    float *vl = (float *)malloc(sizeof(float) * plane * slices);
    if (!vl) return -2;
#define VL(s, y, x) vl[(size_t)(s) * plane + (size_t)((y) - lo) * cols + ((x) - cx0)]
    for (int c = 0; c < channels; c++) {
        const float *inc = in + (long)c * in_sc;
#pragma omp parallel for schedule(static)
        for (int y = lo; y <= hi; y++) {
            const long ry = (long)(o_clampi(y, iy0, iy0 + H - 1) - iy0) * in_sy;
            for (int x = cx0; x < cx0 + cols; x++) {
                const float v = inc[ry + (o_clampi(x, ix0, ix0 + W - 1) - ix0)];
                for (int s = 0; s < slices; s++) VL(s, y, x) = v;
            }
        }
        for (int s = 1; s < slices; s++) {
            const int step = o_clampi(1 << (s - 1), 0, 2 * radius);
#pragma omp parallel for schedule(static)
            for (int y = upd_lo; y <= upd_hi; y++) {
                for (int x = cx0; x < cx0 + cols; x++) {
                    const float a = VL(s - 1, y, x), b = VL(s - 1, y + step, x);
                    VL(s, y, x) = a > b ? a : b;
                }
            }
        }
#pragma omp parallel for schedule(static)
        for (int y = oy0; y < oy0 + oh; y++) {
            for (int x = ox0; x < ox0 + ow; x++) {
                float m = -INFINITY;
                for (int dx = -radius; dx <= radius; dx++) {
                    const int t = o_clampi(fh[dx + radius], 0, radius + 1);
                    const int s = o_clampi(sfr[t], 0, slices);
                    const float a = VL(s, y - t, x + dx), b = VL(s, y + t + 1 - o_clampi(1 << s, 0, 2 * radius), x + dx);
                    const float v = a > b ? a : b;
                    m = m > v ? m : v;
                }
                out[(long)c * out_sc + (long)(y - oy0) * out_sy + (x - ox0)] = m;
            }
        }
    }
#undef VL
    free(vl);
    return 0;
}
