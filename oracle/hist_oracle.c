/* hist_oracle.c — TEST INFRASTRUCTURE ONLY (see oracle_common.h).
 *
 * CPU restatement of /root/reference/apps/hist/hist_generator.cpp:13-56 (histogram equalisation of the luma):
 *   Y  = 0.299 in(.,0) + 0.587 in(.,1) + 0.114 in(.,2)         (u8 -> float, left to right)
 *   Cr = (R - Y) * 0.713 + 128;   Cb = (B - Y) * 0.564 + 128
 *   hist(b)  = #{(x, y) in [0,W) x [0,H) : int(clamp(Y, 0, 255)) == b}      (integer counts: order-free, exact)
 *   cdf(b)   = hist(0) + ... + hist(b)
 *   eq       = clamp(float(cdf(u8(clamp(Y, 0, 255)))) * (255.0f / float(H * W)), 0, 255)
 *   red   = u8(clamp(eq + (Cr - 128) * 1.4, 0, 255))
 *   green = u8(clamp((eq - 0.343 (Cb - 128)) - 0.711 (Cr - 128), 0, 255))
 *   blue  = u8(clamp(eq + 1.765 (Cb - 128), 0, 255))
 * The histogram runs over the WHOLE input (RDom (0, input.width()) x (0, input.height()), :28-36), whose min must be
 * 0; the output region may be any part of it.  Planar u8: in[c*in_sc + y*in_sy + x].  One rounding per operator (canon 0);
 * canon 1 (oracle_common.h) contracts every product above with the add / subtract it feeds (Y included: the bins move with it).
 */
#include "oracle_common.h"

int oracle_hist(const uint8_t *in, int W, int H, long in_sy, long in_sc, uint8_t *out, int ox0, int oy0, int ow, int oh,
                long out_sy, long out_sc, int32_t *cdf_out) {
    if (W < 1 || H < 1 || ow < 0 || oh < 0 || ox0 < 0 || oy0 < 0 || ox0 + ow > W || oy0 + oh > H) return -1;
    int32_t hist[256], cdf[256];
    memset(hist, 0, sizeof hist);
#define LUMA(x, y) o_mad(0.114f, (float)in[2 * in_sc + (long)(y) * in_sy + (x)], \
                         o_mad2(0.299f, (float)in[(long)(y) * in_sy + (x)], 0.587f, (float)in[in_sc + (long)(y) * in_sy + (x)]))
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) hist[(int)o_clampf(LUMA(x, y), 0.0f, 255.0f)]++;
    cdf[0] = hist[0];
    for (int b = 1; b < 256; b++) cdf[b] = cdf[b - 1] + hist[b];
    if (cdf_out) memcpy(cdf_out, cdf, sizeof cdf);
    const float scale = 255.0f / (float)(H * W);
#pragma omp parallel for schedule(static)
    for (int y = 0; y < oh; y++) {
        for (int x = 0; x < ow; x++) {
            const int X = ox0 + x, Yc = oy0 + y;
            const float Y = LUMA(X, Yc);
            const float R = (float)in[(long)Yc * in_sy + X], B = (float)in[2 * in_sc + (long)Yc * in_sy + X];
            const float Cr = o_mad(R - Y, 0.713f, 128.0f), Cb = o_mad(B - Y, 0.564f, 128.0f);
            const uint8_t bin = (uint8_t)o_clampf(Y, 0.0f, 255.0f);
            const float eq = o_clampf((float)cdf[bin] * scale, 0.0f, 255.0f);
            const float red = o_mad(Cr - 128.0f, 1.4f, eq);
            const float green = o_msub(o_msub(eq, 0.343f, Cb - 128.0f), 0.711f, Cr - 128.0f);
            const float blue = o_mad(1.765f, Cb - 128.0f, eq);
            out[(long)y * out_sy + x] = (uint8_t)o_clampf(red, 0.0f, 255.0f);
            out[out_sc + (long)y * out_sy + x] = (uint8_t)o_clampf(green, 0.0f, 255.0f);
            out[2 * out_sc + (long)y * out_sy + x] = (uint8_t)o_clampf(blue, 0.0f, 255.0f);
        }
    }
#undef LUMA
    return 0;
}
