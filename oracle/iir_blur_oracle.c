/* iir_blur_oracle.c — TEST INFRASTRUCTURE ONLY (see oracle_common.h).
 *
 * CPU restatement of /root/reference/apps/iir_blur/iir_blur_generator.cpp:13-31, 146-156 (first-order IIR low pass,
 * down and up every column, then the same along the rows — written in the reference as two "blur columns + transpose"):
 *   blur(x, 0)        = in(x, 0)
 *   blur(x, y)        = (1 - alpha) blur(x, y-1) + alpha in(x, y)          y = 1 .. H-1        (:22-24)
 *   blur(x, y)        = (1 - alpha) blur(x, y+1) + alpha blur(x, y)        y = H-2 .. 0        (:26-28)
 *   transpose(x, y)   = blur(y, x);  output = the same applied to transpose (height := the input's width)
 * The scans are sequential by definition; one rounding per operator (two products, one sum per step); canon 1
 * (oracle_common.h): product + product, the first — (1 - alpha) blur — is the one fused: fma(1 - alpha, blur, alpha in).
 * The reference pins the shape to 1536 x 2560 x 3 (:158-163); the restatement takes any W, H, C.
 * Planar f32: in[c*in_sc + y*in_sy + x], out likewise.  PARITY UNPINNED.
 */
#include "oracle_common.h"

/* columns of src [h rows of w] -> dst transposed [w rows of h] */
static void iir_cols_T(const float *src, long s_sy, int w, int h, float alpha, float *dst, long d_sy) {
    const float c1 = 1.0f - alpha;
#pragma omp parallel for schedule(static)
    for (int x = 0; x < w; x++) {
        float *b = (float *)malloc(sizeof(float) * h);
        b[0] = src[x];
        for (int y = 1; y < h; y++) b[y] = o_mad2(c1, b[y - 1], alpha, src[(long)y * s_sy + x]);
        for (int y = h - 2; y >= 0; y--) b[y] = o_mad2(c1, b[y + 1], alpha, b[y]);
        for (int y = 0; y < h; y++) dst[(long)x * d_sy + y] = b[y];
        free(b);
    }
}

int oracle_iir_blur(const float *in, int W, int H, int C, long in_sy, long in_sc, float alpha, float *out, long out_sy, long out_sc) {
    if (W < 1 || H < 1 || C < 1) return -1;
    float *t = (float *)malloc(sizeof(float) * (size_t)W * H);
    for (int c = 0; c < C; c++) {
        iir_cols_T(in + (long)c * in_sc, in_sy, W, H, alpha, t, H);        /* t: W rows of H */
        iir_cols_T(t, H, H, W, alpha, out + (long)c * out_sc, out_sy);     /* out: H rows of W */
    }
    free(t);
    return 0;
}
