/* conv_layer_oracle.c — TEST INFRASTRUCTURE ONLY (see oracle_common.h).
 *
 * CPU restatement of /root/reference/apps/conv_layer/conv_layer_generator.cpp:21-27:
 *   conv(c,x,y,n)  = bias(c)
 *   conv(c,x,y,n) += filter(c, r.y, r.z, r.x) * input(r.x, x + r.y, y + r.z, n)   RDom r(0,CI, 0,3, 0,3)
 *   relu(c,x,y,n)  = max(0, conv(c,x,y,n))
 * Accumulation order: RDom lexicographic, r.x (= ci) fastest, then r.y (= kx), then r.z (= ky), starting from
 * bias.  PARITY UNPINNED (no golden output in the reference; process.cpp only prints timings).  Canonical
 * rounding defined here: each update is ONE fused multiply-add, acc = fma(filter, input, acc) — the reference's
 * own CPU schedule is an FMA kernel ("94.6% of peak" of AVX-512 FMA units, :123-133; LLVM contracts the
 * mul+add, src/CodeGen_Internal.cpp:614) — so an fmaf chain in RDom order is the faithful scalar model.
 * Layouts as pinned by the generator (:35-50), generalised to any N, W, H, CI, CO:
 *   input [CI, W+2, H+2, N], filter [CO, 3, 3, CI], bias [CO], relu [CO, W, H, N]; dimension 0 innermost.
 */
#include "oracle_common.h"

int oracle_conv_layer(const float *input, const float *filter, const float *bias, float *relu, int CI, int CO, int W,
                      int H, int N) {
    if (CI < 1 || CO < 1 || W < 1 || H < 1 || N < 1) return -1;
    const size_t in_sx = CI, in_sy = (size_t)CI * (W + 2), in_sn = in_sy * (H + 2);
    const size_t f_skx = CO, f_sky = (size_t)CO * 3, f_sci = (size_t)CO * 9;
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; n++) {
        for (int y = 0; y < H; y++) {
            float *acc = (float *)malloc(sizeof(float) * CO);
            for (int x = 0; x < W; x++) {
                for (int c = 0; c < CO; c++) acc[c] = bias[c];
                for (int ky = 0; ky < 3; ky++) {
                    for (int kx = 0; kx < 3; kx++) {
                        const float *ip = input + n * in_sn + (y + ky) * in_sy + (x + kx) * in_sx;
                        const float *fp = filter + kx * f_skx + ky * f_sky;
                        for (int ci = 0; ci < CI; ci++) {
                            const float v = ip[ci];
                            const float *f = fp + ci * f_sci;
                            for (int c = 0; c < CO; c++) acc[c] = fmaf(f[c], v, acc[c]);
                        }
                    }
                }
                float *o = relu + (((size_t)n * H + y) * W + x) * CO;
                for (int c = 0; c < CO; c++) o[c] = acc[c] > 0.0f ? acc[c] : 0.0f;
            }
            free(acc);
        }
    }
    return 0;
}
