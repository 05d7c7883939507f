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

/* bf16 variant (checker of `conv_layer_bf16`, BASELINE.json configs[4]): the operands are rounded to bfloat16
 * (round to nearest even — what v_cvt_pk_bf16_f32 does), products of two bf16 values are exact in binary32 and
 * even more so in double; the sum is taken in DOUBLE from the f32 bias and rounded once.  The matrix cores
 * accumulate in f32 in a hardware-defined order, so parity is by tolerance: the caller also receives
 * mag = |bias| + sum |products| per output to scale it.  The reference has no bf16 path: PARITY UNPINNED. */
static float oracle_bf16_round(float v) {
    uint32_t u;
    memcpy(&u, &v, 4);
    if ((u & 0x7f800000u) == 0x7f800000u) return v; /* inf / nan unchanged */
    u += 0x7fffu + ((u >> 16) & 1u);
    u &= 0xffff0000u;
    memcpy(&v, &u, 4);
    return v;
}

int oracle_conv_layer_bf16(const float *input, const float *filter, const float *bias, float *relu, float *mag, int CI,
                           int CO, int W, int H, int N) {
    if (CI < 1 || CO < 1 || W < 1 || H < 1 || N < 1) return -1;
    const size_t in_sx = CI, in_sy = (size_t)CI * (W + 2), in_sn = in_sy * (H + 2);
    const size_t f_skx = CO, f_sky = (size_t)CO * 3, f_sci = (size_t)CO * 9;
    const size_t nin = in_sn * N, nf = (size_t)CO * 9 * CI;
    float *in16 = (float *)malloc(sizeof(float) * nin), *f16 = (float *)malloc(sizeof(float) * nf);
    for (size_t i = 0; i < nin; i++) in16[i] = oracle_bf16_round(input[i]);
    for (size_t i = 0; i < nf; i++) f16[i] = oracle_bf16_round(filter[i]);
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; n++) {
        for (int y = 0; y < H; y++) {
            double *acc = (double *)malloc(sizeof(double) * CO), *am = (double *)malloc(sizeof(double) * CO);
            for (int x = 0; x < W; x++) {
                for (int c = 0; c < CO; c++) acc[c] = bias[c], am[c] = fabs(bias[c]);
                for (int ky = 0; ky < 3; ky++) {
                    for (int kx = 0; kx < 3; kx++) {
                        const float *ip = in16 + n * in_sn + (y + ky) * in_sy + (x + kx) * in_sx;
                        const float *fp = f16 + kx * f_skx + ky * f_sky;
                        for (int ci = 0; ci < CI; ci++) {
                            const double v = ip[ci];
                            const float *f = fp + ci * f_sci;
                            for (int c = 0; c < CO; c++) {
                                const double pr = (double)f[c] * v;
                                acc[c] += pr;
                                am[c] += fabs(pr);
                            }
                        }
                    }
                }
                const size_t o = (((size_t)n * H + y) * W + x) * CO;
                for (int c = 0; c < CO; c++) {
                    relu[o + c] = acc[c] > 0.0 ? (float)acc[c] : 0.0f;
                    if (mag) mag[o + c] = (float)am[c];
                }
            }
            free(acc);
            free(am);
        }
    }
    free(in16);
    free(f16);
    return 0;
}
