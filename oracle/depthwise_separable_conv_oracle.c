/* depthwise_separable_conv_oracle.c — TEST INFRASTRUCTURE ONLY (see oracle_common.h).
 *
 * CPU restatement of /root/reference/apps/depthwise_separable_conv/depthwise_separable_conv_generator.cpp:24-75:
 *   input_bounded(d,x,y,b)         = (0 <= x < W && 0 <= y < H) ? input(d,x,y,b) : 0                        (:36-43)
 *   depthwise_convolved(d,x,y,b)  += depthwise_filter(rd, d, rx, ry) * input_bounded(d / CM, x+rx-pw, y+ry-ph, b)
 *                                    RDom (rd in [0,CM), rx in [0,FW), ry in [0,FH)), rd fastest, from 0    (:45-62)
 *   pointwise_convolved(d,x,y,b)   = bias(d);  += pointwise_filter(d, rc) * depthwise_convolved(rc,x,y,b)   (:64-72)
 *   output(d,x,y,b)                = max(pointwise_convolved, 0)                                            (:75)
 * with CM = depthwise_filter.dim(0).extent, pw = FW / 2, ph = FH / 2, rc over the IC intermediate channels.
 * As written, depthwise_filter is indexed by the INTERMEDIATE channel d in its second dimension, so that
 * dimension must cover [0, IC) (for CM == 1, the only case the reference's driver uses, IC == CI).
 * PARITY UNPINNED (process.cpp only prints timings).  Canonical rounding, as for conv_layer: every update is
 * one fused multiply-add in RDom order (the reference's schedules are FMA kernels, LLVM contracts the update).
 * Layouts (dimension 0 innermost): input [CI, W, H, N], depthwise_filter [CM, IC', FW, FH] with stride(1) == CM
 * (:283), pointwise_filter [CO, IC], bias [CO], output [CO, W, H, N]; all dense here.
 */
#include "oracle_common.h"

int oracle_depthwise_separable_conv(const float *input, const float *dw, const float *pw, const float *bias, float *output,
                                    int CI, int W, int H, int N, int CM, int FW, int FH, int IC, int CO) {
    if (CI < 1 || W < 1 || H < 1 || N < 1 || CM < 1 || FW < 1 || FH < 1 || IC < 1 || CO < 1) return -1;
    const int padw = FW / 2, padh = FH / 2;
    const size_t in_sx = CI, in_sy = (size_t)CI * W, in_sn = in_sy * H;
    /* depthwise_filter [CM, F1, FW, FH] dense with F1 >= IC: strides 1, CM, CM*F1, CM*F1*FW; the caller passes F1 = IC */
    const size_t d_s1 = CM, d_sx = (size_t)CM * IC, d_sy = d_sx * FW;
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; n++) {
        for (int y = 0; y < H; y++) {
            float *mid = (float *)malloc(sizeof(float) * IC);
            for (int x = 0; x < W; x++) {
                for (int d = 0; d < IC; d++) {
                    float acc = 0.0f;
                    for (int ry = 0; ry < FH; ry++) {
                        for (int rx = 0; rx < FW; rx++) {
                            const int xx = x + rx - padw, yy = y + ry - padh;
                            const int inb = xx >= 0 && xx < W && yy >= 0 && yy < H;
                            for (int rd = 0; rd < CM; rd++) {
                                const float v = inb ? input[n * in_sn + yy * in_sy + xx * in_sx + d / CM] : 0.0f;
                                acc = fmaf(dw[rd + d * d_s1 + rx * d_sx + ry * d_sy], v, acc);
                            }
                        }
                    }
                    mid[d] = acc;
                }
                float *o = output + (((size_t)n * H + y) * W + x) * CO;
                for (int c = 0; c < CO; c++) {
                    float acc = bias[c];
                    for (int rc = 0; rc < IC; rc++) acc = fmaf(pw[c + (size_t)rc * CO], mid[rc], acc);
                    o[c] = acc > 0.0f ? acc : 0.0f;
                }
            }
            free(mid);
        }
    }
    return 0;
}
