/* stencil_chain_oracle.c — TEST INFRASTRUCTURE ONLY (see oracle_common.h).
 *
 * CPU restatement of /root/reference/apps/stencil_chain/stencil_chain_generator.cpp:18-34:
 *   stage_0 = repeat_edge(input)
 *   stage_{s+1}(x,y) = sum_{i=-2..2} sum_{j=-2..2} u16((i+3)*(j+3)) * stage_s(x+i, y+j)      (u16, wraps)
 *   output = stage_{stencils}                      (stencils = 32, GeneratorParam :7)
 * Only stage 0 is clamped, so stage s is evaluated on the output region grown by 2*(stencils - s).
 * Pure integer (mod 2^16) arithmetic => the result does not depend on summation order: this oracle is
 * exact with respect to the reference by construction of the ring; no golden image exists in the
 * reference (its only test is "Success!", apps/stencil_chain/CMakeLists.txt) — parity otherwise unpinned.
 */
#include "oracle_common.h"

int oracle_stencil_chain(const uint16_t *in, int in_sy, int W, int H, int stencils, uint16_t *out, int out_sy) {
    if (W < 1 || H < 1 || stencils < 1) return -1;
    int g = 2 * stencils;
    int cw = W + 2 * g, ch = H + 2 * g;
    uint16_t *cur = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)cw * ch);
    uint16_t *nxt = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)cw * ch);
    /* stage 0 on [-g, W-1+g] x [-g, H-1+g] */
#pragma omp parallel for schedule(static)
    for (int y = 0; y < ch; y++) {
        int yc = o_clampi(y - g, 0, H - 1);
        for (int x = 0; x < cw; x++) {
            int xc = o_clampi(x - g, 0, W - 1);
            cur[(size_t)y * cw + x] = in[(size_t)yc * in_sy + xc];
        }
    }
    for (int s = 1; s <= stencils; s++) {
        int m = 2 * s; /* margin already consumed: stage s valid on [m, cw-1-m] in buffer coordinates */
#pragma omp parallel for schedule(static)
        for (int y = m; y < ch - m; y++) {
            for (int x = m; x < cw - m; x++) {
                uint16_t e = 0;
                for (int i = -2; i <= 2; i++) {
                    for (int j = -2; j <= 2; j++) {
                        e = (uint16_t)(e + (uint16_t)((uint16_t)((i + 3) * (j + 3)) * cur[(size_t)(y + j) * cw + (x + i)]));
                    }
                }
                nxt[(size_t)y * cw + x] = e;
            }
        }
        uint16_t *t = cur;
        cur = nxt;
        nxt = t;
    }
    for (int y = 0; y < H; y++) memcpy(out + (size_t)y * out_sy, cur + (size_t)(y + g) * cw + g, sizeof(uint16_t) * (size_t)W);
    free(cur);
    free(nxt);
    return 0;
}
