/* blur_oracle.c — TEST INFRASTRUCTURE ONLY (see oracle_common.h).
 *
 * CPU restatement of /root/reference/apps/blur/halide_blur_generator.cpp:39-40:
 *   blur_x(x,y) = (input(x,y) + input(x+1,y) + input(x+2,y)) / 3
 *   blur_y(x,y) = (blur_x(x,y) + blur_x(x,y+1) + blur_x(x,y+2)) / 3
 * All arithmetic is uint16 (the int literal 3 is converted to the Expr's type, src/IROperator.cpp:1903-1915),
 * so sums WRAP mod 2^16 before the unsigned division (src/IR.h:29-47).  No boundary condition: the
 * input must cover [x, x+W+1] x [y, y+H+1].
 * PINNED by the reference's own scalar loop, apps/blur/test.cpp:18-33 (which computes in `int`, i.e.
 * agrees with this restatement whenever no 16-bit sum overflows — its inputs are `rand() & 0xfff`,
 * test.cpp:169) — tests/test_blur.py checks both, and oracle/_ref/blur_test runs that file unmodified. */
#include "oracle_common.h"

/* in: (H+2) rows of stride in_sy, at least W+2 valid elements each; out: H rows of stride out_sy. */
int oracle_blur(const uint16_t *in, int in_sy, int W, int H, uint16_t *out, int out_sy) {
    if (W < 0 || H < 0) return -1;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < H; y++) {
        for (int x = 0; x < W; x++) {
            uint16_t bx[3];
            for (int r = 0; r < 3; r++) {
                const uint16_t *p = in + (size_t)(y + r) * (size_t)in_sy + x;
                uint16_t s = (uint16_t)((uint16_t)(p[0] + p[1]) + p[2]);
                bx[r] = (uint16_t)(s / 3);
            }
            uint16_t s = (uint16_t)((uint16_t)(bx[0] + bx[1]) + bx[2]);
            out[(size_t)y * (size_t)out_sy + x] = (uint16_t)(s / 3);
        }
    }
    return 0;
}
