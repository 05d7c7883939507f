/* Exhaustive proof of the division shortcut in halide_amd/csrc/lens_blur.hip (cost_stack): for every integer numerator
 * a in [0, 3 * 255^2] (a stereo cost, lens_blur_generator.cpp:30-39) and every divisor n in [1, 64] (`slices`, :14) the
 * two-FMA correction of q0 = a * RN(1 / n) equals the correctly rounded a / n that the generator's expression evaluates.
 * Prints the number of mismatches (0) — tests/test_lens_blur.py builds and runs it.  Test infrastructure only. */
#include <math.h>
#include <stdio.h>
int main(void) {
    long bad = 0;
    for (int n = 1; n <= 64; n++) {
        volatile float fn = (float)n;
        const float r = 1.0f / fn;
        for (int a = 0; a <= 3 * 255 * 255; a++) {
            const float fa = (float)a, q0 = fa * r, e = fmaf(-fn, q0, fa), q = fmaf(e, r, q0);
            if (q != fa / fn) bad++;
        }
    }
    printf("%ld\n", bad);
    return bad != 0;
}
