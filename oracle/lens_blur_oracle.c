/* lens_blur_oracle.c — TEST INFRASTRUCTURE ONLY (see oracle_common.h).
 *
 * CPU restatement of apps/lens_blur/lens_blur_generator.cpp:24-152 ("THE ALGORITHM"; downsample :279-285,
 * upsample :288-294) of /root/reference.  PARITY UNPINNED, and in one respect BY ASSUMPTION:
 *
 *   * the float stages (cost confidence, the push-pull pyramids, filtered_cost) have no golden output in the
 *     reference and the compiler cannot be built here; the canonical order is the generator's expression order
 *     (oracle_common.h); canon 1 there contracts sum(cost cost) (each square with the running sum), the confidence
 *     sa / slices - sb sb = fma(-sb, sb, sa / slices), and the 3 (b + c) of the down-sampling taps with the add it feeds.  The
 *     up-sampling weights 1/4, 3/4 and the lerp weight 1/2 make one exact product per sum (identical in both forms), the bokeh
 *     weights are 0 or 1, the colour costs are integers below 2^18;
 *   * the sample positions come from Halide's random_float() (:116-117), which is a FIXED hash of
 *     (call id, definition tag, free variables) — src/Random.cpp:20-104, src/IROperator.cpp:2873-2889,
 *     src/Function.cpp:640-648 — but the call ids and the tag are values of two process-global counters of the
 *     COMPILER at the moment the generator defines `sample_locations`: the ids count random_*() calls made so far
 *     (here: 0 for sample_u, 1 for sample_v — the first two calls of the generator process), the tag counts pure
 *     Func definitions made so far (every `f(x, ..) = ..`, including the ones inside BoundaryConditions::repeat_edge,
 *     downsample / upsample and the inline reductions sum / argmin / maximum).  Counting the definitions of
 *     generate() in order gives LB_DEFAULT_TAG = 71; the count cannot be verified without running the compiler, so
 *     the tag is a PARAMETER of this oracle and of the GPU implementation (hlmi_lens_blur_set_random_tag), and the
 *     statement "the oracle equals the reference" is conditional on it.  Everything else is independent of it.
 *
 * Every Func is a total function on Z^n; only left_im / right_im and cost_pyramid_push[1..7] are edge-clamped
 * (:27-28, :62).  Each stage is therefore evaluated on the box its consumers read:
 *   depth, bokeh radius           D   = image grown by R = maximum_blur_radius          (max filters :98-99, samples :127)
 *   cost_pyramid_pull[i]          P_0 = D,  P_{i+1} = [fdiv(lo_i, 2) - 1, fdiv(hi_i, 2) + 1]   (upsample :291-292)
 *   cost_pyramid_push[i], i >= 1  [0, max(w_i, 1) - 1] x [0, max(h_i, 1) - 1], w_i = w_{i-1} / 2         (:58-62)
 *   cost_pyramid_push[0]          E   = P_0 united with [-1, 2 max(w_1, 1)]                    (downsample :282-283)
 */
#include "oracle_common.h"

#define LB_LEVELS 8
#define LB_DEFAULT_TAG 71

typedef struct {
    int x0, x1, y0, y1; /* inclusive */
} lb_box;
static inline int lb_w(lb_box b) { return b.x1 - b.x0 + 1; }
static inline int lb_h(lb_box b) { return b.y1 - b.y0 + 1; }

/* src/Random.cpp:20-63: the quadratic permutation of a u32 */
static inline uint32_t lb_rng32(uint32_t x) { return ((1040796640u * x) + 1121052041u) * x + 576942909u; }
/* random_float(e) with e = {id, tag, z, y, x} (src/Random.cpp:66-104; LowerRandom appends the free variables reversed,
 * then the tag is first of the appended: args = {id} + reverse({x, y, z, tag})) */
float oracle_lens_blur_random(int id, int tag, int z, int y, int x) {
    uint32_t r = lb_rng32((uint32_t)id);
    r = lb_rng32(r + (uint32_t)tag);
    r = lb_rng32(r + (uint32_t)z);
    r = lb_rng32(r + (uint32_t)y);
    r = lb_rng32(r + (uint32_t)x);
    r = r ^ (r >> 16);
    const float f = o_bits2f((127u << 23) | (r >> 9)) - 1.0f;
    return o_clampf(f, 0.0f, 1.0f);
}

int oracle_lens_blur_default_tag(void) { return LB_DEFAULT_TAG; }

/* left: u8 [3][H][W] planar, right: u8 [3][RH][RW]; out: f32 [3][H][W].  depth_out (optional): int32 [H + 2R][W + 2R]
 * = depth on D, for stage-wise comparison.  Returns 0, or -1 on allocation failure. */
int oracle_lens_blur(const uint8_t *left, int W, int H, const uint8_t *right, int RW, int RH, int slices, int focus_depth,
                     float blur_radius_scale, int aperture_samples, int tag, float *out, int32_t *depth_out) {
    if (W <= 0 || H <= 0) return 0;
    const int md = slices - focus_depth > focus_depth ? slices - focus_depth : focus_depth;
    const int R = (int)((float)md * blur_radius_scale); /* :25-26 */
    const float fslices = (float)slices;

    /* ---- boxes */
    lb_box D = {-R, W - 1 + R, -R, H - 1 + R};
    lb_box P[LB_LEVELS];
    P[0] = D;
    for (int i = 1; i < LB_LEVELS; i++) {
        P[i].x0 = o_fdiv(P[i - 1].x0, 2) - 1, P[i].x1 = o_fdiv(P[i - 1].x1, 2) + 1;
        P[i].y0 = o_fdiv(P[i - 1].y0, 2) - 1, P[i].y1 = o_fdiv(P[i - 1].y1, 2) + 1;
    }
    int w[LB_LEVELS], h[LB_LEVELS], we[LB_LEVELS], he[LB_LEVELS];
    w[0] = W, h[0] = H;
    for (int i = 1; i < LB_LEVELS; i++) w[i] = w[i - 1] / 2, h[i] = h[i - 1] / 2; /* :58-61 */
    for (int i = 0; i < LB_LEVELS; i++) we[i] = w[i] > 1 ? w[i] : 1, he[i] = h[i] > 1 ? h[i] : 1;
    lb_box E = P[0];
    if (E.x0 > -1) E.x0 = -1;
    if (E.y0 > -1) E.y0 = -1;
    if (E.x1 < 2 * we[1]) E.x1 = 2 * we[1];
    if (E.y1 < 2 * he[1]) E.y1 = 2 * he[1];

    /* ---- storage: push[i][z][c][y][x], pull[i] likewise */
    float *push[LB_LEVELS], *pull[LB_LEVELS];
    lb_box pb[LB_LEVELS]; /* boxes of push levels */
    pb[0] = E;
    for (int i = 1; i < LB_LEVELS; i++) pb[i] = (lb_box){0, we[i] - 1, 0, he[i] - 1};
    for (int i = 0; i < LB_LEVELS; i++) push[i] = pull[i] = NULL;
    int rc = -1;
    for (int i = 0; i < LB_LEVELS; i++) {
        push[i] = (float *)malloc(sizeof(float) * 2 * (size_t)slices * lb_w(pb[i]) * lb_h(pb[i]));
        if (!push[i]) goto done;
        if (i >= 1) {
            pull[i] = (float *)malloc(sizeof(float) * 2 * (size_t)slices * lb_w(P[i]) * lb_h(P[i]));
            if (!pull[i]) goto done;
        }
    }
    int32_t *depth = (int32_t *)malloc(sizeof(int32_t) * (size_t)lb_w(D) * lb_h(D));
    float *br = (float *)malloc(sizeof(float) * (size_t)lb_w(D) * lb_h(D));
    float *wcy = (float *)malloc(sizeof(float) * (size_t)lb_w(D) * H);
    if (!depth || !br || !wcy) {
        free(depth), free(br), free(wcy);
        goto done;
    }
#define PUSH(i, x, y, z, c) push[i][(((size_t)(z) * 2 + (c)) * lb_h(pb[i]) + ((y) - pb[i].y0)) * lb_w(pb[i]) + ((x) - pb[i].x0)]
#define PULL(i, x, y, z, c) pull[i][(((size_t)(z) * 2 + (c)) * lb_h(P[i]) + ((y) - P[i].y0)) * lb_w(P[i]) + ((x) - P[i].x0)]
    /* push level i as the total function the generator defines: clamped for i >= 1 (:62), direct for i = 0 */
#define PUSHF(i, x, y, z, c) ((i) == 0 ? PUSH(0, x, y, z, c) : PUSH(i, o_clampi(x, 0, we[i] - 1), o_clampi(y, 0, he[i] - 1), z, c))

    /* ---- cost, cost_confidence, cost_pyramid_push[0] on E (:30-56) */
#pragma omp parallel for schedule(static)
    for (int y = E.y0; y <= E.y1; y++) {
        float cost[64];
        for (int x = E.x0; x <= E.x1; x++) {
            const int lx = o_clampi(x, 0, W - 1), ly = o_clampi(y, 0, H - 1), ry = o_clampi(y, 0, RH - 1);
            float sa = 0.0f, sb = 0.0f;
            for (int z = 0; z < slices; z++) {
                float cz = 0.0f;
                for (int c = 0; c < 3; c++) {
                    const int l = left[((size_t)c * H + ly) * W + lx];
                    const int r0 = right[((size_t)c * RH + ry) * RW + o_clampi(x + 2 * z, 0, RW - 1)];
                    const int r1 = right[((size_t)c * RH + ry) * RW + o_clampi(x + 2 * z + 1, 0, RW - 1)];
                    const int d0 = l > r0 ? l - r0 : r0 - l, d1 = l > r1 ? l - r1 : r1 - l; /* absd :32-33 */
                    const float d = (float)(d0 < d1 ? d0 : d1);
                    cz = (c == 0) ? d * d : o_mad(d, d, cz); /* pow(., 2) = e * e (src/IROperator.cpp:1008-1027); integers below 2^18: exact either way */
                }
                cost[z] = cz;
                sa = o_mad(cz, cz, sa); /* sum(pow(cost, 2))  :44 */
                sb = sb + cz / fslices; /* sum(cost / slices) :45 */
            }
            const float conf = o_msub(sa / fslices, sb, sb); /* :44-46 */
            for (int z = 0; z < slices; z++) {
                PUSH(0, x, y, z, 0) = cost[z] * conf; /* :53-54 */
                PUSH(0, x, y, z, 1) = conf;
            }
        }
    }
    /* ---- cost_pyramid_push[1..7] (:57-63): downx then downy (:282-283), then clamped to [0, w) x [0, h) */
    for (int i = 1; i < LB_LEVELS; i++) {
#pragma omp parallel for schedule(static) collapse(2)
        for (int zc = 0; zc < 2 * slices; zc++) {
            for (int y = 0; y < he[i]; y++) {
                const int z = zc >> 1, c = zc & 1;
                for (int x = 0; x < we[i]; x++) {
                    float dx[4];
                    for (int k = 0; k < 4; k++) {
                        const int yy = 2 * y - 1 + k;
                        dx[k] = (o_mad(3.0f, PUSHF(i - 1, 2 * x, yy, z, c) + PUSHF(i - 1, 2 * x + 1, yy, z, c), PUSHF(i - 1, 2 * x - 1, yy, z, c)) +
                                 PUSHF(i - 1, 2 * x + 2, yy, z, c)) * 0.125f;
                    }
                    PUSH(i, x, y, z, c) = (o_mad(3.0f, dx[1] + dx[2], dx[0]) + dx[3]) * 0.125f;
                }
            }
        }
    }
    /* ---- cost_pyramid_pull[7..1] (:65-71) on P_i */
    for (int i = LB_LEVELS - 1; i >= 1; i--) {
#pragma omp parallel for schedule(static) collapse(2)
        for (int zc = 0; zc < 2 * slices; zc++) {
            for (int y = P[i].y0; y <= P[i].y1; y++) {
                const int z = zc >> 1, c = zc & 1;
                for (int x = P[i].x0; x <= P[i].x1; x++) {
                    const float p = PUSHF(i, x, y, z, c);
                    if (i == LB_LEVELS - 1) {
                        PULL(i, x, y, z, c) = p;
                    } else {
                        const int xa = o_fdiv(x, 2) - 1 + 2 * o_fmod(x, 2), xb = o_fdiv(x, 2);
                        const int ya = o_fdiv(y, 2) - 1 + 2 * o_fmod(y, 2), yb = o_fdiv(y, 2);
                        const float ua = 0.25f * PULL(i + 1, xa, ya, z, c) + 0.75f * PULL(i + 1, xb, ya, z, c); /* upx at row ya :291 */
                        const float ub = 0.25f * PULL(i + 1, xa, yb, z, c) + 0.75f * PULL(i + 1, xb, yb, z, c);
                        const float up = 0.25f * ua + 0.75f * ub;                                                /* upy :292 */
                        PULL(i, x, y, z, c) = o_lerp(up, p, 0.5f);                                               /* :68-70 */
                    }
                }
            }
        }
    }
    /* ---- cost_pyramid_pull[0], filtered_cost, depth, bokeh radius on D (:68-86) */
#pragma omp parallel for schedule(static)
    for (int y = D.y0; y <= D.y1; y++) {
        for (int x = D.x0; x <= D.x1; x++) {
            const int xa = o_fdiv(x, 2) - 1 + 2 * o_fmod(x, 2), xb = o_fdiv(x, 2);
            const int ya = o_fdiv(y, 2) - 1 + 2 * o_fmod(y, 2), yb = o_fdiv(y, 2);
            int best_i = 0;
            float best = 3.402823466e38f; /* Float(32).max(), src/InlineReductions.cpp:307 */
            for (int z = 0; z < slices; z++) {
                float v[2];
                for (int c = 0; c < 2; c++) {
                    const float ua = 0.25f * PULL(1, xa, ya, z, c) + 0.75f * PULL(1, xb, ya, z, c);
                    const float ub = 0.25f * PULL(1, xa, yb, z, c) + 0.75f * PULL(1, xb, yb, z, c);
                    const float up = 0.25f * ua + 0.75f * ub;
                    v[c] = o_lerp(up, PUSH(0, x, y, z, c), 0.5f);
                }
                const float fc = v[0] / v[1]; /* :74-75 */
                if (fc < best) best = fc, best_i = z; /* argmin: strict <, first minimum (:311-312) */
            }
            const size_t o = (size_t)(y - D.y0) * lb_w(D) + (x - D.x0);
            depth[o] = best_i;
            const int ad = best_i - focus_depth;
            br[o] = (float)(uint32_t)(ad < 0 ? -ad : ad) * blur_radius_scale; /* :85-86 */
        }
    }
    if (depth_out) memcpy(depth_out, depth, sizeof(int32_t) * (size_t)lb_w(D) * lb_h(D));
#define BR(x, y) br[(size_t)((y) - D.y0) * lb_w(D) + ((x) - D.x0)]
#define DEPTH(x, y) depth[(size_t)((y) - D.y0) * lb_w(D) + ((x) - D.x0)]
    /* ---- worst_case_bokeh_radius (:94-100) */
#pragma omp parallel for schedule(static)
    for (int y = 0; y < H; y++) {
        for (int x = D.x0; x <= D.x1; x++) {
            float m = -INFINITY; /* maximum() starts from the type's minimum */
            for (int r = -R; r <= R; r++) m = BR(x, y + r) > m ? BR(x, y + r) : m;
            wcy[(size_t)y * lb_w(D) + (x - D.x0)] = m;
        }
    }
    /* ---- samples, weights, output (:102-150) */
#pragma omp parallel for schedule(static)
    for (int y = 0; y < H; y++) {
        for (int x = 0; x < W; x++) {
            float worst = -INFINITY;
            for (int r = -R; r <= R; r++) {
                const float v = wcy[(size_t)y * lb_w(D) + (x + r - D.x0)];
                worst = v > worst ? v : worst;
            }
            float acc[4];
            for (int c = 0; c < 3; c++) acc[c] = (float)left[((size_t)c * H + y) * W + x]; /* output = input_with_alpha :110-111 */
            acc[3] = 255.0f;
            const float brs = BR(x, y) * BR(x, y); /* pow(bokeh_radius, 2) :89 */
            const int dxy = DEPTH(x, y);
            for (int s = 0; s < aperture_samples; s++) {
                const float fu = ((oracle_lens_blur_random(0, tag, s, y, x) - 0.5f) * 2.0f) * worst; /* :116 */
                const float fv = ((oracle_lens_blur_random(1, tag, s, y, x) - 0.5f) * 2.0f) * worst; /* :117 */
                const int u = o_clampi((int)fu, -R, R), v = o_clampi((int)fv, -R, R);                  /* :118-119 */
                const int sx = x + u, sy = y + v;
                const float r2 = (float)(u * u + v * v);
                const float brs_s = BR(sx, sy) * BR(sx, sy);
                const int within_this = r2 < brs;                 /* :132-133 */
                const int this_within_sample = r2 < brs_s;        /* :135-136 */
                const int in_front = DEPTH(sx, sy) < dxy;         /* :138-139 */
                const float wgt = ((within_this || in_front) && this_within_sample) ? 1.0f : 0.0f; /* :142-146 */
                const int cx = o_clampi(sx, 0, W - 1), cy = o_clampi(sy, 0, H - 1);
                for (int c = 0; c < 3; c++) acc[c] = acc[c] + wgt * (float)left[((size_t)c * H + cy) * W + cx];
                acc[3] = acc[3] + wgt * 255.0f;
            }
            for (int c = 0; c < 3; c++) out[((size_t)c * H + y) * W + x] = acc[c] / acc[3]; /* :152 */
        }
    }
    free(depth), free(br), free(wcy);
    rc = 0;
done:
    for (int i = 0; i < LB_LEVELS; i++) free(push[i]), free(pull[i]);
    return rc;
}
