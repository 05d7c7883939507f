/* interpolate_oracle.c — TEST INFRASTRUCTURE ONLY (see oracle_common.h).
 *
 * CPU restatement of /root/reference/apps/interpolate/interpolate_generator.cpp:20-77 (levels = 10, :15):
 *   clamped            = repeat_edge(input)                          input f32 [W,H,4], channel 3 = alpha
 *   downsampled[0]     = (c < 3) ? clamped(.,c) * clamped(.,3) : clamped(.,3)
 *   l = 1..9:  prev = downsampled[l-1]; for l == 4 prev is read at (clamp(x, 0, W/8), clamp(y, 0, H/8))   (:40-49)
 *              downx[l](x,y)       = (prev(2x-1,y) + 2 prev(2x,y) + prev(2x+1,y)) * 0.25
 *              downsampled[l](x,y) = (downx[l](x,2y-1) + 2 downx[l](x,2y) + downx[l](x,2y+1)) * 0.25
 *   interpolated[9]    = downsampled[9]
 *   l = 8..0:  upsampledx[l](x,y) = (interpolated[l+1](x/2, y) + interpolated[l+1]((x+1)/2, y)) / 2
 *              upsampled[l](x,y)  = (upsampledx[l](x, y/2) + upsampledx[l](x, (y+1)/2)) / 2          (floor division)
 *              interpolated[l]    = downsampled[l] + (1 - downsampled[l](.,3)) * upsampled[l]
 *   output(x,y,c)      = interpolated[0](x,y,c) / interpolated[0](x,y,3),  c < 3, over exactly [0,W) x [0,H)   (:83-87)
 * Sums left to right as written, one rounding per operator; "/ 2.0f" == "* 0.5f".  PARITY UNPINNED.
 * Canon 1 (oracle_common.h): 2 prev is exact, so the taps are the same in both forms; interpolated[l] = fma(alpha, upsampled, downsampled[l])
 * for l >= 1 (downsampled[l] is stored: compute_root, :150-160); at l = 0 downsampled[0] is INLINE in `normalize` (:178-188), so for a
 * colour channel the sum is product + product and the first, clamped(c) * clamped(3), is the one fused: fma(in_c, in_3, alpha * up).
 * Every Func is a total function on Z^2; level l is evaluated on the box the levels above and below it read (same
 * recursion as the kernels: I_l for interpolated, D_l for downsampled), stored as float[4] per pixel.
 */
#include "oracle_common.h"

#define IL 10
typedef struct { int x0, x1, y0, y1; } ibox_t;
typedef struct { ibox_t b; int w, h; float *v; } ilevel_t;   /* v[((y - y0) * w + (x - x0)) * 4 + c] */

static void ilevel_alloc(ilevel_t *L, ibox_t b) {
    L->b = b, L->w = b.x1 - b.x0 + 1, L->h = b.y1 - b.y0 + 1;
    L->v = (float *)malloc(sizeof(float) * 4 * (size_t)L->w * L->h);
}
static inline const float *ilevel_at(const ilevel_t *L, int x, int y) { return L->v + ((size_t)(y - L->b.y0) * L->w + (x - L->b.x0)) * 4; }

void oracle_interpolate_boxes(int W, int H, int *out40, int *out40d) {  /* I_l and D_l, 4 ints per level (x0 x1 y0 y1) */
    ibox_t I[IL], D[IL];
    I[0].x0 = 0, I[0].x1 = W - 1, I[0].y0 = 0, I[0].y1 = H - 1;
    for (int l = 1; l < IL; l++) {
        I[l].x0 = 0, I[l].y0 = 0, I[l].x1 = o_fdiv(I[l - 1].x1 + 1, 2), I[l].y1 = o_fdiv(I[l - 1].y1 + 1, 2);
    }
    D[IL - 1] = I[IL - 1];
    for (int l = IL - 2; l >= 0; l--) {
        ibox_t n = {2 * D[l + 1].x0 - 1, 2 * D[l + 1].x1 + 1, 2 * D[l + 1].y0 - 1, 2 * D[l + 1].y1 + 1};
        if (l + 1 == 4) {
            const int w = W / 8, h = H / 8;
            n.x0 = o_clampi(n.x0, 0, w), n.x1 = o_clampi(n.x1, 0, w), n.y0 = o_clampi(n.y0, 0, h), n.y1 = o_clampi(n.y1, 0, h);
        }
        D[l].x0 = n.x0 < I[l].x0 ? n.x0 : I[l].x0, D[l].x1 = n.x1 > I[l].x1 ? n.x1 : I[l].x1;
        D[l].y0 = n.y0 < I[l].y0 ? n.y0 : I[l].y0, D[l].y1 = n.y1 > I[l].y1 ? n.y1 : I[l].y1;
    }
    for (int l = 0; l < IL; l++) {
        out40[4 * l] = I[l].x0, out40[4 * l + 1] = I[l].x1, out40[4 * l + 2] = I[l].y0, out40[4 * l + 3] = I[l].y1;
        out40d[4 * l] = D[l].x0, out40d[4 * l + 1] = D[l].x1, out40d[4 * l + 2] = D[l].y0, out40d[4 * l + 3] = D[l].y1;
    }
}

int oracle_interpolate(const float *in, int W, int H, long in_sy, long in_sc, float *out, long out_sy, long out_sc) {
    if (W < 1 || H < 1) return -1;
    int bi[4 * IL], bd[4 * IL];
    oracle_interpolate_boxes(W, H, bi, bd);
    ilevel_t ds[IL], ip[IL];
    memset(ds, 0, sizeof ds), memset(ip, 0, sizeof ip);
#define DS0(X, Y, o)                                                                                              \
    do {                                                                                                          \
        const long off__ = (long)o_clampi((Y), 0, H - 1) * in_sy + o_clampi((X), 0, W - 1);                       \
        const float a__ = in[3 * in_sc + off__];                                                                  \
        (o)[0] = in[off__] * a__, (o)[1] = in[in_sc + off__] * a__, (o)[2] = in[2 * in_sc + off__] * a__, (o)[3] = a__; \
    } while (0)
    for (int l = 1; l < IL; l++) {
        ibox_t b = {bd[4 * l], bd[4 * l + 1], bd[4 * l + 2], bd[4 * l + 3]};
        ilevel_alloc(&ds[l], b);
        const int cw = W / 8, ch = H / 8;
#pragma omp parallel for schedule(static)
        for (int y = b.y0; y <= b.y1; y++) {
            for (int x = b.x0; x <= b.x1; x++) {
                float dx[3][4];
                for (int j = 0; j < 3; j++) {
                    float p[3][4];
                    for (int i = 0; i < 3; i++) {
                        int X = 2 * x - 1 + i, Y = 2 * y - 1 + j;
                        if (l == 4) X = o_clampi(X, 0, cw), Y = o_clampi(Y, 0, ch);
                        if (l == 1) DS0(X, Y, p[i]);
                        else memcpy(p[i], ilevel_at(&ds[l - 1], X, Y), 16);
                    }
                    for (int c = 0; c < 4; c++) dx[j][c] = ((p[0][c] + 2.0f * p[1][c]) + p[2][c]) * 0.25f;
                }
                float *o = (float *)ilevel_at(&ds[l], x, y);
                for (int c = 0; c < 4; c++) o[c] = ((dx[0][c] + 2.0f * dx[1][c]) + dx[2][c]) * 0.25f;
            }
        }
    }
    for (int l = IL - 1; l >= 0; l--) {
        ibox_t b = {bi[4 * l], bi[4 * l + 1], bi[4 * l + 2], bi[4 * l + 3]};
        ilevel_alloc(&ip[l], b);
#pragma omp parallel for schedule(static)
        for (int y = b.y0; y <= b.y1; y++) {
            for (int x = b.x0; x <= b.x1; x++) {
                float *o = (float *)ilevel_at(&ip[l], x, y);
                if (l == IL - 1) {
                    memcpy(o, ilevel_at(&ds[l], x, y), 16);
                    continue;
                }
                float d[4], a0[4] = {1.0f, 1.0f, 1.0f, 1.0f};   /* l == 0: d[c] = a0[c] * d3 with the product still open */
                if (l == 0) {
                    const long off = (long)y * in_sy + x;   /* I_0 is the image itself */
                    d[3] = in[3 * in_sc + off], a0[0] = in[off], a0[1] = in[in_sc + off], a0[2] = in[2 * in_sc + off];
                } else {
                    memcpy(d, ilevel_at(&ds[l], x, y), 16);
                }
                const int xa = o_fdiv(x, 2), xb = o_fdiv(x + 1, 2), ya = o_fdiv(y, 2), yb = o_fdiv(y + 1, 2);
                const float alpha = 1.0f - d[3];
                for (int c = 0; c < 4; c++) {
                    const float ua = (ilevel_at(&ip[l + 1], xa, ya)[c] + ilevel_at(&ip[l + 1], xb, ya)[c]) * 0.5f;
                    const float ub = (ilevel_at(&ip[l + 1], xa, yb)[c] + ilevel_at(&ip[l + 1], xb, yb)[c]) * 0.5f;
                    const float up = (ua + ub) * 0.5f;
                    o[c] = (l == 0 && c < 3) ? o_mad2(a0[c], d[3], alpha, up) : o_mad(alpha, up, d[c]);
                }
            }
        }
    }
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            const float *v = ilevel_at(&ip[0], x, y);
            for (int c = 0; c < 3; c++) out[(long)c * out_sc + (long)y * out_sy + x] = v[c] / v[3];
        }
    for (int l = 0; l < IL; l++) free(ds[l].v), free(ip[l].v);
#undef DS0
    return 0;
}
