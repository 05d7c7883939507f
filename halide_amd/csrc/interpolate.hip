// interpolate.hip — gfx950 implementation of the reference's interpolate AOT pipeline (alpha-weighted pull-push
// pyramid, 10 levels; SURVEY.md §8 f3 — the adjacent app closest to local_laplacian).
// Algorithm: /root/reference/apps/interpolate/interpolate_generator.cpp:20-77; boundary: `int interpolate(
// halide_buffer_t *input, halide_buffer_t *output)`, f32 [W,H,4] planar in (:23 pins 4 channels), f32 [W,H,3] out over
// exactly the input's [0,W) x [0,H) (:83-87).
//
// Every Func is a total function on Z^2 (only the input is edge-clamped, plus the coordinate clamp the generator adds
// in front of level 4, :40-49), so level l is materialised on the box its readers touch:
//   I_0 = [0, W-1],  I_{l+1} = [floor(lo/2), floor((hi+1)/2)]                 interpolated[l]  (read by the level below)
//   D_9 = I_9,  D_l = I_l  U  [2 lo(D_{l+1}) - 1, 2 hi(D_{l+1}) + 1]           downsampled[l]   (clamped to [0, W/8] for l = 3)
// as float4 {r a, g a, b a, a} per pixel (one 16-byte load serves all four channels).  downsampled[0] and
// interpolated[0], [1], [2] are never stored (the last three exist tile by tile in ip_final's LDS).  Sums left to right as written, one rounding per operator (oracle/
// interpolate_oracle.c).  One thread per pixel.  Levels >= 6 (at most 25 x 41 pixels) are launch-latency bound, so ONE
// workgroup walks them all — down 6..9, then up 8..6 — with a barrier between levels (ip_tail): 12 launches instead of 18
// (0.090 instead of 0.095 ms; starting the tail at level 5 or 4 is slower, 0.103 / 0.152 ms: a level costs a lone
// workgroup about as much as a launch costs the chip).
#include "hlmi_device_math.h"
#include "hlmi_internal.h"

using namespace hlmi;

namespace {

constexpr int IL = 10;   // GeneratorParam levels (:15)

struct Lvl {
    float4 *v;           // v[(y - y0) * w + (x - x0)]
    int x0, y0, w, h;
};
struct InGeom {
    const float *in;     // element (0, 0, 0)
    long sy, sc;
    int W, H;
};

__device__ __forceinline__ float4 ds0(const InGeom &g, int X, int Y) {   // downsampled[0] (:31), input edge-clamped
    const long off = (long)dev::clampi(Y, 0, g.H - 1) * g.sy + dev::clampi(X, 0, g.W - 1);
    const float a = g.in[3 * g.sc + off];
    return make_float4(g.in[off] * a, g.in[g.sc + off] * a, g.in[2 * g.sc + off] * a, a);
}
__device__ __forceinline__ float4 at(const Lvl &L, int X, int Y) { return L.v[(size_t)(Y - L.y0) * L.w + (X - L.x0)]; }
__device__ __forceinline__ float4 tap3(float4 a, float4 b, float4 c) {   // (a + 2 b + c) * 0.25 per channel (:51-58)
    return make_float4(((a.x + 2.0f * b.x) + c.x) * 0.25f, ((a.y + 2.0f * b.y) + c.y) * 0.25f, ((a.z + 2.0f * b.z) + c.z) * 0.25f,
                       ((a.w + 2.0f * b.w) + c.w) * 0.25f);
}

// downsampled[l](x, y) from downsampled[l-1] (src; FROM_INPUT: level 0 computed from the input);
// clamp4: the coordinate clamp in front of level 4
template<bool FROM_INPUT>
__device__ __forceinline__ float4 down_value(const InGeom &g, const Lvl &src, int x, int y, bool clamp4, int cw, int ch) {
    float4 dx[3];
#pragma unroll
    for (int j = 0; j < 3; j++) {
        float4 p[3];
#pragma unroll
        for (int i = 0; i < 3; i++) {
            int X = 2 * x - 1 + i, Y = 2 * y - 1 + j;
            if (clamp4) X = dev::clampi(X, 0, cw), Y = dev::clampi(Y, 0, ch);
            p[i] = FROM_INPUT ? ds0(g, X, Y) : at(src, X, Y);
        }
        dx[j] = tap3(p[0], p[1], p[2]);
    }
    return tap3(dx[0], dx[1], dx[2]);
}
template<bool FROM_INPUT, bool CLAMP4>
__global__ __launch_bounds__(256) void ip_down(InGeom g, Lvl src, Lvl dst, int cw, int ch) {
    const int xs = blockIdx.x * 256 + threadIdx.x, ys = blockIdx.y;
    if (xs >= dst.w) return;
    dst.v[(size_t)ys * dst.w + xs] = down_value<FROM_INPUT>(g, src, dst.x0 + xs, dst.y0 + ys, CLAMP4, cw, ch);
}

// downsampled[1] from the input, tiled: a workgroup makes 32 x 16 cells; their 65 x 33 window of downsampled[0] (the input,
// premultiplied by alpha, edge-clamped) goes through LDS once — every input pixel is a tap of up to four cells and a planar
// load of four channels — then the (1 2 1) pass in x and in y, as down_value does them.
constexpr int I0W = 64, I0H = 8;
__global__ __launch_bounds__(256) void ip_down0_tile(InGeom g, Lvl dst) {
    __shared__ float4 s_in[2 * I0H + 1][2 * I0W + 1];
    __shared__ float4 s_dx[2 * I0H + 1][I0W];
    const int tid = threadIdx.x;
    const int tx0 = dst.x0 + blockIdx.x * I0W, ty0 = dst.y0 + blockIdx.y * I0H;
    const int ix0 = 2 * tx0 - 1, iy0 = 2 * ty0 - 1;
    {
        // the thread's nine window pixels are all requested before the first is used (as a loop, every iteration waited for
        // its own four channel loads: nine memory round trips in a row at the head of every workgroup)
        constexpr int NW = (2 * I0H + 1) * (2 * I0W + 1), N1 = (NW + 255) / 256;
        float4 v[N1];
#pragma unroll
        for (int k = 0; k < N1; k++) {
            const int i = min(tid + 256 * k, NW - 1), r = i / (2 * I0W + 1), c = i - r * (2 * I0W + 1);
            v[k] = ds0(g, ix0 + c, iy0 + r);
        }
#pragma unroll
        for (int k = 0; k < N1; k++) {
            const int i = tid + 256 * k;
            if (i < NW) (&s_in[0][0])[i] = v[k];
        }
    }
    __syncthreads();
    for (int i = tid; i < (2 * I0H + 1) * I0W; i += 256) {
        const int r = i / I0W, xo = i - r * I0W;
        s_dx[r][xo] = tap3(s_in[r][2 * xo], s_in[r][2 * xo + 1], s_in[r][2 * xo + 2]);
    }
    __syncthreads();
    for (int i = tid; i < I0H * I0W; i += 256) {
        const int yo = i / I0W, xo = i - yo * I0W;
        const int xs = blockIdx.x * I0W + xo, ys = blockIdx.y * I0H + yo;
        if (xs < dst.w && ys < dst.h) dst.v[(size_t)ys * dst.w + xs] = tap3(s_dx[2 * yo][xo], s_dx[2 * yo + 1][xo], s_dx[2 * yo + 2][xo]);
    }
}

__device__ __forceinline__ float4 interp_value(const float4 d, const Lvl &up, int x, int y) {   // (:61-72)
    const int xa = dev::fdiv2(x), xb = dev::fdiv2(x + 1), ya = dev::fdiv2(y), yb = dev::fdiv2(y + 1);
    const float4 aa = at(up, xa, ya), ba = at(up, xb, ya), ab = at(up, xa, yb), bb = at(up, xb, yb);
    const float alpha = 1.0f - d.w;
    auto one = [&](float d_c, float paa, float pba, float pab, float pbb) {
        const float ua = (paa + pba) * 0.5f, ub = (pab + pbb) * 0.5f;
        return dev::mad(alpha, (ua + ub) * 0.5f, d_c);
    };
    return make_float4(one(d.x, aa.x, ba.x, ab.x, bb.x), one(d.y, aa.y, ba.y, ab.y, bb.y), one(d.z, aa.z, ba.z, ab.z, bb.z),
                       one(d.w, aa.w, ba.w, ab.w, bb.w));
}

// interpolated[l] on dst's box from downsampled[l] (ds) and interpolated[l+1] (up)
__global__ __launch_bounds__(256) void ip_up(Lvl ds, Lvl up, Lvl dst) {
    const int xs = blockIdx.x * 256 + threadIdx.x, ys = blockIdx.y;
    if (xs >= dst.w) return;
    const int x = dst.x0 + xs, y = dst.y0 + ys;
    dst.v[(size_t)ys * dst.w + xs] = interp_value(at(ds, x, y), up, x, y);
}


// ---- ip_down_multi / ip_up_multi: the middle of the pyramid (levels 3, 4, 5) in ONE launch per direction.  These levels are
// 192 x 320 pixels and below: each took its own launch of 3 - 5 us, three quarters of it the dependent-launch gap and the
// ramp of a nearly empty chip.  As in local_laplacian's ll_down_multi / ll_up_multi, a workgroup owns a tile of the LAST level
// of the chain and recomputes, in LDS, whatever it needs of the levels in between; values in a tile's halo are recomputed by
// its neighbours with identical operations (the Funcs are pure functions of their coordinates), every stored pixel is
// written by exactly one workgroup.
struct IBox {
    int x0, x1, y0, y1;   // inclusive
};
__device__ __forceinline__ IBox ibox_of(const Lvl &L) { return IBox{L.x0, L.x0 + L.w - 1, L.y0, L.y0 + L.h - 1}; }
__device__ __forceinline__ int iclamp(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
// the pixels of level box `L` (one axis: [llo, lhi]) whose `sh`-fold parent, clamped into the tile level's box [tlo, thi],
// lies in the tile [a, b] — the tile's share of that level; tiles partition the level.  Empty when lo > hi.
__device__ __forceinline__ void owned_range(int llo, int lhi, int sh, int tlo, int thi, int a, int b, int &lo, int &hi) {
    lo = a <= tlo ? llo : max(llo, a * (1 << sh));                 // the first tile also takes everything left of the tile level's box
    hi = b >= thi ? lhi : min(lhi, (b + 1) * (1 << sh) - 1);       // the last tile everything right of it
}

constexpr int DM5 = 8;                    // tile of downsampled[5]
constexpr int DM4W = 2 * DM5 + 1 + 4;     // window of downsampled[4] a tile needs / owns (+ slack at the level's edge)
constexpr int DM3W = 2 * DM4W + 1 + 4;    // ... of downsampled[3]
__global__ __launch_bounds__(256) void ip_down_multi(Lvl src, Lvl d3, Lvl d4, Lvl d5, int cw, int ch, int ntx) {
    __shared__ float4 s3[DM3W * DM3W], s4[DM4W * DM4W];
    const int tid = threadIdx.x;
    const int bx = blockIdx.x % ntx, by = blockIdx.x / ntx;
    const IBox B3 = ibox_of(d3), B4 = ibox_of(d4), B5 = ibox_of(d5);
    // the tile of level 5
    const IBox T5 = {B5.x0 + bx * DM5, min(B5.x0 + bx * DM5 + DM5 - 1, B5.x1), B5.y0 + by * DM5, min(B5.y0 + by * DM5 + DM5 - 1, B5.y1)};
    // level 4: what the tile reads (2x - 1 .. 2x + 1) and what it owns
    IBox O4, O3;
    owned_range(B4.x0, B4.x1, 1, B5.x0, B5.x1, T5.x0, T5.x1, O4.x0, O4.x1);
    owned_range(B4.y0, B4.y1, 1, B5.y0, B5.y1, T5.y0, T5.y1, O4.y0, O4.y1);
    const IBox W4 = {min(2 * T5.x0 - 1, O4.x0), max(2 * T5.x1 + 1, O4.x1), min(2 * T5.y0 - 1, O4.y0), max(2 * T5.y1 + 1, O4.y1)};
    // level 3: read through the generator's coordinate clamp (:40-49), plus what the tile owns
    owned_range(B3.x0, B3.x1, 2, B5.x0, B5.x1, T5.x0, T5.x1, O3.x0, O3.x1);
    owned_range(B3.y0, B3.y1, 2, B5.y0, B5.y1, T5.y0, T5.y1, O3.y0, O3.y1);
    const IBox W3 = {min(iclamp(2 * W4.x0 - 1, 0, cw), O3.x0), max(iclamp(2 * W4.x1 + 1, 0, cw), O3.x1),
                     min(iclamp(2 * W4.y0 - 1, 0, ch), O3.y0), max(iclamp(2 * W4.y1 + 1, 0, ch), O3.y1)};
    const int w3 = W3.x1 - W3.x0 + 1, h3 = W3.y1 - W3.y0 + 1, w4 = W4.x1 - W4.x0 + 1, h4 = W4.y1 - W4.y0 + 1;
    const InGeom none{};
    for (int i = tid; i < w3 * h3; i += 256) {
        const int yy = i / w3, xx = i - yy * w3, x = W3.x0 + xx, y = W3.y0 + yy;
        const float4 v = down_value<false>(none, src, x, y, false, cw, ch);
        s3[yy * DM3W + xx] = v;
        if (x >= O3.x0 && x <= O3.x1 && y >= O3.y0 && y <= O3.y1) d3.v[(size_t)(y - d3.y0) * d3.w + (x - d3.x0)] = v;
    }
    __syncthreads();
    for (int i = tid; i < w4 * h4; i += 256) {
        const int yy = i / w4, xx = i - yy * w4, x = W4.x0 + xx, y = W4.y0 + yy;
        float4 dx[3];
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const int Y = iclamp(2 * y - 1 + j, 0, ch) - W3.y0;
            const float4 a = s3[Y * DM3W + iclamp(2 * x - 1, 0, cw) - W3.x0], b = s3[Y * DM3W + iclamp(2 * x, 0, cw) - W3.x0],
                         c = s3[Y * DM3W + iclamp(2 * x + 1, 0, cw) - W3.x0];
            dx[j] = tap3(a, b, c);
        }
        const float4 v = tap3(dx[0], dx[1], dx[2]);
        s4[yy * DM4W + xx] = v;
        if (x >= O4.x0 && x <= O4.x1 && y >= O4.y0 && y <= O4.y1) d4.v[(size_t)(y - d4.y0) * d4.w + (x - d4.x0)] = v;
    }
    __syncthreads();
    const int w5 = T5.x1 - T5.x0 + 1, h5 = T5.y1 - T5.y0 + 1;
    for (int i = tid; i < w5 * h5; i += 256) {
        const int yy = i / w5, xx = i - yy * w5, x = T5.x0 + xx, y = T5.y0 + yy;
        float4 dx[3];
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const float4 *row = s4 + (2 * y - 1 + j - W4.y0) * DM4W + (2 * x - 1 - W4.x0);
            dx[j] = tap3(row[0], row[1], row[2]);
        }
        d5.v[(size_t)(y - d5.y0) * d5.w + (x - d5.x0)] = tap3(dx[0], dx[1], dx[2]);
    }
}

// interpolated[3] on its box from interpolated[6] (the tail's result), downsampled[5], [4], [3]; interpolated[5] and [4]
// exist only in LDS (their only reader is the level below)
constexpr int UM3 = 32;                   // tile of interpolated[3]
constexpr int UM4W = UM3 / 2 + 2, UM5W = UM4W / 2 + 2;
__device__ __forceinline__ float4 interp_from(const float4 d, const float4 aa, const float4 ba, const float4 ab, const float4 bb) {   // (:61-72)
    const float alpha = 1.0f - d.w;
    auto one = [&](float d_c, float paa, float pba, float pab, float pbb) {
        const float ua = (paa + pba) * 0.5f, ub = (pab + pbb) * 0.5f;
        return dev::mad(alpha, (ua + ub) * 0.5f, d_c);
    };
    return make_float4(one(d.x, aa.x, ba.x, ab.x, bb.x), one(d.y, aa.y, ba.y, ab.y, bb.y), one(d.z, aa.z, ba.z, ab.z, bb.z),
                       one(d.w, aa.w, ba.w, ab.w, bb.w));
}
__global__ __launch_bounds__(256) void ip_up_multi(Lvl d3, Lvl d4, Lvl d5, Lvl up6, Lvl dst3, int ntx) {
    __shared__ float4 s5[UM5W * UM5W], s4[UM4W * UM4W];
    const int tid = threadIdx.x;
    const int bx = blockIdx.x % ntx, by = blockIdx.x / ntx;
    const IBox T3 = {dst3.x0 + bx * UM3, min(dst3.x0 + bx * UM3 + UM3 - 1, dst3.x0 + dst3.w - 1), dst3.y0 + by * UM3,
                     min(dst3.y0 + by * UM3 + UM3 - 1, dst3.y0 + dst3.h - 1)};
    const IBox N4 = {dev::fdiv2(T3.x0), dev::fdiv2(T3.x1 + 1), dev::fdiv2(T3.y0), dev::fdiv2(T3.y1 + 1)};
    const IBox N5 = {dev::fdiv2(N4.x0), dev::fdiv2(N4.x1 + 1), dev::fdiv2(N4.y0), dev::fdiv2(N4.y1 + 1)};
    const int w5 = N5.x1 - N5.x0 + 1, h5 = N5.y1 - N5.y0 + 1, w4 = N4.x1 - N4.x0 + 1, h4 = N4.y1 - N4.y0 + 1;
    for (int i = tid; i < w5 * h5; i += 256) {
        const int yy = i / w5, xx = i - yy * w5, x = N5.x0 + xx, y = N5.y0 + yy;
        s5[yy * UM5W + xx] = interp_value(at(d5, x, y), up6, x, y);
    }
    __syncthreads();
    for (int i = tid; i < w4 * h4; i += 256) {
        const int yy = i / w4, xx = i - yy * w4, x = N4.x0 + xx, y = N4.y0 + yy;
        const int xa = dev::fdiv2(x) - N5.x0, xb = dev::fdiv2(x + 1) - N5.x0, ya = dev::fdiv2(y) - N5.y0, yb = dev::fdiv2(y + 1) - N5.y0;
        s4[yy * UM4W + xx] = interp_from(at(d4, x, y), s5[ya * UM5W + xa], s5[ya * UM5W + xb], s5[yb * UM5W + xa], s5[yb * UM5W + xb]);
    }
    __syncthreads();
    const int w3 = T3.x1 - T3.x0 + 1, h3 = T3.y1 - T3.y0 + 1;
    for (int i = tid; i < w3 * h3; i += 256) {
        const int yy = i / w3, xx = i - yy * w3, x = T3.x0 + xx, y = T3.y0 + yy;
        const int xa = dev::fdiv2(x) - N4.x0, xb = dev::fdiv2(x + 1) - N4.x0, ya = dev::fdiv2(y) - N4.y0, yb = dev::fdiv2(y + 1) - N4.y0;
        dst3.v[(size_t)(y - dst3.y0) * dst3.w + (x - dst3.x0)] =
            interp_from(at(d3, x, y), s4[ya * UM4W + xa], s4[ya * UM4W + xb], s4[yb * UM4W + xa], s4[yb * UM4W + xb]);
    }
}

// Levels from..IL-1 in one launch of ONE workgroup: downsampled[from..9], then interpolated[8..from]; a level is
// complete (and visible to the workgroup, which shares its CU's L1) at the barrier that follows it.
struct TailArgs {
    Lvl ds[IL], ip[IL];
    int cw, ch, from;
};
__global__ __launch_bounds__(1024) void ip_tail(TailArgs a) {
    const InGeom none{};
    for (int l = a.from; l < IL; l++) {
        const Lvl src = a.ds[l - 1], dst = a.ds[l];
        for (int i = threadIdx.x; i < dst.w * dst.h; i += 1024) {
            const int ys = i / dst.w, xs = i - ys * dst.w;
            dst.v[i] = down_value<false>(none, src, dst.x0 + xs, dst.y0 + ys, l == 4, a.cw, a.ch);
        }
        __syncthreads();
    }
    for (int l = IL - 2; l >= a.from; l--) {
        const Lvl d = a.ds[l], up = a.ip[l + 1], dst = a.ip[l];
        for (int i = threadIdx.x; i < dst.w * dst.h; i += 1024) {
            const int ys = i / dst.w, xs = i - ys * dst.w;
            dst.v[i] = interp_value(at(d, dst.x0 + xs, dst.y0 + ys), up, dst.x0 + xs, dst.y0 + ys);
        }
        __syncthreads();
    }
}

// levels 2, 1 and 0: interpolated[2] and [1] of a tile's neighbourhood in LDS (from downsampled[2], [1] and interpolated[3]; each
// is read only by the level below, so neither is stored: 16 B per pixel of those levels less each way, and two launches less than
// ip_up:2 + ip_up:1 + a per-pixel final kernel), then interpolated[0] (never stored either) and the normalisation (:74-75),
// planar output.  A workgroup owns F0W x F0H output pixels, four rows of them per thread; cells on a tile's edge are recomputed
// by the neighbour with the same operations.
constexpr int F0W = 64, F0H = 16, F1W = F0W / 2 + 1, F1H = F0H / 2 + 1, F2W = F1W / 2 + 2, F2H = F1H / 2 + 2;
__global__ __launch_bounds__(256) void ip_final(InGeom g, Lvl d1, Lvl d2, Lvl up3, float *__restrict__ out, long out_sy, long out_sc) {
    __shared__ float4 s1[F1H * F1W], s2[F2H * F2W];
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * F0W, y0 = blockIdx.y * F0H;
    const int x1 = min(x0 + F0W, g.W) - 1, y1 = min(y0 + F0H, g.H) - 1;
    // the cells of interpolated[1] the tile reads, and those of interpolated[2] they read
    const int ax0 = x0 >> 1, ay0 = y0 >> 1, w1 = ((x1 + 1) >> 1) - ax0 + 1, h1 = ((y1 + 1) >> 1) - ay0 + 1;
    const int bx0 = ax0 >> 1, by0 = ay0 >> 1, w2 = ((ax0 + w1) >> 1) - bx0 + 1, h2 = ((ay0 + h1) >> 1) - by0 + 1;
    for (int i = tid; i < w2 * h2; i += 256) {
        const int yy = i / w2, xx = i - yy * w2, x = bx0 + xx, y = by0 + yy;
        s2[yy * F2W + xx] = interp_value(at(d2, x, y), up3, x, y);
    }
    __syncthreads();
    for (int i = tid; i < w1 * h1; i += 256) {
        const int yy = i / w1, xx = i - yy * w1, x = ax0 + xx, y = ay0 + yy;
        const int xa = (x >> 1) - bx0, xb = ((x + 1) >> 1) - bx0, ya = (y >> 1) - by0, yb = ((y + 1) >> 1) - by0;
        s1[yy * F1W + xx] = interp_from(at(d1, x, y), s2[ya * F2W + xa], s2[ya * F2W + xb], s2[yb * F2W + xa], s2[yb * F2W + xb]);
    }
    __syncthreads();
    const int x = x0 + (tid & 63);
    if (x > x1) return;
    const int xa = (x >> 1) - ax0, xb = ((x + 1) >> 1) - ax0;
#pragma unroll
    for (int k = 0; k < F0H / 4; k++) {
        const int y = y0 + (tid >> 6) + 4 * k;
        if (y > y1) break;
        const int ya = (y >> 1) - ay0, yb = ((y + 1) >> 1) - ay0;
        // interpolated[0] with downsampled[0] INLINE, as the reference's CPU schedule has it inside `normalize` (:178-188): for the colour
        // channels the sum is product + product — under the fma canon the first one, input * alpha-channel, is the one fused (dev::mad2)
        float4 v;
        {
            const long off = (long)y * g.sy + x;   // inside the image
            const float a = g.in[3 * g.sc + off], alpha = 1.0f - a;
            const float4 aa = s1[ya * F1W + xa], ba = s1[ya * F1W + xb], ab = s1[yb * F1W + xa], bb = s1[yb * F1W + xb];
            auto up = [](float paa, float pba, float pab, float pbb) {
                const float ua = (paa + pba) * 0.5f, ub = (pab + pbb) * 0.5f;
                return (ua + ub) * 0.5f;
            };
            v.x = dev::mad2(g.in[off], a, alpha, up(aa.x, ba.x, ab.x, bb.x));
            v.y = dev::mad2(g.in[g.sc + off], a, alpha, up(aa.y, ba.y, ab.y, bb.y));
            v.z = dev::mad2(g.in[2 * g.sc + off], a, alpha, up(aa.z, ba.z, ab.z, bb.z));
            v.w = dev::mad(alpha, up(aa.w, ba.w, ab.w, bb.w), a);
        }
        float *o = out + (long)y * out_sy + x;
        o[0] = v.x / v.w, o[out_sc] = v.y / v.w, o[2 * out_sc] = v.z / v.w;
    }
}

const int64_t e0 = 0, ew = 1536, eh = 2560, e4 = 4, e3 = 3;
const int64_t *const est_in[6] = {&e0, &ew, &e0, &eh, &e0, &e4};
const int64_t *const est_out[6] = {&e0, &ew, &e0, &eh, &e0, &e3};
const halide_type_t ty_f32 = {(decltype(halide_type_t::code))2, 32, 0};
// estimates: generator :204-211
const halide_filter_argument_t ip_args[2] = {
    {"input", halide_argument_kind_input_buffer, 3, ty_f32, nullptr, nullptr, nullptr, nullptr, est_in},
    {"output", halide_argument_kind_output_buffer, 3, ty_f32, nullptr, nullptr, nullptr, nullptr, est_out},
};
const halide_filter_metadata_t ip_md = {1, 2, ip_args, kTargetString, "interpolate"};

struct Box {
    int x0, x1, y0, y1;
};

}  // namespace

extern "C" int interpolate(halide_buffer_t *input, halide_buffer_t *output) {
    void *uc = nullptr;
    BufArg args[2] = {{"input", input, T_F32, 3, false}, {"output", output, T_F32, 3, true}};
    int r = check_not_null(uc, args, 2);
    if (r) return r;
    if ((r = check_type_and_dims(uc, args, 2))) return r;
    auto real = [](halide_buffer_t *b) { return !(b->host == nullptr && b->device == 0); };
    if (any_bounds_query(args, 2)) {
        // output = input's [0,W) x [0,H) with 3 channels (:83-87); input has channels [0,4) (:23)
        // (a real input pins the box; else the output's shape — real, or given by the caller of an all-null query — else a shaped
        // input, else the estimates)
        halide_buffer_t *k = real(input) ? input : buffer_known(output) ? output : input;
        const bool any = real(input) || buffer_known(output) || buffer_has_shape(input);
        const int w = any ? k->dim[0].extent : 1536, h = any ? k->dim[1].extent : 2560;
        int z[3] = {0, 0, 0}, ei[3] = {w, h, 4}, eo[3] = {w, h, 3};
        answer_query(input, z, ei);
        answer_query(output, z, eo);
        return 0;
    }
    if ((r = check_shape(uc, args[0])) || (r = check_shape(uc, args[1]))) return r;
    const int W = input->dim[0].extent, H = input->dim[1].extent;
    // input.dim(2).set_bounds(0, 4) (:23); normalize.bound(x, 0, input.width()) etc. (:83-87)
    if ((r = check_equal(uc, "input.min.2", input->dim[2].min, "0", 0)) || (r = check_equal(uc, "input.extent.2", input->dim[2].extent, "4", 4)) ||
        (r = check_equal(uc, "input.min.0", input->dim[0].min, "0", 0)) || (r = check_equal(uc, "input.min.1", input->dim[1].min, "0", 0)) ||
        (r = check_equal(uc, "output.min.0", output->dim[0].min, "0", 0)) || (r = check_equal(uc, "output.extent.0", output->dim[0].extent, "input.extent.0", W)) ||
        (r = check_equal(uc, "output.min.1", output->dim[1].min, "0", 0)) || (r = check_equal(uc, "output.extent.1", output->dim[1].extent, "input.extent.1", H)) ||
        (r = check_equal(uc, "output.min.2", output->dim[2].min, "0", 0)) || (r = check_equal(uc, "output.extent.2", output->dim[2].extent, "3", 3))) {
        return r;
    }
    DeviceCtx ctx;
    if ((r = acquire_device(uc, &ctx))) return r;
    if ((r = input_to_device(uc, ctx, args[0]))) return r;
    if ((r = output_on_device(uc, ctx, args[1]))) return r;
    if (W > 0 && H > 0) {
        Box I[IL], D[IL];
        I[0] = {0, W - 1, 0, H - 1};
        for (int l = 1; l < IL; l++) I[l] = {0, floor_div(I[l - 1].x1 + 1, 2), 0, floor_div(I[l - 1].y1 + 1, 2)};
        D[IL - 1] = I[IL - 1];
        const int cw = W / 8, ch = H / 8;
        auto clampi = [](int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); };
        for (int l = IL - 2; l >= 0; l--) {
            Box n = {2 * D[l + 1].x0 - 1, 2 * D[l + 1].x1 + 1, 2 * D[l + 1].y0 - 1, 2 * D[l + 1].y1 + 1};
            if (l + 1 == 4) n = {clampi(n.x0, 0, cw), clampi(n.x1, 0, cw), clampi(n.y0, 0, ch), clampi(n.y1, 0, ch)};
            D[l] = {n.x0 < I[l].x0 ? n.x0 : I[l].x0, n.x1 > I[l].x1 ? n.x1 : I[l].x1, n.y0 < I[l].y0 ? n.y0 : I[l].y0,
                    n.y1 > I[l].y1 ? n.y1 : I[l].y1};
        }
        size_t off_d[IL], off_i[IL], total = 0;
        auto area = [](const Box &b) { return ((size_t)(b.x1 - b.x0 + 1) * (b.y1 - b.y0 + 1) + 15) & ~(size_t)15; };
        for (int l = 1; l < IL; l++) off_d[l] = total, total += area(D[l]);
        for (int l = 3; l < IL - 1; l++) off_i[l] = total, total += area(I[l]);   // interpolated[2] and [1] live in ip_final's LDS
        void *ws = nullptr;
        if ((r = get_workspace(uc, ctx, total * sizeof(float4), &ws))) return r;
        float4 *base = (float4 *)ws;
        auto lvl = [&](float4 *p, const Box &b) { return Lvl{p, b.x0, b.y0, b.x1 - b.x0 + 1, b.y1 - b.y0 + 1}; };
        Lvl ds[IL] = {}, ip[IL] = {};
        for (int l = 1; l < IL; l++) ds[l] = lvl(base + off_d[l], D[l]);
        for (int l = 3; l < IL - 1; l++) ip[l] = lvl(base + off_i[l], I[l]);
        ip[IL - 1] = ds[IL - 1];   // interpolated[9] = downsampled[9], and D_9 == I_9
        InGeom g{dev_ptr<float>(input), input->dim[1].stride, input->dim[2].stride, W, H};
        hipStream_t st = ctx.stream;
        char nm[32];
        // levels >= T go through ip_tail (default 6: 25 x 41 pixels and below)
        const char *te = getenv("HLMI_IP_TAIL_FROM");
        int T = te ? atoi(te) : 6;
        if (T < 3 || T > IL - 2) T = IL;   // (interpolated[2] and [1] are not stored: the tail cannot start below level 3)
        // levels 3, 4, 5 in one launch per direction (ip_down_multi / ip_up_multi) when the tail starts at 6 and every tile's
        // windows fit the kernels' LDS arrays (they do for any image: the check guards the constants, not the input)
        bool fused = T == 6 && !env_flag("HLMI_IP_UNFUSED");
        const int ntx5 = (ds[5].w + DM5 - 1) / DM5, nty5 = (ds[5].h + DM5 - 1) / DM5;
        if (fused) {
            auto owned = [](int llo, int lhi, int sh, int tlo, int thi, int a, int b, int &lo, int &hi) {
                lo = a <= tlo ? llo : (llo > a * (1 << sh) ? llo : a * (1 << sh));
                hi = b >= thi ? lhi : (lhi < (b + 1) * (1 << sh) - 1 ? lhi : (b + 1) * (1 << sh) - 1);
            };
            auto mn = [](int a, int b) { return a < b ? a : b; };
            auto mx = [](int a, int b) { return a > b ? a : b; };
            for (int axis = 0; axis < 2 && fused; axis++) {
                const int b5lo = axis ? D[5].y0 : D[5].x0, b5hi = axis ? D[5].y1 : D[5].x1, b4lo = axis ? D[4].y0 : D[4].x0,
                          b4hi = axis ? D[4].y1 : D[4].x1, b3lo = axis ? D[3].y0 : D[3].x0, b3hi = axis ? D[3].y1 : D[3].x1;
                const int c = axis ? ch : cw;
                for (int a = b5lo; a <= b5hi; a += DM5) {
                    const int b = mn(a + DM5 - 1, b5hi);
                    int o4l, o4h, o3l, o3h;
                    owned(b4lo, b4hi, 1, b5lo, b5hi, a, b, o4l, o4h);
                    owned(b3lo, b3hi, 2, b5lo, b5hi, a, b, o3l, o3h);
                    const int w4l = mn(2 * a - 1, o4l), w4h = mx(2 * b + 1, o4h);
                    const int w3l = mn(clampi(2 * w4l - 1, 0, c), o3l), w3h = mx(clampi(2 * w4h + 1, 0, c), o3h);
                    if (w4h - w4l + 1 > DM4W || w3h - w3l + 1 > DM3W || w4l < b4lo || w4h > b4hi || w3l < b3lo || w3h > b3hi) fused = false;
                }
            }
        }
        for (int l = 1; l < IL && l < T; l++) {
            if (fused && l >= 3) {
                if (l == 3) {
                    HLMI_LAUNCH(uc, "ip_down_multi:3", st, ip_down_multi, dim3(ntx5 * nty5), dim3(256), 0, ds[2], ds[3], ds[4], ds[5], cw, ch, ntx5);
                }
                continue;
            }
            dim3 grid((ds[l].w + 255) / 256, ds[l].h), block(256);
            snprintf(nm, sizeof nm, "ip_down:%d", l);
            if (l == 1) {
                HLMI_LAUNCH(uc, nm, st, ip_down0_tile, dim3((ds[1].w + I0W - 1) / I0W, (ds[1].h + I0H - 1) / I0H), dim3(256), 0, g, ds[1]);
            }
            else if (l == 4) HLMI_LAUNCH(uc, nm, st, (ip_down<false, true>), grid, block, 0, g, ds[l - 1], ds[l], cw, ch);
            else HLMI_LAUNCH(uc, nm, st, (ip_down<false, false>), grid, block, 0, g, ds[l - 1], ds[l], cw, ch);
        }
        if (T < IL) {
            TailArgs ta;
            for (int l = 0; l < IL; l++) ta.ds[l] = ds[l], ta.ip[l] = ip[l];
            ta.cw = cw, ta.ch = ch, ta.from = T;
            snprintf(nm, sizeof nm, "ip_tail:%d", T);
            HLMI_LAUNCH(uc, nm, st, ip_tail, dim3(1), dim3(1024), 0, ta);
        }
        for (int l = (T < IL ? T - 1 : IL - 2); l >= 3; l--) {   // (levels 2 and 1 are made inside ip_final)
            if (fused && l >= 3) {
                if (l == 3) {
                    const int ntx3 = (ip[3].w + UM3 - 1) / UM3, nty3 = (ip[3].h + UM3 - 1) / UM3;
                    HLMI_LAUNCH(uc, "ip_up_multi:3", st, ip_up_multi, dim3(ntx3 * nty3), dim3(256), 0, ds[3], ds[4], ds[5], ip[6], ip[3], ntx3);
                }
                continue;
            }
            dim3 grid((ip[l].w + 255) / 256, ip[l].h), block(256);
            snprintf(nm, sizeof nm, "ip_up:%d", l);
            HLMI_LAUNCH(uc, nm, st, ip_up, grid, block, 0, ds[l], ip[l + 1], ip[l]);
        }
        timing_note_bytes(28.0 * W * H);
        HLMI_LAUNCH(uc, "ip_final", st, ip_final, dim3((W + F0W - 1) / F0W, (H + F0H - 1) / F0H), dim3(256), 0, g, ds[1], ds[2], ip[3], dev_ptr<float>(output),
                    (long)output->dim[1].stride, (long)output->dim[2].stride);
    }
    mark_output_written(output);
    return 0;
}

extern "C" int interpolate_argv(void **a) { return interpolate((halide_buffer_t *)a[0], (halide_buffer_t *)a[1]); }
extern "C" const halide_filter_metadata_t *interpolate_metadata(void) { return &ip_md; }
extern "C" int interpolate_auto_schedule(halide_buffer_t *input, halide_buffer_t *output) { return interpolate(input, output); }
