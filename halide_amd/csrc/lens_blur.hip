// lens_blur.hip — gfx950 implementation of the reference's lens_blur AOT pipeline (adjacent app, same
// boundary).  Algorithm: /root/reference/apps/lens_blur/lens_blur_generator.cpp:24-152 (downsample :279-285, upsample
// :288-294); boundary: `int lens_blur(halide_buffer_t *left_im, halide_buffer_t *right_im, int32_t slices, int32_t
// focus_depth, float blur_radius_scale, int32_t aperture_samples, halide_buffer_t *final)` (:12-21, :301).
//
// Stages (one launch each unless noted; every Func is evaluated on the box its consumers read, as the oracle does —
// oracle/lens_blur_oracle.c has the derivation of the boxes D, P_i, E):
//   lb_cost     per pixel of E: the `slices` stereo costs, their confidence (variance across the stack) and
//               cost_pyramid_push[0] = {cost * confidence, confidence}                                  (:30-56)
//   lb_down     cost_pyramid_push[i], i = 1..7: 1-3-3-1 in x then in y of the level above, clamped box (:57-63)
//   lb_pull     cost_pyramid_pull[i], i = 7..1: lerp(upsample(pull[i+1]), push[i], 1/2) on P_i         (:65-71)
//   lb_depth    pull[0] on the fly, filtered_cost, argmin over the slices -> depth, bokeh radius on D   (:73-89)
//   lb_wcy      vertical running maximum of the bokeh radius                                           (:94-99)
//   lb_final    horizontal maximum, the aperture samples, weights, accumulation, normalisation         (:99-152)
// Float operations are in the generator's order with one rounding each (-ffp-contract=off).  The sample positions come
// from Halide's random_float(): a fixed hash (src/Random.cpp:20-104) of (call id, definition tag, s, y, x); the call ids
// are 0 and 1, the definition tag is a compile-time counter of the reference's compiler that cannot be observed here —
// it defaults to the count derived in the oracle's header (71) and is settable: hlmi_lens_blur_set_random_tag().
//
//   lb_tail     the levels with at most 128 x 128 elements per plane (4..7 at 768 x 1280) of BOTH pyramids in one launch:
//               planes never mix, so a workgroup takes its plane down and back up with a barrier between levels
// Layout: push[i] / pull[i] are float [slices + 1][box_h][box_w] (x fastest; plane `slices` = the confidence, see at()).
// What is shared goes through LDS (the right-image row segment of lb_cost, the source window of lb_down), what a stage needs
// per sample is packed where it is produced (lb_depth -> lb_final); the rest is one thread per element.
#include "hlmi_device_math.h"
#include "hlmi_internal.h"

#include <atomic>
#include <stdlib.h>

using namespace hlmi;

namespace {

constexpr int LV = 8;
std::atomic<int> g_rand_tag{71};

struct Box {
    int x0, y0, w, h;   // origin (absolute), extents
};
struct LBGeom {
    int lx0, lx1, ly0, ly1;      // clamp box of left_im (absolute)
    int rx0, rx1, ry0, ry1;      // clamp box of right_im
    long l_sy, l_sc, r_sy, r_sc; // strides (elements); channel index clamped by the caller: offsets below
    long l_c[3], r_c[3];         // element offsets of the (clamped) channels 0..2
    int slices, focus, samples, R, tag;
    float scale, fslices;
};

// plane of (z, c): c = 0 (cost x confidence) has one plane per slice, c = 1 (the confidence itself) does not depend on z at
// any level — the generator carries `slices` identical copies of it through both pyramids (:56-71) — and is kept ONCE, as
// plane `slices`: half the planes, half the traffic of every stage up to the depth map.
__device__ __forceinline__ size_t at(const Box &b, int slices, int z, int c, int x, int y) {
    return ((size_t)(c ? slices : z) * b.h + (y - b.y0)) * b.w + (x - b.x0);
}

constexpr int LB_MAXS = 64;   // generator :14: slices <= 64

// The cost stack of one pixel (:30-55).  cost(x, y, z) = sum over the channels of min(|l - r(x + 2z)|, |l - r(x + 2z + 1)|)^2
// in float — every intermediate is an integer below 2^24, so the float operations of the generator are exact and the same
// value comes out of integer arithmetic: packed 16-bit differences for channels 0 and 1, a dot product for the squares.  The
// right-image pixels of a row segment sit in LDS as one 8-byte record {c0 | c1 << 16, c2} each.  cost / slices is a
// correctly rounded division in the generator: for an integer numerator below 2^18 and a divisor 1..64 the two-FMA
// correction of q0 = a * RN(1 / n) IS the correctly rounded quotient (checked exhaustively: tests/test_lens_blur.py,
// tests/cpp/lb_div_check.c), which takes 3 instructions instead of the 11 of a general division.
typedef short lb_s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short lb_u16x2 __attribute__((ext_vector_type(2)));
struct CostRow {
    uint32_t l01;   // c0 | c1 << 16 of the left pixel
    int l2;
};
// (a + 3 (b + c) + d) / 8 (:282-283); the product has one use: fused with the first add under the fma canon
__device__ __forceinline__ float lb_down4(float a, float b, float c, float d) { return (dev::mad(3.0f, b + c, a) + d) * 0.125f; }
__device__ __forceinline__ uint2 lb_pack(int c0, int c1, int c2) { return make_uint2((uint32_t)c0 | (uint32_t)c1 << 16, (uint32_t)c2); }
__device__ __forceinline__ float lb_cost1(const uint2 A, const uint2 B, const CostRow &c) {
    const lb_s16x2 l = __builtin_bit_cast(lb_s16x2, c.l01), a = __builtin_bit_cast(lb_s16x2, A.x), b = __builtin_bit_cast(lb_s16x2, B.x);
    const lb_s16x2 da = l - a, db = l - b, z2 = {0, 0};
    const lb_u16x2 ma = __builtin_bit_cast(lb_u16x2, __builtin_elementwise_max(da, z2 - da)),
                   mb = __builtin_bit_cast(lb_u16x2, __builtin_elementwise_max(db, z2 - db));
    const lb_u16x2 m01 = __builtin_elementwise_min(ma, mb);
    const int ea = c.l2 - (int)A.y, eb = c.l2 - (int)B.y;
    const int m2 = min(max(ea, -ea), max(eb, -eb));
    const uint32_t s = __builtin_amdgcn_udot2(m01, m01, (uint32_t)__mul24(m2, m2), false);
    return (float)s;
}
// FULL: slices == SB (the default 32, and 64): no guards at all.  Otherwise the whole stack of SB costs is computed without
// branches (the LDS reads stay inside the staged array; what lies beyond the row segment is never used) and the slices that
// do not exist add 0.0f to the two sums, which leaves a non-negative sum as it is.
template<int SB, bool FULL>
__device__ __forceinline__ float cost_stack(const uint2 *__restrict__ sr, int tid, const CostRow &c, const LBGeom &g, float (&cz)[SB]) {
    float sa = 0.0f, sb = 0.0f;
    const float rn = 1.0f / g.fslices;
#pragma unroll
    for (int z = 0; z < SB; z++) {
        cz[z] = lb_cost1(sr[tid + 2 * z], sr[tid + 2 * z + 1], c);
        const bool on = FULL || z < g.slices;
        sa = dev::mad(on ? cz[z] : 0.0f, cz[z], sa);   // sum(pow(cost, 2)) (:44); a slice that does not exist adds +0
        const float q0 = cz[z] * rn;                                   // cz / fslices, see above
        const float q = __builtin_fmaf(__builtin_fmaf(-g.fslices, q0, cz[z]), rn, q0);
        sb = sb + (on ? q : 0.0f);
    }
    return dev::msub(sa / g.fslices, sb, sb);   // the confidence (:44-46)
}

// One workgroup = 256 consecutive pixels of a row of E.  Every pixel compares itself with the 2 * slices right-image pixels
// to its right (:33-36), so neighbours share almost all of them: the row segment [x0, x0 + 256 + 2 slices) of the (clamped)
// right image is staged in LDS once and the left pixel sits in registers.
// SB: compile-time bound of `slices` (32 or 64): the costs of a pixel stay in registers between the pass that finds their
// variance and the pass that writes them scaled by it
template<int SB, bool FULL>
__global__ __launch_bounds__(256) void lb_cost(const uint8_t *__restrict__ L, const uint8_t *__restrict__ Rr, LBGeom g, Box E,
                                              float *__restrict__ push0) {
    __shared__ uint2 sr[256 + 2 * LB_MAXS];
    const int tid = threadIdx.x, xi = blockIdx.x * 256 + tid, yi = blockIdx.y;
    const int x = E.x0 + xi, y = E.y0 + yi;
    {
        const long ro = (long)(dev::clampi(y, g.ry0, g.ry1) - g.ry0) * g.r_sy;
        const int xb = E.x0 + blockIdx.x * 256;
        for (int i = tid; i < 256 + 2 * g.slices; i += 256) {
            const long o = ro + (dev::clampi(xb + i, g.rx0, g.rx1) - g.rx0);
            sr[i] = lb_pack(Rr[o + g.r_c[0]], Rr[o + g.r_c[1]], Rr[o + g.r_c[2]]);
        }
    }
    __syncthreads();
    if (xi >= E.w) return;
    const long lo = (long)(dev::clampi(y, g.ly0, g.ly1) - g.ly0) * g.l_sy + (dev::clampi(x, g.lx0, g.lx1) - g.lx0);
    const CostRow c = {(uint32_t)L[lo + g.l_c[0]] | (uint32_t)L[lo + g.l_c[1]] << 16, (int)L[lo + g.l_c[2]]};
    float cz[SB];
    const float conf = cost_stack<SB, FULL>(sr, tid, c, g, cz);
#pragma unroll
    for (int z = 0; z < SB; z++)
        if (FULL || z < g.slices) push0[at(E, g.slices, z, 0, x, y)] = cz[z] * conf;
    push0[at(E, g.slices, 0, 1, x, y)] = conf;
}

// src level as the total function the generator defines: clamped to its box for levels >= 1 (:62), direct for level 0
template<bool CLAMP>
__device__ __forceinline__ float src_at(const float *__restrict__ s, const Box &b, int zc, int x, int y) {
    if (CLAMP) x = dev::clampi(x, b.x0, b.x0 + b.w - 1), y = dev::clampi(y, b.y0, b.y0 + b.h - 1);
    return s[((size_t)zc * b.h + (y - b.y0)) * b.w + (x - b.x0)];
}

// ---- 32-bit index helpers of the *32 kernels (planes below 2^29 elements, boxes below 2^23 wide / high: the host checks)
template<int D, int MAXE>
__device__ __forceinline__ uint32_t lb_divc(uint32_t e) {   // e / D for 0 <= e <= MAXE
    constexpr uint32_t M = (65536u + D - 1) / D;
    static_assert((unsigned long long)(M * D - 65536u) * MAXE < 65536ull, "lb_divc: range");
    return __umul24(e, M) >> 16;
}
__device__ __forceinline__ uint32_t lb_mul24(uint32_t a, uint32_t b) {   // kept apart from a following add (else: v_mad_u64_u32, quarter rate)
    uint32_t r = __umul24(a, b);
    asm("" : "+v"(r));
    return r;
}
__device__ __forceinline__ uint32_t lb_clamped_off(const Box &b, int x, int y) {   // element offset of (x, y) clamped to the box, in its plane
    const int cx = dev::clampi(x, b.x0, b.x0 + b.w - 1) - b.x0, cy = dev::clampi(y, b.y0, b.y0 + b.h - 1) - b.y0;
    return lb_mul24((uint32_t)cy, (uint32_t)b.w) + (uint32_t)cx;
}

// A workgroup makes a 64 x 16 tile of one plane of the next level: its 130 x 34 source window goes through LDS once (every
// source value is a tap of four outputs), the 1-3-3-1 pass in x over the 34 rows, then in y — the generator's expressions
// (:279-285), operand for operand.  Source coordinates are clamped to the source box when staging: that IS the function for
// levels >= 1 (:62) and only feeds outputs outside the destination box for level 0.
constexpr int DTW = 64, DTH = 16;
template<bool SRC_CLAMP>
__global__ __launch_bounds__(256) void lb_down(const float *__restrict__ src, Box sb, float *__restrict__ dst, Box db) {
    __shared__ float s_in[2 * DTH + 2][2 * DTW + 3];
    __shared__ float s_dx[2 * DTH + 2][DTW + 1];
    const int tid = threadIdx.x, zc = blockIdx.z;
    const int tx0 = db.x0 + blockIdx.x * DTW, ty0 = db.y0 + blockIdx.y * DTH;
    const int ix0 = 2 * tx0 - 1, iy0 = 2 * ty0 - 1;
    for (int i = tid; i < (2 * DTH + 2) * (2 * DTW + 2); i += 256) {
        const int r = i / (2 * DTW + 2), c = i - r * (2 * DTW + 2);
        s_in[r][c] = src_at<true>(src, sb, zc, ix0 + c, iy0 + r);
    }
    __syncthreads();
    for (int i = tid; i < (2 * DTH + 2) * DTW; i += 256) {
        const int r = i / DTW, xo = i - r * DTW;
        const float *q = &s_in[r][2 * xo];
        s_dx[r][xo] = lb_down4(q[0], q[1], q[2], q[3]);
    }
    __syncthreads();
    for (int i = tid; i < DTH * DTW; i += 256) {
        const int yo = i / DTW, xo = i - yo * DTW;
        const int xi = blockIdx.x * DTW + xo, yi = blockIdx.y * DTH + yo;
        if (xi < db.w && yi < db.h)
            dst[((size_t)zc * db.h + yi) * db.w + xi] = lb_down4(s_dx[2 * yo][xo], s_dx[2 * yo + 1][xo], s_dx[2 * yo + 2][xo], s_dx[2 * yo + 3][xo]);
    }
}

// lb_down32: lb_down with a thread owning ONE source column (its clamped offset computed once) and a row pair per step: all 17 + 1
// loads of a thread are requested before the first is written to LDS, a load costs five full-rate integer instructions (lb_down:
// a division by 130, two clamps and a 64-bit index per element — the launch was bound by that arithmetic, not by its 10 MB).
template<bool SRC_CLAMP>
__global__ __launch_bounds__(256) void lb_down32(const float *__restrict__ src, Box sb, float *__restrict__ dst, Box db) {
    __shared__ float s_in[2 * DTH + 2][2 * DTW + 3];
    __shared__ float s_dx[2 * DTH + 2][DTW + 1];
    static_assert(2 * DTW == 128 && (2 * DTH + 2) % 2 == 0 && 2 * (2 * DTH + 2) <= 256, "lb_down32: thread layout");
    const int tid = threadIdx.x, zc = blockIdx.z;
    const int tx0 = db.x0 + blockIdx.x * DTW, ty0 = db.y0 + blockIdx.y * DTH;
    const int ix0 = 2 * tx0 - 1, iy0 = 2 * ty0 - 1;
    const float *const f = src + (size_t)zc * sb.h * sb.w;
    {
        constexpr int NR = (2 * DTH + 2) / 2;
        const int cl = tid & 127, rh = tid >> 7;
        const uint32_t cxo = (uint32_t)(dev::clampi(ix0 + cl, sb.x0, sb.x0 + sb.w - 1) - sb.x0);
        float v[NR];
#pragma unroll
        for (int it = 0; it < NR; it++) {
            const int cy = dev::clampi(iy0 + 2 * it + rh, sb.y0, sb.y0 + sb.h - 1) - sb.y0;
            v[it] = f[lb_mul24((uint32_t)cy, (uint32_t)sb.w) + cxo];
        }
        // columns 128, 129 of the window: one element each for the first 2 (2 DTH + 2) threads
        const int er = min(tid >> 1, 2 * DTH + 1), ec = 2 * DTW + (tid & 1);
        const float ve = f[lb_clamped_off(sb, ix0 + ec, iy0 + er)];
#pragma unroll
        for (int it = 0; it < NR; it++) s_in[2 * it + rh][cl] = v[it];
        if (tid < 2 * (2 * DTH + 2)) s_in[er][ec] = ve;
    }
    __syncthreads();
    for (int i = tid; i < (2 * DTH + 2) * DTW; i += 256) {
        const int r = i / DTW, xo = i - r * DTW;
        const float *q = &s_in[r][2 * xo];
        s_dx[r][xo] = lb_down4(q[0], q[1], q[2], q[3]);
    }
    __syncthreads();
    float *const o = dst + (size_t)zc * db.h * db.w;
    for (int i = tid; i < DTH * DTW; i += 256) {
        const int yo = i / DTW, xo = i - yo * DTW;
        const int xi = blockIdx.x * DTW + xo, yi = blockIdx.y * DTH + yo;
        if (xi < db.w && yi < db.h)
            o[lb_mul24((uint32_t)yi, (uint32_t)db.w) + (uint32_t)xi] = lb_down4(s_dx[2 * yo][xo], s_dx[2 * yo + 1][xo], s_dx[2 * yo + 2][xo], s_dx[2 * yo + 3][xo]);
    }
}

// upsample(f)(x, y) (:288-294) of a pull level stored on box b
__device__ __forceinline__ float up_at(const float *__restrict__ f, const Box &b, int zc, int x, int y) {
    const int xa = (x >> 1) - 1 + 2 * (x & 1), xb = x >> 1, ya = (y >> 1) - 1 + 2 * (y & 1), yb = y >> 1;
    const float *p = f + (size_t)zc * b.h * b.w;
    const float *ra = p + (size_t)(ya - b.y0) * b.w - b.x0, *rb = p + (size_t)(yb - b.y0) * b.w - b.x0;
    const float ua = 0.25f * ra[xa] + 0.75f * ra[xb];
    const float ub = 0.25f * rb[xa] + 0.75f * rb[xb];
    return 0.25f * ua + 0.75f * ub;
}

// pull[i] on P_i: TOP: pull[7] = push[7]; else lerp(upsample(pull[i+1]), push[i], 0.5)
template<bool TOP>
__global__ __launch_bounds__(256) void lb_pull(const float *__restrict__ push, Box pb, const float *__restrict__ coarse, Box cb,
                                              float *__restrict__ dst, Box db) {
    const int xi = blockIdx.x * 64 + (threadIdx.x & 63), yi = blockIdx.y * 4 + (threadIdx.x >> 6), y = db.y0 + yi, zc = blockIdx.z;
    if (xi >= db.w || yi >= db.h) return;
    const int x = db.x0 + xi;
    const float p = src_at<true>(push, pb, zc, x, y);
    float v = p;
    if (!TOP) v = dev::lerpf(up_at(coarse, cb, zc, x, y), p, 0.5f);
    dst[((size_t)zc * db.h + (y - db.y0)) * db.w + xi] = v;
}

// pull[1] on P_1 in ONE launch from pull[4] and push[3], [2], [1]: a workgroup owns a 64 x 16 tile of one plane of pull[1] and
// makes the cells of pull[3] and pull[2] its tile reads (at most 21 x 9 and 35 x 11: P_i is by construction what level i - 1
// reads of level i) in LDS with lb_pull's expressions — their only reader is the level below, so neither level is stored, and
// two launches of a few microseconds of work each go (lb_pull:3, lb_pull:2).  Cells in a tile's halo are recomputed by its
// neighbours with identical operations.
constexpr int PMW = 64, PMH = 16, PM2W = PMW / 2 + 3, PM2H = PMH / 2 + 3, PM3W = PM2W / 2 + 4, PM3H = PM2H / 2 + 4;
__device__ __forceinline__ float up_lds(const float *s, int pitch, int wx0, int wy0, int x, int y) {   // upsample (:288-294) of a window in LDS
    const int xa = (x >> 1) - 1 + 2 * (x & 1) - wx0, xb = (x >> 1) - wx0, ya = (y >> 1) - 1 + 2 * (y & 1) - wy0, yb = (y >> 1) - wy0;
    const float ua = 0.25f * s[ya * pitch + xa] + 0.75f * s[ya * pitch + xb];
    const float ub = 0.25f * s[yb * pitch + xa] + 0.75f * s[yb * pitch + xb];
    return 0.25f * ua + 0.75f * ub;
}
__global__ __launch_bounds__(256) void lb_pull_multi(const float *__restrict__ push1, Box pb1, const float *__restrict__ push2, Box pb2,
                                                    const float *__restrict__ push3, Box pb3, const float *__restrict__ pull4, Box p4,
                                                    float *__restrict__ pull1, Box p1) {
    __shared__ float s2[PM2H * PM2W], s3[PM3H * PM3W];
    const int tid = threadIdx.x, zc = blockIdx.z;
    const int x0 = p1.x0 + blockIdx.x * PMW, x1 = min(x0 + PMW, p1.x0 + p1.w) - 1;
    const int y0 = p1.y0 + blockIdx.y * PMH, y1 = min(y0 + PMH, p1.y0 + p1.h) - 1;
    const int ax0 = (x0 >> 1) - 1, ay0 = (y0 >> 1) - 1, w2 = (x1 >> 1) + 1 - ax0 + 1, h2 = (y1 >> 1) + 1 - ay0 + 1;                     // pull[2] cells read
    const int bx0 = (ax0 >> 1) - 1, by0 = (ay0 >> 1) - 1, w3 = ((ax0 + w2 - 1) >> 1) + 1 - bx0 + 1, h3 = ((ay0 + h2 - 1) >> 1) + 1 - by0 + 1;   // pull[3]
    // every push value a thread will need is requested before the first phase (as three dependent phases of load -> use,
    // a workgroup waited out three memory round trips in a row)
    constexpr int N2 = (PM2H * PM2W + 255) / 256, N1 = PMH / 4;
    float q2[N2], q1[N1];
#pragma unroll
    for (int k = 0; k < N2; k++) {
        const int i = min(tid + 256 * k, w2 * h2 - 1), yy = i / w2, xx = i - yy * w2;
        q2[k] = src_at<true>(push2, pb2, zc, ax0 + xx, ay0 + yy);
    }
    const int x = min(x0 + (tid & 63), x1);
#pragma unroll
    for (int k = 0; k < N1; k++) q1[k] = src_at<true>(push1, pb1, zc, x, min(y0 + (tid >> 6) + 4 * k, y1));
    for (int i = tid; i < w3 * h3; i += 256) {
        const int yy = i / w3, xx = i - yy * w3, x3 = bx0 + xx, y3 = by0 + yy;
        s3[yy * PM3W + xx] = dev::lerpf(up_at(pull4, p4, zc, x3, y3), src_at<true>(push3, pb3, zc, x3, y3), 0.5f);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < N2; k++) {
        const int i = tid + 256 * k;
        if (i < w2 * h2) {
            const int yy = i / w2, xx = i - yy * w2;
            s2[yy * PM2W + xx] = dev::lerpf(up_lds(s3, PM3W, bx0, by0, ax0 + xx, ay0 + yy), q2[k], 0.5f);
        }
    }
    __syncthreads();
    if (x0 + (tid & 63) > x1) return;
#pragma unroll
    for (int k = 0; k < N1; k++) {
        const int y = y0 + (tid >> 6) + 4 * k;
        if (y > y1) break;
        pull1[((size_t)zc * p1.h + (y - p1.y0)) * p1.w + (x - p1.x0)] = dev::lerpf(up_lds(s2, PM2W, ax0, ay0, x, y), q1[k], 0.5f);
    }
}

// lb_pull_multi32: the same launch with the index arithmetic it takes to get there cut down.  lb_pull_multi is 659 VALU
// instructions per thread of which 99 are the float operations of the algorithm; 108 of the rest are quarter-rate 32 / 64-bit integer
// multiplies (element indices as size_t, divisions by the run-time tile widths).  Here: the plane's base pointer is uniform and every
// access is base + 32-bit element offset built from 24-bit multiplies (the host checks that every level's plane is below 2^29
// elements and 2^23 wide / high); cells are numbered on the FIXED pitches PM2W / PM3W (division by a constant = one 24-bit multiply,
// cells past the tile's own w2 x h2 / w3 x h3 idle); the four taps of an upsample are one offset and two +-1 / +-pitch steps
// (xa = xb +- 1 by the parity of x, :288-294).  Same float expressions in the same order as lb_pull_multi (= lb_pull's).
struct LbUp {   // the four taps of upsample(f)(x, y): offsets of f(yb, xb) and the steps to xa / ya
    uint32_t o;
    int dx, dy;
};
__device__ __forceinline__ LbUp lb_up_taps(int x, int y, int x0, int y0, int pitch) {
    LbUp t;
    t.o = lb_mul24((uint32_t)((y >> 1) - y0), (uint32_t)pitch) + (uint32_t)((x >> 1) - x0);
    t.dx = (x & 1) ? 1 : -1, t.dy = (y & 1) ? pitch : -pitch;
    return t;
}
template<typename F>
__device__ __forceinline__ float lb_up_from(const LbUp &t, F f) {   // f(offset) -> value; the operation order of up_at / up_lds
    const uint32_t oa = t.o + (uint32_t)t.dy;
    const float ua = 0.25f * f(oa + (uint32_t)t.dx) + 0.75f * f(oa);
    const float ub = 0.25f * f(t.o + (uint32_t)t.dx) + 0.75f * f(t.o);
    return 0.25f * ua + 0.75f * ub;
}
__global__ __launch_bounds__(256) void lb_pull_multi32(const float *__restrict__ push1, Box pb1, const float *__restrict__ push2, Box pb2,
                                                      const float *__restrict__ push3, Box pb3, const float *__restrict__ pull4, Box p4,
                                                      float *__restrict__ pull1, Box p1) {
    __shared__ float s2[PM2H * PM2W], s3[PM3H * PM3W];
    const int tid = threadIdx.x, zc = blockIdx.z;
    const int x0 = p1.x0 + blockIdx.x * PMW, x1 = min(x0 + PMW, p1.x0 + p1.w) - 1;
    const int y0 = p1.y0 + blockIdx.y * PMH, y1 = min(y0 + PMH, p1.y0 + p1.h) - 1;
    const int ax0 = (x0 >> 1) - 1, ay0 = (y0 >> 1) - 1, w2 = (x1 >> 1) + 1 - ax0 + 1, h2 = (y1 >> 1) + 1 - ay0 + 1;                     // pull[2] cells read
    const int bx0 = (ax0 >> 1) - 1, by0 = (ay0 >> 1) - 1, w3 = ((ax0 + w2 - 1) >> 1) + 1 - bx0 + 1, h3 = ((ay0 + h2 - 1) >> 1) + 1 - by0 + 1;   // pull[3]
    // the planes of this workgroup (uniform pointers)
    const float *const f1 = push1 + (size_t)zc * pb1.h * pb1.w, *const f2 = push2 + (size_t)zc * pb2.h * pb2.w;
    const float *const f3 = push3 + (size_t)zc * pb3.h * pb3.w, *const f4 = pull4 + (size_t)zc * p4.h * p4.w;
    float *const o1 = pull1 + (size_t)zc * p1.h * p1.w;
    constexpr int N2 = (PM2H * PM2W + 255) / 256, N1 = PMH / 4;
    float q2[N2], q1[N1];
    int c2x[N2], c2y[N2];
    bool on2[N2];
#pragma unroll
    for (int k = 0; k < N2; k++) {
        const uint32_t i = (uint32_t)min(tid + 256 * k, PM2H * PM2W - 1), yy = lb_divc<PM2W, PM2H * PM2W>(i), xx = i - yy * PM2W;
        on2[k] = tid + 256 * k < PM2H * PM2W && (int)xx < w2 && (int)yy < h2;
        c2x[k] = (int)xx, c2y[k] = (int)yy;
        q2[k] = f2[lb_clamped_off(pb2, ax0 + (int)xx, ay0 + (int)yy)];
    }
    const int x = min(x0 + (tid & 63), x1);
#pragma unroll
    for (int k = 0; k < N1; k++) q1[k] = f1[lb_clamped_off(pb1, x, min(y0 + (tid >> 6) + 4 * k, y1))];
    {
        const uint32_t i = (uint32_t)min(tid, PM3H * PM3W - 1), yy = lb_divc<PM3W, PM3H * PM3W>(i), xx = i - yy * PM3W;
        const int x3 = bx0 + (int)xx, y3 = by0 + (int)yy;
        if (tid < PM3H * PM3W && (int)xx < w3 && (int)yy < h3) {
            const LbUp t = lb_up_taps(x3, y3, p4.x0, p4.y0, p4.w);
            const float up = lb_up_from(t, [&](uint32_t o) { return f4[o]; });
            s3[yy * PM3W + xx] = dev::lerpf(up, f3[lb_clamped_off(pb3, x3, y3)], 0.5f);
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < N2; k++) {
        if (on2[k]) {
            const LbUp t = lb_up_taps(ax0 + c2x[k], ay0 + c2y[k], bx0, by0, PM3W);
            const float up = lb_up_from(t, [&](uint32_t o) { return s3[o]; });
            s2[c2y[k] * PM2W + c2x[k]] = dev::lerpf(up, q2[k], 0.5f);
        }
    }
    __syncthreads();
    if (x0 + (tid & 63) > x1) return;
#pragma unroll
    for (int k = 0; k < N1; k++) {
        const int y = y0 + (tid >> 6) + 4 * k;
        if (y > y1) break;
        const LbUp t = lb_up_taps(x, y, ax0, ay0, PM2W);
        const float up = lb_up_from(t, [&](uint32_t o) { return s2[o]; });
        o1[lb_mul24((uint32_t)(y - p1.y0), (uint32_t)p1.w) + (uint32_t)(x - p1.x0)] = dev::lerpf(up, q1[k], 0.5f);
    }
}

// The planes of the two pyramids never mix (down- and up-sampling are per plane), so below some level one workgroup can take
// a plane all the way down and back up with a barrier between levels — its own stores are visible to it after the barrier —
// instead of one launch per level: levels `from`..7 of the push pyramid, then levels 7..`from` of the pull pyramid.
struct TailArgs {
    float *push[LV], *pull[LV];
    Box PB[LV], P[LV];
    int from;
};
__global__ __launch_bounds__(1024) void lb_tail(TailArgs a) {
    const int zc = blockIdx.x, tid = threadIdx.x;
    for (int l = a.from; l < LV; l++) {
        const Box sb = a.PB[l - 1], db = a.PB[l];
        const float *src = a.push[l - 1];
        float *dst = a.push[l];
        for (int i = tid; i < db.w * db.h; i += 1024) {
            const int yi = i / db.w, xi = i - yi * db.w, x = db.x0 + xi, y = db.y0 + yi;
            float dx[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int yy = 2 * y - 1 + k;
                dx[k] = lb_down4(src_at<true>(src, sb, zc, 2 * x - 1, yy), src_at<true>(src, sb, zc, 2 * x, yy), src_at<true>(src, sb, zc, 2 * x + 1, yy),
                                 src_at<true>(src, sb, zc, 2 * x + 2, yy));
            }
            dst[(size_t)zc * db.h * db.w + i] = lb_down4(dx[0], dx[1], dx[2], dx[3]);
        }
        __syncthreads();
    }
    for (int l = LV - 1; l >= a.from; l--) {
        const Box pb = a.PB[l], db = a.P[l];
        for (int i = tid; i < db.w * db.h; i += 1024) {
            const int yi = i / db.w, xi = i - yi * db.w, x = db.x0 + xi, y = db.y0 + yi;
            const float p = src_at<true>(a.push[l], pb, zc, x, y);
            a.pull[l][(size_t)zc * db.h * db.w + i] = (l == LV - 1) ? p : dev::lerpf(up_at(a.pull[l + 1], a.P[l + 1], zc, x, y), p, 0.5f);
        }
        __syncthreads();
    }
}

// lb_tail with the levels in LDS: when the planes of levels `from`..7 of both pyramids fit (TAIL_LDS floats) a workgroup keeps
// its plane there — the only global traffic is reading push[from - 1] and writing pull[from] (what lb_pull:from-1 reads);
// a phase then costs an LDS round trip instead of a global one.  Same expressions, same order.
constexpr int TAIL_LDS = 15360;
__global__ __launch_bounds__(1024) void lb_tail_lds(TailArgs a) {
    __shared__ float s_lv[TAIL_LDS];
    const int zc = blockIdx.x, tid = threadIdx.x;
    int opush[LV], opull[LV], off = 0;
    for (int l = a.from; l < LV; l++) opush[l] = off, off += a.PB[l].w * a.PB[l].h;
    for (int l = a.from; l < LV; l++) opull[l] = off, off += a.P[l].w * a.P[l].h;
    auto lds_at = [&](int base, const Box &b, int x, int y) -> float {   // clamped to the box, as src_at<true>
        x = dev::clampi(x, b.x0, b.x0 + b.w - 1), y = dev::clampi(y, b.y0, b.y0 + b.h - 1);
        return s_lv[base + (y - b.y0) * b.w + (x - b.x0)];
    };
    for (int l = a.from; l < LV; l++) {
        const Box sb = a.PB[l - 1], db = a.PB[l];
        const bool first = l == a.from;
        const float *src = a.push[l - 1];
        for (int i = tid; i < db.w * db.h; i += 1024) {
            const int yi = i / db.w, xi = i - yi * db.w, x = db.x0 + xi, y = db.y0 + yi;
            float dx[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int yy = 2 * y - 1 + k;
                float t[4];
#pragma unroll
                for (int j = 0; j < 4; j++)
                    t[j] = first ? src_at<true>(src, sb, zc, 2 * x - 1 + j, yy) : lds_at(opush[l - 1], sb, 2 * x - 1 + j, yy);
                dx[k] = lb_down4(t[0], t[1], t[2], t[3]);
            }
            s_lv[opush[l] + i] = lb_down4(dx[0], dx[1], dx[2], dx[3]);
        }
        __syncthreads();
    }
    for (int l = LV - 1; l >= a.from; l--) {
        const Box pb = a.PB[l], db = a.P[l];
        for (int i = tid; i < db.w * db.h; i += 1024) {
            const int yi = i / db.w, xi = i - yi * db.w, x = db.x0 + xi, y = db.y0 + yi;
            const float p = lds_at(opush[l], pb, x, y);
            float v = p;
            if (l != LV - 1) {
                const Box cb = a.P[l + 1];
                const int xa = (x >> 1) - 1 + 2 * (x & 1), xb = x >> 1, ya = (y >> 1) - 1 + 2 * (y & 1), yb = y >> 1;
                const float *ra = &s_lv[opull[l + 1] + (ya - cb.y0) * cb.w - cb.x0], *rb = &s_lv[opull[l + 1] + (yb - cb.y0) * cb.w - cb.x0];
                const float ua = 0.25f * ra[xa] + 0.75f * ra[xb];
                const float ub = 0.25f * rb[xa] + 0.75f * rb[xb];
                v = dev::lerpf(0.25f * ua + 0.75f * ub, p, 0.5f);
            }
            s_lv[opull[l] + i] = v;
            if (l == a.from) a.pull[l][(size_t)zc * db.h * db.w + i] = v;
        }
        __syncthreads();
    }
}

__device__ __forceinline__ float bokeh_radius(int depth, const LBGeom &g) {   // :88-89
    const int ad = depth - g.focus;
    return (float)(unsigned)(ad < 0 ? -ad : ad) * g.scale;
}

// depth and bokeh radius on D; `rec` packs what lb_final gathers per aperture sample into ONE word per pixel of D:
// depth | left(x, y, 0) << 8 | left(.., 1) << 16 | left(.., 2) << 24 (the bokeh radius is a function of the depth, the left
// image is edge-clamped: both are properties of the sample's position) — one 4-byte gather per sample instead of five.
__global__ __launch_bounds__(256) void lb_depth(const float *__restrict__ push0, Box E, const float *__restrict__ pull1, Box P1,
                                               const uint8_t *__restrict__ L, LBGeom g, Box D, uint32_t *__restrict__ rec,
                                               float *__restrict__ br) {
    const int xi = blockIdx.x * 256 + threadIdx.x, yi = blockIdx.y;
    if (xi >= D.w) return;
    const int x = D.x0 + xi, y = D.y0 + yi;
    int best_i = 0;
    float best = 3.402823466e38f;
    const float v1 = dev::lerpf(up_at(pull1, P1, g.slices, x, y), push0[at(E, g.slices, 0, 1, x, y)], 0.5f);   // the same for every z
    for (int z = 0; z < g.slices; z++) {
        const float v0 = dev::lerpf(up_at(pull1, P1, z, x, y), push0[at(E, g.slices, z, 0, x, y)], 0.5f);
        const float fc = v0 / v1;
        if (fc < best) best = fc, best_i = z;
    }
    const size_t o = (size_t)yi * D.w + xi;
    const long lo = (long)(dev::clampi(y, g.ly0, g.ly1) - g.ly0) * g.l_sy + (dev::clampi(x, g.lx0, g.lx1) - g.lx0);
    rec[o] = (uint32_t)best_i | (uint32_t)L[lo + g.l_c[0]] << 8 | (uint32_t)L[lo + g.l_c[1]] << 16 | (uint32_t)L[lo + g.l_c[2]] << 24;
    br[o] = bokeh_radius(best_i, g);
}

// ---- the fused front end: push[0] (one float per pixel of E per plane: 130 MB at 768 x 1280 x 33, written once and read
// twice) is never stored.  lb_cost_down makes push[1] directly — a workgroup owns a strip of push[1] columns and a segment of
// its rows and walks the source rows: every thread computes the cost stack of one source pixel (the right-image row segment
// and the next row's bytes are staged / prefetched as in lb_cost), the row of all planes goes through LDS, the 1-3-3-1 pass
// in x is taken there and the pass in y on rolling registers — the generator's expressions (:279-285) operand for operand.
// lb_depth_rc recomputes the cost stack of its own pixel (the same device function, so the same bits) instead of reading it.
constexpr int CD_RP = 260;   // floats per plane row in LDS (256 source pixels + pad)
template<int SB, bool FULL>
__global__ __launch_bounds__(256) void lb_cost_down(const uint8_t *__restrict__ L, const uint8_t *__restrict__ Rr, LBGeom g,
                                                   float *__restrict__ dst, Box db, int NDX, int NDY) {
    extern __shared__ __align__(16) float s_row[];   // [slices + 1][CD_RP]
    __shared__ uint2 sr[2][256 + 2 * LB_MAXS];
    constexpr int KPT = ((SB + 1) * 127 + 255) / 256;
    const int tid = threadIdx.x, zc = g.slices + 1;
    const int dx0 = db.x0 + blockIdx.x * NDX, dy0 = db.y0 + blockIdx.y * NDY;
    const int ndx = min(NDX, db.x0 + db.w - dx0), ndy = min(NDY, db.y0 + db.h - dy0);
    const int sx0 = 2 * dx0 - 1, sy0 = 2 * dy0 - 1, nsx = 2 * ndx + 2, nrows = 2 * ndy + 2, nr = nsx + 2 * g.slices;
    const int total = zc * ndx;
    // the (plane, column) pairs of this thread in the x pass: LDS offset of the first tap, offset in a destination row
    int qoff[KPT], doff[KPT];
    float p0[KPT], p1[KPT], p2[KPT];
#pragma unroll
    for (int k = 0; k < KPT; k++) {
        const int idx = tid + 256 * k, p = idx < total ? idx / ndx : 0, xl = idx < total ? idx - p * ndx : 0;   // slots past `total` read LDS offset 0
        qoff[k] = p * CD_RP + 2 * xl;
        doff[k] = idx < total ? (int)((size_t)p * db.h * db.w) + (dx0 - db.x0) + xl : -1;
        p0[k] = p1[k] = p2[k] = 0.0f;
    }
    const int lxo = dev::clampi(sx0 + tid, g.lx0, g.lx1) - g.lx0;
    const int rxo0 = dev::clampi(sx0 + tid, g.rx0, g.rx1) - g.rx0, rxo1 = dev::clampi(sx0 + tid + 256, g.rx0, g.rx1) - g.rx0;
    uint8_t nl[3], nr0[3], nr1[3];
    auto fetch = [&](int j) {   // the bytes of source row j of the segment: all loads unconditional, indices clamped
        const int y = sy0 + j;
        const long lo = (long)(dev::clampi(y, g.ly0, g.ly1) - g.ly0) * g.l_sy, ro = (long)(dev::clampi(y, g.ry0, g.ry1) - g.ry0) * g.r_sy;
#pragma unroll
        for (int c = 0; c < 3; c++) nl[c] = L[lo + lxo + g.l_c[c]], nr0[c] = Rr[ro + rxo0 + g.r_c[c]], nr1[c] = Rr[ro + rxo1 + g.r_c[c]];
    };
    auto stage = [&](int b) {
        sr[b][tid] = lb_pack(nr0[0], nr0[1], nr0[2]);
        if (tid + 256 < nr) sr[b][tid + 256] = lb_pack(nr1[0], nr1[1], nr1[2]);
    };
    fetch(0);
    stage(0);
    CostRow cur = {(uint32_t)nl[0] | (uint32_t)nl[1] << 16, (int)nl[2]};
    __syncthreads();
    for (int j = 0; j < nrows; j++) {
        if (j + 1 < nrows) fetch(j + 1);
        if (tid < nsx) {
            float cz[SB];
            const float conf = cost_stack<SB, FULL>(sr[j & 1], tid, cur, g, cz);
#pragma unroll
            for (int z = 0; z < SB; z++)
                if (FULL || z < g.slices) s_row[z * CD_RP + tid] = cz[z] * conf;
            s_row[g.slices * CD_RP + tid] = conf;
        }
        __syncthreads();
        // y pass on three registers per slot, no rotation: on an even row (tap 0 of an output row, tap 2 of the one before)
        // t = a + 3 (b + d), a = d; on an odd row (tap 3 / tap 1) the output (t + d) / 8 is complete, b = d
        if (j & 1) {
            const bool emit = j >= 3;
            float *drow = dst + (size_t)(dy0 - db.y0 + ((j - 3) >> 1)) * db.w;
#pragma unroll
            for (int k = 0; k < KPT; k++) {   // slots past the end read LDS offset 0 and store nothing
                const float2 qa = *(const float2 *)&s_row[qoff[k]], qb = *(const float2 *)&s_row[qoff[k] + 2];
                const float d = lb_down4(qa.x, qa.y, qb.x, qb.y);
                if (emit && doff[k] >= 0) drow[doff[k]] = (p2[k] + d) * 0.125f;
                p1[k] = d;
            }
        } else {
#pragma unroll
            for (int k = 0; k < KPT; k++) {
                const float2 qa = *(const float2 *)&s_row[qoff[k]], qb = *(const float2 *)&s_row[qoff[k] + 2];
                const float d = lb_down4(qa.x, qa.y, qb.x, qb.y);
                p2[k] = dev::mad(3.0f, p1[k] + d, p0[k]);
                p0[k] = d;
            }
        }
        if (j + 1 < nrows) {
            stage((j + 1) & 1);
            cur = {(uint32_t)nl[0] | (uint32_t)nl[1] << 16, (int)nl[2]};
        }
        __syncthreads();
    }
}

// lb_cost_down2: the same walk, TWO source rows per step by 512 threads (threads 0..255 the even row, 256..511 the odd one).
// lb_cost_down holds a pixel's cost stack AND the y-pass state of 17 (plane, column) slots per thread: 192 registers, two
// waves per SIMD, and the VALU idles a third of the time behind LDS round trips.  With twice the threads a thread keeps 9
// slots, a step makes one even and one odd row (exactly one `t = a + 3 (b + d)` and one output per slot), and a CU holds
// sixteen waves.
template<int SB, bool FULL>
__global__ __launch_bounds__(512) void lb_cost_down2(const uint8_t *__restrict__ L, const uint8_t *__restrict__ Rr, LBGeom g,
                                                    float *__restrict__ dst, Box db, int NDX, int NDY) {
    extern __shared__ __align__(16) float s_row[];   // [2 rows][slices + 1][CD_RP]
    __shared__ uint2 sr[2][2][256 + 2 * SB];         // [buffer][row of the pair][pixel] (sized by SB: two workgroups per CU at 32)
    constexpr int KPT = ((SB + 1) * 127 + 511) / 512;
    const int tid = threadIdx.x, t = tid & 255, half = tid >> 8, zc = g.slices + 1;
    const int dx0 = db.x0 + blockIdx.x * NDX, dy0 = db.y0 + blockIdx.y * NDY;
    const int ndx = min(NDX, db.x0 + db.w - dx0), ndy = min(NDY, db.y0 + db.h - dy0);
    const int sx0 = 2 * dx0 - 1, sy0 = 2 * dy0 - 1, nsx = 2 * ndx + 2, nsteps = ndy + 1, nr = nsx + 2 * g.slices;
    const int total = zc * ndx, rowsz = zc * CD_RP;
    int qoff[KPT], doff[KPT];
    float p0[KPT], p1[KPT], p2[KPT];
#pragma unroll
    for (int k = 0; k < KPT; k++) {
        const int idx = tid + 512 * k, p = idx < total ? idx / ndx : 0, xl = idx < total ? idx - p * ndx : 0;
        qoff[k] = p * CD_RP + 2 * xl;
        doff[k] = idx < total ? (int)((size_t)p * db.h * db.w) + (dx0 - db.x0) + xl : -1;
        p0[k] = p1[k] = p2[k] = 0.0f;
    }
    const int lxo = dev::clampi(sx0 + t, g.lx0, g.lx1) - g.lx0;
    const int rxo0 = dev::clampi(sx0 + t, g.rx0, g.rx1) - g.rx0, rxo1 = dev::clampi(sx0 + t + 256, g.rx0, g.rx1) - g.rx0;
    uint8_t nl[3], nr0[3], nr1[3];
    auto fetch = [&](int step) {   // this thread's row of the pair: source row 2 step + half of the segment
        const int y = sy0 + 2 * step + half;
        const long lo = (long)(dev::clampi(y, g.ly0, g.ly1) - g.ly0) * g.l_sy, ro = (long)(dev::clampi(y, g.ry0, g.ry1) - g.ry0) * g.r_sy;
#pragma unroll
        for (int c = 0; c < 3; c++) nl[c] = L[lo + lxo + g.l_c[c]], nr0[c] = Rr[ro + rxo0 + g.r_c[c]], nr1[c] = Rr[ro + rxo1 + g.r_c[c]];
    };
    auto stage = [&](int b) {
        sr[b][half][t] = lb_pack(nr0[0], nr0[1], nr0[2]);
        if (t + 256 < nr) sr[b][half][t + 256] = lb_pack(nr1[0], nr1[1], nr1[2]);
    };
    fetch(0);
    stage(0);
    CostRow cur = {(uint32_t)nl[0] | (uint32_t)nl[1] << 16, (int)nl[2]};
    __syncthreads();
    for (int st = 0; st < nsteps; st++) {
        if (st + 1 < nsteps) fetch(st + 1);
        if (t < nsx) {
            float cz[SB];
            const float conf = cost_stack<SB, FULL>(sr[st & 1][half], t, cur, g, cz);
            float *row = s_row + half * rowsz;
#pragma unroll
            for (int z = 0; z < SB; z++)
                if (FULL || z < g.slices) row[z * CD_RP + t] = cz[z] * conf;
            row[g.slices * CD_RP + t] = conf;
        }
        __syncthreads();
        {
            // the even row (tap 0 of an output row, tap 2 of the one before): t = a + 3 (b + d), a = d; then the odd row
            // (tap 3 / tap 1): the output (t + d) / 8 is complete, b = d
            const bool emit = st >= 1;
            float *drow = dst + (size_t)(dy0 - db.y0 + st - 1) * db.w;
#pragma unroll
            for (int k = 0; k < KPT; k++) {   // slots past the end read LDS offset 0 and store nothing
                const float2 ea = *(const float2 *)&s_row[qoff[k]], eb = *(const float2 *)&s_row[qoff[k] + 2];
                const float2 oa = *(const float2 *)&s_row[rowsz + qoff[k]], ob = *(const float2 *)&s_row[rowsz + qoff[k] + 2];
                const float de = lb_down4(ea.x, ea.y, eb.x, eb.y);
                const float dd = lb_down4(oa.x, oa.y, ob.x, ob.y);
                p2[k] = dev::mad(3.0f, p1[k] + de, p0[k]);
                p0[k] = de;
                if (emit && doff[k] >= 0) drow[doff[k]] = (p2[k] + dd) * 0.125f;
                p1[k] = dd;
            }
        }
        if (st + 1 < nsteps) {
            stage((st + 1) & 1);
            cur = {(uint32_t)nl[0] | (uint32_t)nl[1] << 16, (int)nl[2]};
        }
        __syncthreads();
    }
}

template<int SB, bool FULL>
__global__ __launch_bounds__(256) void lb_depth_rc(const uint8_t *__restrict__ L, const uint8_t *__restrict__ Rr, const float *__restrict__ pull1, Box P1,
                                                  LBGeom g, Box D, uint32_t *__restrict__ rec, float *__restrict__ br) {
    // a wave per row segment of 64 pixels, four rows per workgroup: narrow segments waste few lanes at the right edge
    __shared__ uint2 srw[4][64 + 2 * LB_MAXS];
    const int tid = threadIdx.x & 63, wv = threadIdx.x >> 6, xi = blockIdx.x * 64 + tid, yi = min((int)blockIdx.y * 4 + wv, D.h - 1);
    const int x = D.x0 + xi, y = D.y0 + yi;
    const uint2 *sr = srw[wv];
    {
        const long ro = (long)(dev::clampi(y, g.ry0, g.ry1) - g.ry0) * g.r_sy;
        const int xb = D.x0 + blockIdx.x * 64;
        for (int i = tid; i < 64 + 2 * g.slices; i += 64) {
            const long o = ro + (dev::clampi(xb + i, g.rx0, g.rx1) - g.rx0);
            srw[wv][i] = lb_pack(Rr[o + g.r_c[0]], Rr[o + g.r_c[1]], Rr[o + g.r_c[2]]);
        }
    }
    __syncthreads();
    if (xi >= D.w || (int)blockIdx.y * 4 + wv >= D.h) return;
    const long lo = (long)(dev::clampi(y, g.ly0, g.ly1) - g.ly0) * g.l_sy + (dev::clampi(x, g.lx0, g.lx1) - g.lx0);
    const int c0 = L[lo + g.l_c[0]], c1 = L[lo + g.l_c[1]], c2 = L[lo + g.l_c[2]];
    const CostRow c = {(uint32_t)c0 | (uint32_t)c1 << 16, c2};
    float cz[SB];
    const float conf = cost_stack<SB, FULL>(sr, tid, c, g, cz);
    int best_i = 0;
    float best = 3.402823466e38f;
    // upsample(pull[1]) (:288-294): the four taps are the same elements of every plane — 32-bit element offsets per lane, the
    // plane base stays uniform
    const int xa = (x >> 1) - 1 + 2 * (x & 1), xb = x >> 1, ya = (y >> 1) - 1 + 2 * (y & 1), yb = y >> 1;
    // (BYTE offsets: a zero-extended 32-bit lane offset on a uniform base is one load instruction, no address arithmetic)
    const uint32_t oaa = 4u * (uint32_t)((ya - P1.y0) * P1.w + (xa - P1.x0)), oab = 4u * (uint32_t)((ya - P1.y0) * P1.w + (xb - P1.x0)),
                   oba = 4u * (uint32_t)((yb - P1.y0) * P1.w + (xa - P1.x0)), obb = 4u * (uint32_t)((yb - P1.y0) * P1.w + (xb - P1.x0));
    const size_t plane = (size_t)P1.h * P1.w;
    struct Taps {
        float aa, ab, ba, bb;
    };
    auto taps = [&](int zc) -> Taps {
        const char *__restrict__ pz = (const char *)(pull1 + (size_t)zc * plane);
        return {*(const float *)(pz + oaa), *(const float *)(pz + oab), *(const float *)(pz + oba), *(const float *)(pz + obb)};
    };
    auto up = [&](const Taps &t) -> float {
        const float ua = 0.25f * t.aa + 0.75f * t.ab;
        const float ub = 0.25f * t.ba + 0.75f * t.bb;
        return 0.25f * ua + 0.75f * ub;
    };
    const float v1 = dev::lerpf(up(taps(g.slices)), conf, 0.5f);   // the same for every z
    constexpr int ZB = 8;   // the taps of ZB slices are in flight while the ZB before them are divided and compared
    Taps nx[ZB];
#pragma unroll
    for (int i = 0; i < ZB; i++) nx[i] = taps(FULL ? i : min(i, g.slices - 1));
#pragma unroll
    for (int zb = 0; zb < SB; zb += ZB) {
        Taps cur[ZB];
#pragma unroll
        for (int i = 0; i < ZB; i++) cur[i] = nx[i];
        if (zb + ZB < SB) {
#pragma unroll
            for (int i = 0; i < ZB; i++) nx[i] = taps(FULL ? zb + ZB + i : min(zb + ZB + i, g.slices - 1));
        }
#pragma unroll
        for (int i = 0; i < ZB; i++) {
            const int z = zb + i;
            if (FULL || z < g.slices) {
                const float v0 = dev::lerpf(up(cur[i]), cz[z] * conf, 0.5f);
                const float fc = v0 / v1;
                if (fc < best) best = fc, best_i = z;
            }
        }
    }
    const size_t o = (size_t)yi * D.w + xi;
    rec[o] = (uint32_t)best_i | (uint32_t)c0 << 8 | (uint32_t)c1 << 16 | (uint32_t)c2 << 24;
    br[o] = bokeh_radius(best_i, g);
}

__global__ __launch_bounds__(256) void lb_wcy(const float *__restrict__ br, Box D, int R, int oy0, int oh, float *__restrict__ wcy) {
    const int xi = blockIdx.x * 256 + threadIdx.x, yo = blockIdx.y;   // yo: output row
    if (xi >= D.w) return;
    const int y = oy0 + yo;
    float m = -INFINITY;
    // eight rows requested at a time (a run-time trip count one load at a time pays the round trip per row); rows past the
    // window repeat its last row: the maximum does not change
    const float *col = br + (size_t)(y - R - D.y0) * D.w + xi;
    for (int r0 = 0; r0 <= 2 * R; r0 += 8) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = col[(size_t)min(r0 + j, 2 * R) * D.w];
#pragma unroll
        for (int j = 0; j < 8; j++) m = fmaxf(m, v[j]);
    }
    wcy[(size_t)yo * D.w + xi] = m;
}

// random_float() of the reference: src/Random.cpp:20-104 with args {id, tag, s, y, x}
__device__ __forceinline__ uint32_t rng32(uint32_t x) { return ((1040796640u * x) + 1121052041u) * x + 576942909u; }
// the last step of the hash and the conversion to [0, 1): `ry` = the state after the sample index and the row went in
__device__ __forceinline__ float rand_float_x(uint32_t ry, int x) {
    uint32_t r = rng32(ry + (uint32_t)x);
    r = r ^ (r >> 16);
    return dev::clampf(__uint_as_float((127u << 23) | (r >> 9)) - 1.0f, 0.0f, 1.0f);
}

// WCY: `wcy` is the bokeh-radius plane itself and the workgroup makes the 256 + 2 R column maxima over rows y - R .. y + R that
// its pixels' windows read (lb_wcy's values — a maximum has no rounding and no order) in LDS instead of reading them from a plane an
// lb_wcy launch wrote: 2 R + 1 loads per column from a 4 MB plane that sits in L2 against a launch of its own in the chain.
constexpr int LB_MAXR = 64;   // generator :14-19: (slices - focus or focus) x scale <= 64
template<bool WCY>
__global__ __launch_bounds__(256) void lb_final(LBGeom g, const uint32_t *__restrict__ rec, const float *__restrict__ wcy, Box D, int ox0, int oy0,
                                               int ow, int nc, float *__restrict__ out, long out_sy, long out_sc) {
    // the hash consumes (id, tag, sample, y, x) in that order: everything up to the row is the same for the whole workgroup
    // The last step hashes t = ry + x with the quadratic rng32(t) = C1 t^2 + C2 t + C3 (mod 2^32).  With t = t0 + l (t0: the
    // state plus the workgroup's first x, l: the lane's offset) that is rng32(t0) + l (2 C1 t0 + C2) + C1 l^2: the first two
    // coefficients are per (sample, row) — computed once per workgroup —, the last per lane: one 32-bit multiply per hash
    // instead of two, the same residue.
    __shared__ uint2 s_ru[LB_MAXS], s_rv[LB_MAXS];
    const int xo = blockIdx.x * 256 + threadIdx.x, yo = blockIdx.y;
    const int x = ox0 + xo, y = oy0 + yo;
    if ((int)threadIdx.x < g.samples) {
        const uint32_t seed_u = rng32(rng32(0u) + (uint32_t)g.tag), seed_v = rng32(rng32(1u) + (uint32_t)g.tag);
        const uint32_t xb = (uint32_t)(ox0 + (int)blockIdx.x * 256);
        const uint32_t tu = rng32(rng32(seed_u + threadIdx.x) + (uint32_t)y) + xb, tv = rng32(rng32(seed_v + threadIdx.x) + (uint32_t)y) + xb;
        s_ru[threadIdx.x] = make_uint2(rng32(tu), 2u * 1040796640u * tu + 1121052041u);
        s_rv[threadIdx.x] = make_uint2(rng32(tv), 2u * 1040796640u * tv + 1121052041u);
    }
    const uint32_t ln = threadIdx.x, kl = 1040796640u * ln * ln;
    auto rand_lane = [&](const uint2 ab) -> float {
        uint32_t r = ab.x + ln * ab.y + kl;
        r = r ^ (r >> 16);
        return dev::clampf(__uint_as_float((127u << 23) | (r >> 9)) - 1.0f, 0.0f, 1.0f);
    };
    __shared__ float s_w[WCY ? 256 + 2 * LB_MAXR : 1];
    if (WCY) {
        const int c0 = ox0 + (int)blockIdx.x * 256 - g.R - D.x0;   // first column of the workgroup's windows, relative to D
        for (int c = threadIdx.x; c < 256 + 2 * g.R; c += 256) {
            const float *col = wcy + (size_t)(y - g.R - D.y0) * D.w + min(c0 + c, D.w - 1);   // columns past D feed pixels past the output only
            float m = -INFINITY;
            for (int r0 = 0; r0 <= 2 * g.R; r0 += 8) {   // batched as in lb_wcy; rows past the window repeat its last row
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; j++) v[j] = col[(size_t)min(r0 + j, 2 * g.R) * D.w];
#pragma unroll
                for (int j = 0; j < 8; j++) m = fmaxf(m, v[j]);
            }
            s_w[c] = m;
        }
    }
    __syncthreads();
    if (xo >= ow) return;
    float worst = -INFINITY;
    if (WCY) {
        for (int j = 0; j <= 2 * g.R; j++) worst = fmaxf(worst, s_w[threadIdx.x + j]);
    } else {
        const float *wrow = wcy + (size_t)yo * D.w + (x - g.R - D.x0);
        for (int r0 = 0; r0 <= 2 * g.R; r0 += 8) {   // batched as in lb_wcy
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; j++) v[j] = wrow[min(r0 + j, 2 * g.R)];
#pragma unroll
            for (int j = 0; j < 8; j++) worst = fmaxf(worst, v[j]);
        }
    }
    const size_t o = (size_t)(y - D.y0) * D.w + (x - D.x0);
    const uint32_t w0 = rec[o];
    float acc[4] = {(float)((w0 >> 8) & 255u), (float)((w0 >> 16) & 255u), (float)(w0 >> 24), 255.0f};
    const int dxy = (int)(w0 & 255u);
    const float br0 = bokeh_radius(dxy, g), brs = br0 * br0;
    // Samples in batches of SBATCH: the gathers of a batch are all requested before the first one is used (one at a time,
    // every sample paid the round trip of its gather).  Samples past the count re-use the last one with weight 0: adding
    // 0.0f leaves the non-negative sums as they are.  32-bit element offsets (the host checks D fits): the row product is
    // one 24-bit multiply.
    constexpr int SBATCH = 4;
    const uint32_t *const rrow = rec + (size_t)(y - D.y0) * D.w + (x - D.x0);
    for (int s0 = 0; s0 < g.samples; s0 += SBATCH) {
        uint32_t ws[SBATCH];
        float r2[SBATCH];
#pragma unroll
        for (int j = 0; j < SBATCH; j++) {
            const int s = min(s0 + j, g.samples - 1);
            const float fu = ((rand_lane(s_ru[s]) - 0.5f) * 2.0f) * worst;
            const float fv = ((rand_lane(s_rv[s]) - 0.5f) * 2.0f) * worst;
            const int u = dev::clampi((int)fu, -g.R, g.R), v = dev::clampi((int)fv, -g.R, g.R);
            ws[j] = rrow[__mul24(v, D.w) + u];
            r2[j] = (float)(__mul24(u, u) + __mul24(v, v));
        }
#pragma unroll
        for (int j = 0; j < SBATCH; j++) {
            const int ds = (int)(ws[j] & 255u);
            const float bs = bokeh_radius(ds, g);
            const bool take = (s0 + j < g.samples) && ((r2[j] < brs) || (ds < dxy)) && (r2[j] < bs * bs);
            const float wgt = take ? 1.0f : 0.0f;
            acc[0] = acc[0] + wgt * (float)((ws[j] >> 8) & 255u);
            acc[1] = acc[1] + wgt * (float)((ws[j] >> 16) & 255u);
            acc[2] = acc[2] + wgt * (float)(ws[j] >> 24);
            acc[3] = acc[3] + wgt * 255.0f;
        }
    }
#pragma unroll
    for (int c = 0; c < 3; c++) {
        if (c < nc) out[(long)yo * out_sy + xo + (long)c * out_sc] = acc[c] / acc[3];
    }
}

const int64_t e0 = 0, e192 = 192, e320 = 320, e3 = 3;
const int64_t *const est_im[6] = {&e0, &e192, &e0, &e320, &e0, &e3};
halide_scalar_value_t mk_i(int v) { halide_scalar_value_t s{}; s.u.i32 = v; return s; }
halide_scalar_value_t mk_f(float v) { halide_scalar_value_t s{}; s.u.f32 = v; return s; }
// defaults / ranges / estimates: generator :14-19, :158-166
const halide_scalar_value_t s32 = mk_i(32), s13 = mk_i(13), s1 = mk_i(1), s64 = mk_i(64), fhalf = mk_f(0.5f), f0 = mk_f(0.0f), f1 = mk_f(1.0f);
const halide_type_t ty_u8 = {(decltype(halide_type_t::code))1, 8, 0};
const halide_type_t ty_i32 = {(decltype(halide_type_t::code))0, 32, 0};
const halide_type_t ty_f32 = {(decltype(halide_type_t::code))2, 32, 0};
const halide_filter_argument_t lb_args[7] = {
    {"left_im", halide_argument_kind_input_buffer, 3, ty_u8, nullptr, nullptr, nullptr, nullptr, est_im},
    {"right_im", halide_argument_kind_input_buffer, 3, ty_u8, nullptr, nullptr, nullptr, nullptr, est_im},
    {"slices", halide_argument_kind_input_scalar, 0, ty_i32, &s32, &s1, &s64, &s32, nullptr},
    {"focus_depth", halide_argument_kind_input_scalar, 0, ty_i32, &s13, &s1, &s32, &s13, nullptr},
    {"blur_radius_scale", halide_argument_kind_input_scalar, 0, ty_f32, &fhalf, &f0, &f1, &fhalf, nullptr},
    {"aperture_samples", halide_argument_kind_input_scalar, 0, ty_i32, &s32, &s1, &s64, &s32, nullptr},
    {"final", halide_argument_kind_output_buffer, 3, ty_f32, nullptr, nullptr, nullptr, nullptr, est_im},
};
const halide_filter_metadata_t lb_md = {1, 7, lb_args, kTargetString, "lens_blur"};

}  // namespace

extern "C" void hlmi_lens_blur_set_random_tag(int tag) { g_rand_tag.store(tag); }
extern "C" int hlmi_lens_blur_get_random_tag(void) { return g_rand_tag.load(); }

extern "C" int lens_blur(halide_buffer_t *left_im, halide_buffer_t *right_im, int32_t slices, int32_t focus_depth,
                         float blur_radius_scale, int32_t aperture_samples, halide_buffer_t *final_) {
    void *uc = nullptr;
    BufArg args[3] = {{"left_im", left_im, T_U8, 3, false}, {"right_im", right_im, T_U8, 3, false}, {"final", final_, T_F32, 3, true}};
    int r = check_not_null(uc, args, 3);
    if (r) return r;
    // scalar ranges are checked before the image checks (src/Lower.cpp:189 vs :251); generator :14-19
    struct { const char *name; double v, lo, hi; } pr[4] = {{"slices", (double)slices, 1, 64}, {"focus_depth", (double)focus_depth, 1, 32},
                                                            {"blur_radius_scale", (double)blur_radius_scale, 0, 1},
                                                            {"aperture_samples", (double)aperture_samples, 1, 64}};
    for (auto &p : pr) {
        if (!(p.v >= p.lo)) return report(uc, halide_error_code_param_too_small, "Parameter %s is %g but must be at least %g", p.name, p.v, p.lo);
        if (!(p.v <= p.hi)) return report(uc, halide_error_code_param_too_large, "Parameter %s is %g but must be at most %g", p.name, p.v, p.hi);
    }
    if ((r = check_type_and_dims(uc, args, 3))) return r;
    if (any_bounds_query(args, 3)) {
        // every tap of the inputs is clamped (:27-28); propose the output's x / y region and the three channels
        int mins[3] = {final_->dim[0].min, final_->dim[1].min, 0}, ext[3] = {final_->dim[0].extent, final_->dim[1].extent, 3};
        answer_query(left_im, mins, ext);
        answer_query(right_im, mins, ext);
        answer_query(final_, mins, ext);
        return 0;
    }
    for (int i = 0; i < 3; i++)
        if ((r = check_shape(uc, args[i]))) return r;
    if ((r = check_equal(uc, "final.min.2", final_->dim[2].min, "0", 0))) return r;
    if (final_->dim[2].extent > 3) {
        return report(uc, halide_error_code_constraint_violated, "Output buffer final has %d channels, at most 3 are defined", final_->dim[2].extent);
    }
    const int ow = final_->dim[0].extent, oh = final_->dim[1].extent, nc = final_->dim[2].extent;
    if (ow > 0 && oh > 0 && nc > 0) {
        for (int i = 0; i < 2; i++) {
            const halide_buffer_t *b = args[i].buf;
            if (b->dim[0].extent < 1 || b->dim[1].extent < 1 || b->dim[2].extent < 1) {
                return report(uc, halide_error_code_access_out_of_bounds, "Input buffer %s is empty but is accessed (clamped)", args[i].name);
            }
        }
    }
    DeviceCtx ctx;
    if ((r = acquire_device(uc, &ctx))) return r;
    if ((r = input_to_device(uc, ctx, args[0])) || (r = input_to_device(uc, ctx, args[1]))) return r;
    if ((r = output_on_device(uc, ctx, args[2]))) return r;
    if (ow == 0 || oh == 0 || nc == 0) {
        mark_output_written(final_);
        return 0;
    }
    LBGeom g;
    auto box_of = [](const halide_buffer_t *b, int &x0, int &x1, int &y0, int &y1) {
        x0 = b->dim[0].min, x1 = x0 + b->dim[0].extent - 1, y0 = b->dim[1].min, y1 = y0 + b->dim[1].extent - 1;
    };
    box_of(left_im, g.lx0, g.lx1, g.ly0, g.ly1);
    box_of(right_im, g.rx0, g.rx1, g.ry0, g.ry1);
    g.l_sy = left_im->dim[1].stride, g.l_sc = left_im->dim[2].stride, g.r_sy = right_im->dim[1].stride, g.r_sc = right_im->dim[2].stride;
    for (int c = 0; c < 3; c++) {
        auto cl = [](const halide_buffer_t *b, int c) { const int lo = b->dim[2].min, hi = lo + b->dim[2].extent - 1; return (c < lo ? lo : (c > hi ? hi : c)) - lo; };
        g.l_c[c] = (long)cl(left_im, c) * g.l_sc, g.r_c[c] = (long)cl(right_im, c) * g.r_sc;
    }
    g.slices = slices, g.focus = focus_depth, g.samples = aperture_samples, g.scale = blur_radius_scale, g.fslices = (float)slices;
    g.R = (int)((float)(slices - focus_depth > focus_depth ? slices - focus_depth : focus_depth) * blur_radius_scale);
    g.tag = g_rand_tag.load();
    const int ox0 = final_->dim[0].min, oy0 = final_->dim[1].min;

    // ---- boxes (oracle/lens_blur_oracle.c); the pyramids' clamp extents come from left_im's extents (:57-61)
    Box D = {ox0 - g.R, oy0 - g.R, ow + 2 * g.R, oh + 2 * g.R};
    if (D.w >= (1 << 23)) {   // lb_final addresses a sample's row with a 24-bit multiply
        return report(uc, halide_error_code_buffer_extents_too_large, "Output buffer final is %d wide: at most %d columns are supported", ow, (1 << 23) - 1 - 2 * g.R);
    }
    Box P[LV], PB[LV];
    P[0] = D;
    for (int i = 1; i < LV; i++) {
        const int x0 = floor_div(P[i - 1].x0, 2) - 1, x1 = floor_div(P[i - 1].x0 + P[i - 1].w - 1, 2) + 1;
        const int y0 = floor_div(P[i - 1].y0, 2) - 1, y1 = floor_div(P[i - 1].y0 + P[i - 1].h - 1, 2) + 1;
        P[i] = {x0, y0, x1 - x0 + 1, y1 - y0 + 1};
    }
    int w = left_im->dim[0].extent, h = left_im->dim[1].extent;
    for (int i = 1; i < LV; i++) {
        w /= 2, h /= 2;
        PB[i] = {0, 0, w > 1 ? w : 1, h > 1 ? h : 1};
    }
    {
        int x0 = min(P[0].x0, -1), y0 = min(P[0].y0, -1);
        int x1 = max(P[0].x0 + P[0].w - 1, 2 * PB[1].w), y1 = max(P[0].y0 + P[0].h - 1, 2 * PB[1].h);
        PB[0] = {x0, y0, x1 - x0 + 1, y1 - y0 + 1};
    }
    const Box E = PB[0];
    // ---- workspace
    auto al = [](size_t n) { return (n + 63) & ~(size_t)63; };
    size_t off_push[LV], off_pull[LV], total = 0;
    // fused front end (lb_cost_down, lb_depth_rc): push[0] is never stored.  HLMI_LB_UNFUSED=1: one stage per launch (A/B)
    bool fused = (size_t)(slices + 1) * PB[1].w * PB[1].h < ((size_t)1 << 31);
    {
        const char *e = getenv("HLMI_LB_UNFUSED");
        if (e && *e && atoi(e) != 0) fused = false;
    }
    for (int i = 0; i < LV; i++) off_push[i] = total, total += (i == 0 && fused) ? 0 : al((size_t)(slices + 1) * PB[i].w * PB[i].h);
    for (int i = 1; i < LV; i++) off_pull[i] = total, total += al((size_t)(slices + 1) * P[i].w * P[i].h);
    const size_t off_depth = total;
    total += al((size_t)D.w * D.h);
    const size_t off_br = total;
    total += al((size_t)D.w * D.h);
    const size_t off_wcy = total;
    total += al((size_t)D.w * oh);
    void *ws = nullptr;
    if ((r = get_workspace(uc, ctx, total * sizeof(float), &ws))) return r;
    float *wsf = (float *)ws;
    float *push[LV], *pull[LV];
    for (int i = 0; i < LV; i++) push[i] = wsf + off_push[i];
    for (int i = 1; i < LV; i++) pull[i] = wsf + off_pull[i];
    pull[0] = nullptr;
    uint32_t *depth = (uint32_t *)(wsf + off_depth);
    float *br = wsf + off_br, *wcy = wsf + off_wcy;

    const uint8_t *dl = dev_ptr<uint8_t>(left_im), *dr = dev_ptr<uint8_t>(right_im);
    hipStream_t st = ctx.stream;
    const unsigned zc = (unsigned)slices + 1u;   // planes: cost x confidence per slice + the confidence
    const bool s64 = slices > 32, full = slices == 32 || slices == 64;
#define LB_DISPATCH(K, ...)                                                             \
    do {                                                                                \
        if (!s64 && full) HLMI_LAUNCH(uc, #K, st, (K<32, true>), __VA_ARGS__);          \
        else if (!s64) HLMI_LAUNCH(uc, #K, st, (K<32, false>), __VA_ARGS__);            \
        else if (full) HLMI_LAUNCH(uc, #K, st, (K<64, true>), __VA_ARGS__);             \
        else HLMI_LAUNCH(uc, #K, st, (K<64, false>), __VA_ARGS__);                      \
    } while (0)
    if (fused) {
        const int strips = (PB[1].w + 126) / 127, ndx = (PB[1].w + strips - 1) / strips;
        // rows of push[1] per workgroup: a workgroup walks 2 ndy + 2 source rows (the 2 are halo), and the chip holds
        // `cap` workgroups at once (registers: two per CU at 32 slices, one at 64) — the fewest row-steps over all rounds
        int ndy = 8;
        {
            const long cap = (long)stream_cu_count(ctx.device, ctx.stream) * (s64 ? 1 : 2);
            long best_steps = -1;
            for (int c = 2; c <= 32; c++) {
                const long wgs = (long)strips * ((PB[1].h + c - 1) / c), steps = ((wgs + cap - 1) / cap) * (2 * c + 2);
                if (best_steps < 0 || steps < best_steps) best_steps = steps, ndy = c;
            }
            const char *e = getenv("HLMI_LB_NDY");
            if (e && *e) ndy = max(1, min(1024, atoi(e)));
        }
        const size_t lds = (size_t)zc * CD_RP * sizeof(float);
        const dim3 grid(strips, (PB[1].h + ndy - 1) / ndy);
        if (s64) {
            HLMI_HIP(uc, hipFuncSetAttribute((const void *)lb_cost_down<64, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            HLMI_HIP(uc, hipFuncSetAttribute((const void *)lb_cost_down<64, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        }
        const char *two = getenv("HLMI_LB_ROWS2");   // two source rows per step, 512 threads (default up to 32 slices; 0 / 1: A/B)
        if (two && *two ? *two != '0' : !s64) {
            const size_t lds2 = 2 * lds;
            if (s64) {
                HLMI_HIP(uc, hipFuncSetAttribute((const void *)lb_cost_down2<64, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
                HLMI_HIP(uc, hipFuncSetAttribute((const void *)lb_cost_down2<64, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
            } else {
                HLMI_HIP(uc, hipFuncSetAttribute((const void *)lb_cost_down2<32, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
                HLMI_HIP(uc, hipFuncSetAttribute((const void *)lb_cost_down2<32, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
            }
            LB_DISPATCH(lb_cost_down2, grid, dim3(512), lds2, dl, dr, g, push[1], PB[1], ndx, ndy);
        } else {
            LB_DISPATCH(lb_cost_down, grid, dim3(256), lds, dl, dr, g, push[1], PB[1], ndx, ndy);
        }
    } else {
        LB_DISPATCH(lb_cost, dim3((E.w + 255) / 256, E.h), dim3(256), 0, dl, dr, g, E, push[0]);
    }
    // the *32 kernels (32-bit plane offsets from 24-bit multiplies) take boxes that are small enough for them; HLMI_LB_NO_A32=1: never (A/B)
    const bool a32 = !env_flag("HLMI_LB_NO_A32");
    auto box_small = [&](const Box &b) { return a32 && b.w < (1 << 23) && b.h < (1 << 23) && (long)b.w * b.h < (1L << 29); };
    // levels below `tail` (at most 128 x 128 elements per plane) go down and up in ONE launch, a workgroup per plane
    int tail = LV;
    for (int i = LV - 1; i >= 2; i--) {
        if ((long)PB[i - 1].w * PB[i - 1].h <= 128 * 128 * 4 && (long)P[i].w * P[i].h <= 128 * 128) tail = i;
        else break;
    }
    for (int i = fused ? 2 : 1; i < tail; i++) {
        char nm[24];
        snprintf(nm, sizeof nm, "lb_down:%d", i);
        if (i == 1) HLMI_LAUNCH(uc, nm, st, lb_down<false>, dim3((PB[i].w + DTW - 1) / DTW, (PB[i].h + DTH - 1) / DTH, zc), dim3(256), 0, push[0], PB[0], push[i], PB[i]);
        else if (box_small(PB[i - 1]) && box_small(PB[i]))
            HLMI_LAUNCH(uc, nm, st, lb_down32<true>, dim3((PB[i].w + DTW - 1) / DTW, (PB[i].h + DTH - 1) / DTH, zc), dim3(256), 0, push[i - 1], PB[i - 1], push[i], PB[i]);
        else HLMI_LAUNCH(uc, nm, st, lb_down<true>, dim3((PB[i].w + DTW - 1) / DTW, (PB[i].h + DTH - 1) / DTH, zc), dim3(256), 0, push[i - 1], PB[i - 1], push[i], PB[i]);
    }
    if (tail < LV) {
        TailArgs ta;
        for (int i = 0; i < LV; i++) ta.push[i] = push[i], ta.pull[i] = pull[i], ta.PB[i] = PB[i], ta.P[i] = P[i];
        ta.from = tail;
        char nm[24];
        snprintf(nm, sizeof nm, "lb_tail:%d", tail);
        long need = 0;
        for (int i = tail; i < LV; i++) need += (long)PB[i].w * PB[i].h + (long)P[i].w * P[i].h;
        if (need <= TAIL_LDS) HLMI_LAUNCH(uc, nm, st, lb_tail_lds, dim3(zc), dim3(1024), 0, ta);
        else HLMI_LAUNCH(uc, nm, st, lb_tail, dim3(zc), dim3(1024), 0, ta);
    }
    // levels 3, 2, 1 in one launch when all three lie above the tail (pull[3] and pull[2] then exist only in LDS)
    const bool pull_multi = tail >= 4;
    if (pull_multi) {
        for (int i = min(LV - 1, tail - 1); i >= 4; i--) {
            char nm[24];
            snprintf(nm, sizeof nm, "lb_pull:%d", i);
            if (i == LV - 1) HLMI_LAUNCH(uc, nm, st, lb_pull<true>, dim3((P[i].w + 63) / 64, (P[i].h + 3) / 4, zc), dim3(256), 0, push[i], PB[i], (const float *)nullptr, P[i], pull[i], P[i]);
            else HLMI_LAUNCH(uc, nm, st, lb_pull<false>, dim3((P[i].w + 63) / 64, (P[i].h + 3) / 4, zc), dim3(256), 0, push[i], PB[i], pull[i + 1], P[i + 1], pull[i], P[i]);
        }
        // 32-bit plane offsets from 24-bit multiplies (lb_pull_multi32) when every plane involved is small enough for them
        bool small = true;
        for (const Box *b : {&PB[1], &PB[2], &PB[3], &P[4], &P[1]}) small = small && box_small(*b);
        if (small)
            HLMI_LAUNCH(uc, "lb_pull_multi:1", st, lb_pull_multi32, dim3((P[1].w + PMW - 1) / PMW, (P[1].h + PMH - 1) / PMH, zc), dim3(256), 0, push[1],
                        PB[1], push[2], PB[2], push[3], PB[3], pull[4], P[4], pull[1], P[1]);
        else
            HLMI_LAUNCH(uc, "lb_pull_multi:1", st, lb_pull_multi, dim3((P[1].w + PMW - 1) / PMW, (P[1].h + PMH - 1) / PMH, zc), dim3(256), 0, push[1], PB[1],
                        push[2], PB[2], push[3], PB[3], pull[4], P[4], pull[1], P[1]);
    }
    for (int i = pull_multi ? 0 : min(LV - 1, tail - 1); i >= 1; i--) {
        char nm[24];
        snprintf(nm, sizeof nm, "lb_pull:%d", i);
        if (i == LV - 1) HLMI_LAUNCH(uc, nm, st, lb_pull<true>, dim3((P[i].w + 63) / 64, (P[i].h + 3) / 4, zc), dim3(256), 0, push[i], PB[i], (const float *)nullptr, P[i], pull[i], P[i]);
        else HLMI_LAUNCH(uc, nm, st, lb_pull<false>, dim3((P[i].w + 63) / 64, (P[i].h + 3) / 4, zc), dim3(256), 0, push[i], PB[i], pull[i + 1], P[i + 1], pull[i], P[i]);
    }
    if (!fused) HLMI_LAUNCH(uc, "lb_depth", st, lb_depth, dim3((D.w + 255) / 256, D.h), dim3(256), 0, push[0], E, pull[1], P[1], dl, g, D, depth, br);
    else LB_DISPATCH(lb_depth_rc, dim3((D.w + 63) / 64, (D.h + 3) / 4), dim3(256), 0, dl, dr, pull[1], P[1], g, D, depth, br);
#undef LB_DISPATCH
    if (!fused || env_flag("HLMI_LB_WCY_LAUNCH")) {
        HLMI_LAUNCH(uc, "lb_wcy", st, lb_wcy, dim3((D.w + 255) / 256, oh), dim3(256), 0, br, D, g.R, oy0, oh, wcy);
        HLMI_LAUNCH(uc, "lb_final", st, lb_final<false>, dim3((ow + 255) / 256, oh), dim3(256), 0, g, depth, wcy, D, ox0, oy0, ow, nc,
                    dev_ptr<float>(final_), (long)final_->dim[1].stride, (long)final_->dim[2].stride);
    } else {
        HLMI_LAUNCH(uc, "lb_final", st, lb_final<true>, dim3((ow + 255) / 256, oh), dim3(256), 0, g, depth, br, D, ox0, oy0, ow, nc,
                    dev_ptr<float>(final_), (long)final_->dim[1].stride, (long)final_->dim[2].stride);
    }
    mark_output_written(final_);
    return 0;
}

extern "C" int lens_blur_argv(void **a) {
    return lens_blur((halide_buffer_t *)a[0], (halide_buffer_t *)a[1], *(int32_t *)a[2], *(int32_t *)a[3], *(float *)a[4], *(int32_t *)a[5],
                     (halide_buffer_t *)a[6]);
}
extern "C" const halide_filter_metadata_t *lens_blur_metadata(void) { return &lb_md; }
extern "C" int lens_blur_auto_schedule(halide_buffer_t *left_im, halide_buffer_t *right_im, int32_t slices, int32_t focus_depth,
                                       float blur_radius_scale, int32_t aperture_samples, halide_buffer_t *final_) {
    return lens_blur(left_im, right_im, slices, focus_depth, blur_radius_scale, aperture_samples, final_);
}
