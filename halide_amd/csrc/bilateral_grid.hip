// bilateral_grid.hip — gfx950 implementation of the reference's bilateral_grid AOT pipeline.
//
// Algorithm: /root/reference/apps/bilateral_grid/bilateral_grid_generator.cpp:14-67 (s_sigma = 8, :8);
// boundary: `int bilateral_grid(halide_buffer_t *input, float r_sigma, halide_buffer_t *bilateral_grid)`
// (:10-12, :203).  HBM-bound at 8 B/px (4 read + 4 written); the grid (~3 MB at 1080p) lives in L2/MALL.
//
// Kernels (the decomposition of the reference's GPU schedule, :85-119; for grids of at most 16 planes the last three stages
// run as ONE launch, bg_blur_slice, with the blurred cells of a pixel tile in LDS):
//   bg_histogram_blurz  one thread per grid cell: zero an LDS histogram, accumulate its 8x8 pixels SERIALLY in
//                       RDom order (float sums are order-sensitive, :28-29), blur in z, write float2 {value, weight}
//   bg_blurx, bg_blury  5-tap [1 4 6 4 1], left-to-right association (:38-47)
//   bg_slice            trilinear lerp x->y->z of both channels, divide (:50-67)
// Grid layout in HBM: float2 {value, weight} per cell, x,y covering the cells the output region touches (+2 halo for the
// pre-blur stages), z in [0, zmax+1] with zmax = int(1/r_sigma + 0.5).  Grids of at most 16 planes (the two-launch path) keep the
// blurz grid z-INNERMOST, G[y][x][z]: the (cell, bin) threads of the histogram kernel store one contiguous run per workgroup and
// a tile row of bg_blur_slice is one contiguous run too; larger grids (the serial histogram + three launches) keep G[z][y][x].
#include "hlmi_device_math.h"
#include "hlmi_internal.h"

#include <stdlib.h>

using namespace hlmi;

namespace {

constexpr int S = 8;  // s_sigma

struct BGeom {
    int ix0, ix1, iy0, iy1;  // clamp box of the input (absolute)
    int gx0, gy0;            // absolute cell coordinate of blury's x=0 / y=0
    int GX, GY, HX, HY;      // blury extents; histogram/blurz extents (GX+4, GY+4)
    int ZH, ZD;              // histogram bins, blurred planes
    float inv_r;
};

__device__ __forceinline__ float blur5(float a, float b, float c, float d, float e) {
    return dev::mad(d, 4.0f, dev::mad(c, 6.0f, dev::mad(b, 4.0f, a))) + e;   // (((a + b 4) + c 6) + d 4) + e; the products have one use each
}

__global__ void bg_histogram_blurz(const float *__restrict__ in, long in_sy, BGeom g, float2 *__restrict__ bz) {
    extern __shared__ float hist[];  // [ZH][2][T]
    const int T = blockDim.x, t = threadIdx.x;
    const int cx = blockIdx.x * T + t, cy = blockIdx.y;
    for (int i = t; i < g.ZH * 2 * T; i += T) hist[i] = 0.0f;
    __syncthreads();
    const bool active = cx < g.HX;
    if (active) {
        const int gx = g.gx0 - 2 + cx, gy = g.gy0 - 2 + cy;
        // all 64 pixel loads of the cell are issued before the (inherently serial, order-defining) accumulation: the
        // dependent chain is then 64 LDS read-modify-writes, not 64 memory latencies
        float v[S][S];
#pragma unroll
        for (int ry = 0; ry < S; ry++) {
            const int py = dev::clampi(gy * S + ry - S / 2, g.iy0, g.iy1) - g.iy0;
            const float *row = in + (long)py * in_sy;
#pragma unroll
            for (int rx = 0; rx < S; rx++) {
                const int px = dev::clampi(gx * S + rx - S / 2, g.ix0, g.ix1) - g.ix0;
                v[ry][rx] = row[px];
            }
        }
#pragma unroll
        for (int ry = 0; ry < S; ry++) {
#pragma unroll
            for (int rx = 0; rx < S; rx++) {
                float val = dev::clampf(v[ry][rx], 0.0f, 1.0f);
                int zi = (int)dev::mad(val, g.inv_r, 0.5f);
                float *h = &hist[(zi * 2) * T + t];
                h[0] = h[0] + val;
                h[T] = h[T] + 1.0f;
            }
        }
        const size_t plane = (size_t)g.HX * g.HY;
        for (int z = 0; z < g.ZD; z++) {
            float v[2];
#pragma unroll
            for (int c = 0; c < 2; c++) {
                auto H = [&](int zz) -> float { return (zz >= 0 && zz < g.ZH) ? hist[(zz * 2 + c) * T + t] : 0.0f; };
                v[c] = blur5(H(z - 2), H(z - 1), H(z), H(z + 1), H(z + 2));
            }
            bz[(size_t)z * plane + (size_t)cy * g.HX + cx] = make_float2(v[0], v[1]);
        }
    }
}

// Same function for grids of at most 16 planes (r_sigma >= 1/14.5; the reference's 0.1 gives 12), parallel over
// (cell, range bin): thread (c, z) walks the 64 pixels of cell c in RDom order and adds those that fall into bin z —
// per bin exactly the additions, in exactly the order, of the serial histogram (:28-29), but a register chain of 64
// steps instead of 64 LDS read-modify-writes, and HZ x the threads.
//   * HZ = bin slots per cell: 12 when the grid has at most 12 planes (r_sigma = 0.1: 11 bins, 12 planes; 21 cells per
//     workgroup of 252 threads), else 16 — idle bin lanes are pure loss in a kernel bound by VALU issue.
//   * the weight channel is a COUNT (sums of 1.0f up to 64 are exact in any order): an integer add-with-carry per pixel
//     instead of a select and a float add.
//   * LDS rows are padded to 72 words: the 16 (or 12) threads of a cell read one address (broadcast), but unpadded rows put
//     every cell's row on the same banks (4-way conflicts on both 16-byte reads of each step; 1.1 M conflict cycles per launch
//     in profiles/r02f_apps_pmc.txt — with four SIMDs sharing the LDS that, not the VALU, set the pace).  72, not 68: the
//     staging's 16-byte stores are served eight lanes at a time on 32 banks, and a lane pair of the next cell landed on the
//     banks of this cell's second half (scripts/model/lds_banks.py: 182 -> 86 cycles per workgroup).
constexpr int HTH = 256;         // at most this many threads per workgroup
constexpr int HROW = S * S + 8;  // padded row of the staged cell
template<int HZ>
__global__ __launch_bounds__(HTH) void bg_histogram_blurz_par(const float *__restrict__ in, long in_sy, BGeom g,
                                                              float2 *__restrict__ bz, int vec) {
    constexpr int HC = HTH / HZ, NT = HC * HZ;   // cells per workgroup (one grid row segment), threads used
    __shared__ __attribute__((aligned(16))) float s_val[HC][HROW];
    __shared__ __attribute__((aligned(16))) int s_zi[HC][HROW];
    __shared__ float2 s_h[HC][HZ + 4];          // histogram, bins -2 .. HZ+1 (zero padded for the z blur)
    const int t = threadIdx.x, c = t / HZ, z = t - c * HZ;
    const int cx0 = blockIdx.x * HC, cy = blockIdx.y;
    const int gy = g.gy0 - 2 + cy;
    // staging.  Interior workgroups (every staged column inside the image, rows 16-byte aligned: `vec`) move float4: a wave
    // reads 1 KB of one image row per instruction and writes the two LDS arrays 16 bytes at a time — 6 instead of 35 VALU
    // instructions per pixel; workgroups that touch the left / right image edge clamp every column.
    const int xlo = (g.gx0 - 2 + cx0) * S - S / 2;   // absolute x of the first staged column
    constexpr int RQ = HC * S / 4;                   // float4 per staged row
    if (vec && xlo >= g.ix0 && xlo + HC * S - 1 <= g.ix1) {
        for (int i = t; i < S * RQ; i += NT) {
            const int ry = i / RQ, q = i - ry * RQ;
            const int py = dev::clampi(gy * S + ry - S / 2, g.iy0, g.iy1) - g.iy0;
            const float4 v = *reinterpret_cast<const float4 *>(in + (long)py * in_sy + (xlo - g.ix0) + 4 * q);
            const int cc = q >> 1, o = ry * S + (q & 1) * 4;
            float4 cv;
            cv.x = dev::clampf(v.x, 0.0f, 1.0f), cv.y = dev::clampf(v.y, 0.0f, 1.0f);
            cv.z = dev::clampf(v.z, 0.0f, 1.0f), cv.w = dev::clampf(v.w, 0.0f, 1.0f);
            int4 zi;
            zi.x = (int)dev::mad(cv.x, g.inv_r, 0.5f), zi.y = (int)dev::mad(cv.y, g.inv_r, 0.5f);
            zi.z = (int)dev::mad(cv.z, g.inv_r, 0.5f), zi.w = (int)dev::mad(cv.w, g.inv_r, 0.5f);
            *reinterpret_cast<float4 *>(&s_val[cc][o]) = cv;
            *reinterpret_cast<int4 *>(&s_zi[cc][o]) = zi;
        }
    } else {
        for (int i = t; i < HC * S * S; i += NT) {
            const int ry = i / (HC * S), xx = i - ry * (HC * S), cc = xx / S, rx = xx % S;   // lanes run along the image row
            const int py = dev::clampi(gy * S + ry - S / 2, g.iy0, g.iy1) - g.iy0;
            const int px = dev::clampi(xlo + xx, g.ix0, g.ix1) - g.ix0;
            const float val = dev::clampf(in[(long)py * in_sy + px], 0.0f, 1.0f);
            s_val[cc][ry * S + rx] = val;
            s_zi[cc][ry * S + rx] = (int)dev::mad(val, g.inv_r, 0.5f);
        }
    }
    for (int i = t; i < HC * (HZ + 4); i += NT) s_h[i / (HZ + 4)][i % (HZ + 4)] = make_float2(0.0f, 0.0f);
    __syncthreads();
    // four pixels per LDS instruction; a sum that starts at +0 and only ever adds non-negative terms is never -0, so adding
    // +0 for the pixels of other bins leaves it bit for bit what the selective add would give
    float hv = 0.0f;
    int cnt = 0;
    const float4 *v4 = reinterpret_cast<const float4 *>(s_val[c]);
    const int4 *z4 = reinterpret_cast<const int4 *>(s_zi[c]);
#pragma unroll 4
    for (int q = 0; q < S * S / 4; q++) {
        const float4 v = v4[q];
        const int4 zz = z4[q];
        hv = hv + (zz.x == z ? v.x : 0.0f), cnt += (zz.x == z);
        hv = hv + (zz.y == z ? v.y : 0.0f), cnt += (zz.y == z);
        hv = hv + (zz.z == z ? v.z : 0.0f), cnt += (zz.z == z);
        hv = hv + (zz.w == z ? v.w : 0.0f), cnt += (zz.w == z);
    }
    if (z < g.ZH) s_h[c][z + 2] = make_float2(hv, (float)cnt);
    __syncthreads();
    const int cx = cx0 + c;
    if (z < g.ZD && cx < g.HX) {
        const float2 a = s_h[c][z], b = s_h[c][z + 1], m = s_h[c][z + 2], d = s_h[c][z + 3], e = s_h[c][z + 4];
        // z innermost: the workgroup's (cell, bin) threads store one contiguous run
        bz[((size_t)cy * g.HX + cx) * g.ZD + z] = make_float2(blur5(a.x, b.x, m.x, d.x, e.x), blur5(a.y, b.y, m.y, d.y, e.y));
    }
}

// blurx: out[z][y][x] (GX wide) from bz (HX wide), y over HY
__global__ void bg_blurx(const float2 *__restrict__ bz, BGeom g, float2 *__restrict__ bx) {
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, z = blockIdx.z;
    if (x >= g.GX) return;
    const float2 *s = bz + ((size_t)z * g.HY + y) * g.HX + x;
    float2 a = s[0], b = s[1], c = s[2], d = s[3], e = s[4];
    bx[((size_t)z * g.HY + y) * g.GX + x] = make_float2(blur5(a.x, b.x, c.x, d.x, e.x), blur5(a.y, b.y, c.y, d.y, e.y));
}

__global__ void bg_blury(const float2 *__restrict__ bx, BGeom g, float2 *__restrict__ by) {
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, z = blockIdx.z;
    if (x >= g.GX) return;
    const float2 *s = bx + ((size_t)z * g.HY + y) * g.GX + x;
    float2 a = s[0], b = s[g.GX], c = s[2 * g.GX], d = s[3 * g.GX], e = s[4 * g.GX];
    by[((size_t)z * g.GY + y) * g.GX + x] = make_float2(blur5(a.x, b.x, c.x, d.x, e.x), blur5(a.y, b.y, c.y, d.y, e.y));
}

__device__ __forceinline__ float2 lerp2(float2 a, float2 b, float w) {
    return make_float2(dev::lerpf(a.x, b.x, w), dev::lerpf(a.y, b.y, w));
}

__global__ __launch_bounds__(256) void bg_slice(const float *__restrict__ in, long in_sy, BGeom g, const float2 *__restrict__ by,
                                               float *__restrict__ out, long out_sy, int ox0, int oy0, int ow) {
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= ow) return;
    const int ax = ox0 + x, ay = oy0 + y;
    float val = dev::clampf(in[(long)(ay - g.iy0) * in_sy + (ax - g.ix0)], 0.0f, 1.0f);
    float zv = val * g.inv_r;
    int zi = (int)zv;
    float zf = zv - (float)zi;
    float xf = (float)dev::fmod8(ax) * 0.125f, yf = (float)dev::fmod8(ay) * 0.125f;
    int xi = dev::fdiv8(ax) - g.gx0, yi = dev::fdiv8(ay) - g.gy0;
    const size_t plane = (size_t)g.GX * g.GY;
    const float2 *p0 = by + (size_t)zi * plane + (size_t)yi * g.GX + xi;
    const float2 *p1 = p0 + plane;
    float2 a = lerp2(lerp2(p0[0], p0[1], xf), lerp2(p0[g.GX], p0[g.GX + 1], xf), yf);
    float2 b = lerp2(lerp2(p1[0], p1[1], xf), lerp2(p1[g.GX], p1[g.GX + 1], xf), yf);
    float2 r = lerp2(a, b, zf);
    out[(long)y * out_sy + x] = r.x / r.y;
}

// ---- bg_blur_slice: blurx + blury + slice in ONE launch (grids of at most 16 planes).  A workgroup owns 64 x 32 output
// pixels; the blury cells they interpolate between (at most 10 x 6 x ZD), the blurx rows under those (10 x 10) and the
// blurz cells under those (14 x 10) are produced in LDS by the same blur5 chains the separate kernels run — two launches
// and the write + re-read of two grids less (the pipeline is launch-latency bound: 4 launches took 30 us for 16 MB).
// Round 5, LDS layout: all three tiles are z-INNERMOST, [row][cell][ZP] float2 with ZP = 12 or 16 planes of pitch.
//   * the slicing stage's eight taps per pixel are data-dependent in z: ds_read_b64 serves a wave in two 32-lane halves on 64
//     banks, so a half must not put two distinct (z, cell) pairs on one bank pair.  With [z][row][cell] tiles a half covered four
//     cells x up to 12 planes = 48 pairs on 32 bank pairs (planes z and z + 8 collided: 2 x the cycles on a noise image,
//     1.4 conflict cycles per LDS instruction in profiles/r04f_apps_pmc.txt).  Now a half covers 16 pixels x 2 rows = TWO
//     cells, and (cell, z) -> dword 2 ZP cell + 2 z: two windows of at most 32 dwords, no two pairs on one bank.
//   * the two blur stages walk their elements z-fastest: every LDS access of a wave is one contiguous run.
//   * staging copies contiguous runs of the z-innermost blurz grid (a tile row = 14 cells x ZD planes) to contiguous LDS.
// (scripts/model/lds_banks.py models the three stages: 1684 -> 946 LDS cycles per workgroup on a noise image.)
constexpr int FPX = 64, FCX = 10, FZ = 16;
// e / D for 0 <= e <= MAXE as one full-rate 24-bit multiply and a shift (v_mul_hi_u32, what `/` by a constant compiles to, issues
// at a quarter of the rate): exact while e (M D - 2^16) < 2^16 with M = ceil(2^16 / D)
template<int D, int MAXE>
__device__ __forceinline__ uint32_t divc(uint32_t e) {
    if ((D & (D - 1)) == 0) return e / (uint32_t)D;
    constexpr uint32_t M = (65536u + D - 1) / D;
    static_assert((unsigned long long)(M * D - 65536u) * MAXE < 65536ull && (unsigned long long)M * MAXE < (1ull << 32), "divc: range");
    return __umul24(e, M) >> 16;
}
// a x b of two values below 2^24 as v_mul_u32_u24, kept apart from a following add (the compiler folds `mul24 + c` into
// v_mad_u64_u32: a quarter-rate instruction again)
__device__ __forceinline__ uint32_t mul24o(uint32_t a, uint32_t b) {
    uint32_t r = __umul24(a, b);
    asm("" : "+v"(r));
    return r;
}
// A32: every byte offset into the input, the output and the blurz grid fits 31 bits and every row number / row stride 23 (the
// host checks): addresses are `uniform base + 32-bit byte offset` built from full-rate 24-bit multiplies.  As 64-bit element
// indices the address arithmetic of this kernel was 92 quarter-rate instructions (v_mul_lo_u32, v_mul_hi_u32, v_mad_u64_u32) of
// 958: 30 % of its issue time.
// HIST (round 6): the WHOLE pipeline in one launch.  The blurz cells of the tile are not read from a grid another launch made but
// built here: the histogram of the tile's 14 x 10 cells from the input (every cell's 64 pixels in RDom order, thread = (cell, bin) as
// in bg_histogram_blurz_par: per bin exactly the additions of :28-29 in their order) and its z blur, straight into s_bz.  A cell is
// rebuilt by every tile whose blur footprint holds it (4.4 x the histogram work of the two-launch path, its input re-read from L2)
// in exchange for one launch, no grid round trip and no dependent launch.  Measured: slower (0.0297 vs 0.0180 ms per call), so this
// mode is opt-in (HLMI_BG_ONE_LAUNCH=1).  The tile's 112 x 80 input pixels are requested first, all of them, before anything is waited for.
template<int FPY, int ZP, bool A32, bool HIST = false>   // tile height (32; 16 and 64 measured the same or slower); z pitch of the LDS tiles (>= g.ZD)
__global__ __launch_bounds__(256) void bg_blur_slice(const float *__restrict__ in, long in_sy, BGeom g, const float2 *__restrict__ bz,
                                                    float *__restrict__ out, long out_sy, int ox0, int oy0, int ow, int oh) {
    constexpr int FCY = FPY / S + 2;
    constexpr int NBZ = (FCY + 4) * (FCX + 4) * ZP, NBX = (FCY + 4) * FCX * ZP, NBY = FCY * FCX * ZP;
    // one array: the staging of the histogram passes (HIST) lives where s_bx / s_by will be
    __shared__ __attribute__((aligned(16))) float2 s_all[NBZ + NBX + NBY];
    float2 *const s_bz = s_all, *const s_bx = s_all + NBZ, *const s_by = s_all + NBZ + NBX;
    const int t = threadIdx.x;
    const int x0 = blockIdx.x * FPX, y0 = blockIdx.y * FPY;
    const int cxa = dev::fdiv8(ox0 + x0) - g.gx0, cya = dev::fdiv8(oy0 + y0) - g.gy0;   // first blury cell of the tile
    // lane -> pixel: a wave covers 32 columns x 2 rows per step (each 32-lane half: 16 columns x 2 rows), the four waves are the
    // two column halves of the tile x two row pairs, a step advances by four rows
    const int wave = t >> 6, lane = t & 63;
    const int pcol = (wave & 1) * 32 + ((lane >> 5) << 4) + (lane & 15);
    const int prow = ((wave >> 1) << 1) + ((lane >> 4) & 1);
    // the thread's own pixels first: their latency hides behind the three grid phases (a load cannot move above a barrier
    // by itself); rows / columns past the output re-read its last row / column and are never stored
    float pix[FPY / 4];
    {
        const int axc = ox0 + min(x0 + pcol, ow - 1);
#pragma unroll
        for (int k = 0; k < FPY / 4; k++) {
            const int ayc = oy0 + min(y0 + prow + 4 * k, oh - 1);
            if (A32) {
                const uint32_t ob = (mul24o((uint32_t)(ayc - g.iy0), (uint32_t)in_sy) + (uint32_t)(axc - g.ix0)) << 2;
                pix[k] = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(in) + ob);
            } else {
                pix[k] = in[(long)(ayc - g.iy0) * in_sy + (axc - g.ix0)];
            }
        }
    }
    if (HIST) {
        constexpr int CW = FCX + 4, CH = FCY + 4, NCELL = CW * CH;        // the tile's blurz cells: 14 x 10
        constexpr int RW = CW * S, RH = CH * S, NPX = RW * RH;            // their pixels: 112 x 80
        constexpr int NLD = (NPX + 255) / 256;                            // loads per thread (35)
        // idx / RW for idx < NLD * 256 as one 24-bit multiply and a shift: exact while idx (M RW - 2^21) < 2^21
        constexpr uint32_t RWM = ((1u << 21) + RW - 1) / RW;
        static_assert((unsigned long long)(RWM * RW - (1u << 21)) * (NLD * 256) < (1ull << 21) && (unsigned long long)RWM * (NLD * 256) < (1ull << 32), "row division");
        constexpr int HC = 256 / ZP, NTH = HC * ZP;                       // cells per histogram pass, threads that accumulate
        constexpr int NPASS = (NCELL + HC - 1) / HC;
        constexpr int SP = 2 * S * S + 8;                                 // staged cell: 64 {value, bin} pairs, pitch padded by 8 words
        static_assert((size_t)HC * SP * 4 <= (size_t)(NBX + NBY) * 8, "bg_blur_slice<HIST>: staging does not fit under s_bx / s_by");
        __shared__ float2 s_h[HC][ZP + 4];                                // a pass's histograms, bins -2 .. ZP + 1 (zero padded)
        float *const s_st = reinterpret_cast<float *>(s_bx);
        // the region's pixels, row-major over 112 x 80, edge-clamped (:21-23); element idx = t + 256 k
        const int X0 = (g.gx0 - 2 + cxa) * S - S / 2, Y0 = (g.gy0 - 2 + cya) * S - S / 2;
        float v[NLD];
#pragma unroll
        for (int k = 0; k < NLD; k++) {
            const uint32_t idx = min((uint32_t)(t + 256 * k), (uint32_t)(NPX - 1));
            const uint32_t row = __umul24(idx, RWM) >> 21, col = idx - row * RW;
            const int px = dev::clampi(X0 + (int)col, g.ix0, g.ix1) - g.ix0, py = dev::clampi(Y0 + (int)row, g.iy0, g.iy1) - g.iy0;
            if (A32) v[k] = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(in) + ((mul24o((uint32_t)py, (uint32_t)in_sy) + (uint32_t)px) << 2));
            else v[k] = in[(long)py * in_sy + px];
        }
        for (int i = t; i < HC * (ZP + 4); i += 256) (&s_h[0][0])[i] = make_float2(0.0f, 0.0f);
        const int hc = t / ZP, hz = t - hc * ZP;                          // this thread's (cell of the pass, bin)
#pragma unroll
        for (int p = 0; p < NPASS; p++) {
            // pass p = cells [p HC, p HC + HC) in row-major order; their pixels are rows [8 jlo, 8 jhi + 7] of the region, i.e. only the
            // thread's loads klo .. khi can belong to it (compile-time window, run-time test)
            const int e0 = p * HC, e1 = min(e0 + HC, NCELL) - 1;
            const int jlo = e0 / CW, jhi = e1 / CW;
            const int klo = max(0, (jlo * S * RW - 255) / 256), khi = min(NLD - 1, ((jhi * S + S) * RW - 1) / 256);
#pragma unroll
            for (int k = 0; k < NLD; k++) {
                if (k < klo || k > khi) continue;
                const uint32_t idx = (uint32_t)(t + 256 * k);
                const uint32_t row = __umul24(idx, RWM) >> 21, col = idx - row * RW;
                const int e = (int)(row >> 3) * CW + (int)(col >> 3);      // the pixel's cell
                if (idx < (uint32_t)NPX && e >= e0 && e <= e1) {
                    const float val = dev::clampf(v[k], 0.0f, 1.0f);
                    const int zi = (int)dev::mad(val, g.inv_r, 0.5f);       // (:25)
                    *reinterpret_cast<float2 *>(s_st + (e - e0) * SP + 2 * (int)(((row & 7) << 3) + (col & 7))) = make_float2(val, __int_as_float(zi));
                }
            }
            __syncthreads();
            if (t < NTH) {
                // a sum that starts at +0 and only ever adds non-negative terms is never -0: adding +0 for the pixels of other bins leaves it
                // bit for bit what the selective add gives; the weight channel is a count (sums of 1.0f up to 64 are exact in any order)
                float hv = 0.0f;
                int cnt = 0;
                const float4 *q4 = reinterpret_cast<const float4 *>(s_st + hc * SP);
#pragma unroll 8
                for (int q = 0; q < S * S / 2; q++) {
                    const float4 w = q4[q];                                 // {v0, bin0, v1, bin1}
                    const int z0 = __float_as_int(w.y), z1 = __float_as_int(w.w);
                    hv = hv + (z0 == hz ? w.x : 0.0f), cnt += (z0 == hz);
                    hv = hv + (z1 == hz ? w.z : 0.0f), cnt += (z1 == hz);
                }
                s_h[hc][hz + 2] = make_float2(hv, (float)cnt);
            }
            __syncthreads();
            if (t < NTH && e0 + hc <= e1) {
                const float2 a = s_h[hc][hz], b = s_h[hc][hz + 1], m = s_h[hc][hz + 2], d = s_h[hc][hz + 3], e = s_h[hc][hz + 4];
                // planes past ZD - 1 are never interpolated; what is computed there is a blur of zeros and in-range bins: harmless
                s_bz[(e0 + hc) * ZP + hz] = make_float2(blur5(a.x, b.x, m.x, d.x, e.x), blur5(a.y, b.y, m.y, d.y, e.y));
            }
        }
    } else {
        // all of the thread's cells are requested before the first one is waited for (a loop with a run-time trip count
        // pays the round trip per iteration: 7 x ~0.7 us was most of this kernel)
        constexpr int N1 = (NBZ + 255) / 256;
        float2 v[N1];
#pragma unroll
        for (int k = 0; k < N1; k++) {
            const int e = min(t + 256 * k, NBZ - 1);
            const int c = (int)divc<ZP, NBZ>((uint32_t)e), z = e - c * ZP, j = (int)divc<FCX + 4, NBZ / ZP>((uint32_t)c), i = c - j * (FCX + 4);
            const int hx = min(cxa + i, g.HX - 1), hy = min(cya + j, g.HY - 1);     // cells past the grid are never interpolated
            if (A32) {                                                              // planes past the grid likewise
                const uint32_t ob = (mul24o(mul24o((uint32_t)hy, (uint32_t)g.HX) + (uint32_t)hx, (uint32_t)g.ZD) + (uint32_t)min(z, g.ZD - 1)) << 3;
                v[k] = *reinterpret_cast<const float2 *>(reinterpret_cast<const char *>(bz) + ob);
            } else {
                v[k] = bz[((size_t)hy * g.HX + hx) * g.ZD + min(z, g.ZD - 1)];
            }
        }
#pragma unroll
        for (int k = 0; k < N1; k++) {
            const int e = t + 256 * k;
            if (e < NBZ) s_bz[e] = v[k];
        }
    }
    __syncthreads();
    for (int e = t; e < NBX; e += 256) {
        const int c = (int)divc<ZP, NBX + 256>((uint32_t)e), z = e - c * ZP, j = (int)divc<FCX, (NBX + 256) / ZP + 1>((uint32_t)c), i = c - j * FCX;
        const float2 *sp = s_bz + ((j * (FCX + 4) + i) * ZP + z);
        const float2 a = sp[0], b = sp[ZP], c2 = sp[2 * ZP], d = sp[3 * ZP], q = sp[4 * ZP];
        s_bx[e] = make_float2(blur5(a.x, b.x, c2.x, d.x, q.x), blur5(a.y, b.y, c2.y, d.y, q.y));
    }
    __syncthreads();
    for (int e = t; e < NBY; e += 256) {   // (row j + d, cell i, plane z) of s_bx is element e + d FCX ZP
        const float2 *sp = s_bx + e;
        const float2 a = sp[0], b = sp[FCX * ZP], c2 = sp[2 * FCX * ZP], d = sp[3 * FCX * ZP], q = sp[4 * FCX * ZP];
        s_by[e] = make_float2(blur5(a.x, b.x, c2.x, d.x, q.x), blur5(a.y, b.y, c2.y, d.y, q.y));
    }
    __syncthreads();
    const int x = x0 + pcol;
    if (x >= ow) return;
    const int ax = ox0 + x;
    const float xf = (float)dev::fmod8(ax) * 0.125f;
    const int xi = dev::fdiv8(ax) - g.gx0 - cxa;
#pragma unroll
    for (int k = 0; k < FPY / 4; k++) {
        const int y = y0 + prow + 4 * k;
        if (y >= oh) break;
        const int ay = oy0 + y;
        const float val = dev::clampf(pix[k], 0.0f, 1.0f);
        const float zv = val * g.inv_r;
        const int zi = (int)zv;
        const float zf = zv - (float)zi;
        const float yf = (float)dev::fmod8(ay) * 0.125f;
        const int yi = dev::fdiv8(ay) - g.gy0 - cya;
        // rows yi, yi + 1; cell xi + 1 is ZP further, plane zi + 1 one (all indices are small: 24-bit multiplies)
        const float2 *p0 = s_by + (mul24o(mul24o((uint32_t)yi, FCX) + (uint32_t)xi, ZP) + (uint32_t)zi), *p1 = p0 + FCX * ZP;
        const float2 a = lerp2(lerp2(p0[0], p0[ZP], xf), lerp2(p1[0], p1[ZP], xf), yf);
        const float2 b = lerp2(lerp2(p0[1], p0[ZP + 1], xf), lerp2(p1[1], p1[ZP + 1], xf), yf);
        const float2 r = lerp2(a, b, zf);
        if (A32) {
            const uint32_t ob = (mul24o((uint32_t)y, (uint32_t)out_sy) + (uint32_t)x) << 2;
            *reinterpret_cast<float *>(reinterpret_cast<char *>(out) + ob) = r.x / r.y;
        } else {
            out[(long)y * out_sy + x] = r.x / r.y;
        }
    }
}

const int64_t e0 = 0, ew = 1536, eh = 2560;
const int64_t *const est[4] = {&e0, &ew, &e0, &eh};
const halide_scalar_value_t est_rs = [] { halide_scalar_value_t v{}; v.u.f32 = 0.1f; return v; }();
const halide_type_t ty_f32 = {(decltype(halide_type_t::code))2, 32, 0};
// estimates: generator :73-81
const halide_filter_argument_t bg_args[3] = {
    {"input", halide_argument_kind_input_buffer, 2, ty_f32, nullptr, nullptr, nullptr, nullptr, est},
    {"r_sigma", halide_argument_kind_input_scalar, 0, ty_f32, nullptr, nullptr, nullptr, &est_rs, nullptr},
    {"bilateral_grid", halide_argument_kind_output_buffer, 2, ty_f32, nullptr, nullptr, nullptr, nullptr, est},
};
const halide_filter_metadata_t bg_md = {1, 3, bg_args, kTargetString, "bilateral_grid"};

}  // namespace

extern "C" int bilateral_grid(halide_buffer_t *input, float r_sigma, halide_buffer_t *output) {
    void *uc = nullptr;
    BufArg args[2] = {{"input", input, T_F32, 2, false}, {"bilateral_grid", output, T_F32, 2, true}};
    int r = check_not_null(uc, args, 2);
    if (r) return r;
    if ((r = check_type_and_dims(uc, args, 2))) return r;
    if (any_bounds_query(args, 2)) {
        // the unclamped `input(x, y)` of the slicing stage (:50) needs the output's region; histogram taps are clamped
        int mins[2] = {output->dim[0].min, output->dim[1].min}, ext[2] = {output->dim[0].extent, output->dim[1].extent};
        answer_query(input, mins, ext);
        answer_query(output, mins, ext);
        return 0;
    }
    if ((r = check_shape(uc, args[0])) || (r = check_shape(uc, args[1]))) return r;
    const int ow = output->dim[0].extent, oh = output->dim[1].extent;
    const int ox0 = output->dim[0].min, oy0 = output->dim[1].min;
    if ((r = check_covers(uc, args[0], 0, ox0, ow)) || (r = check_covers(uc, args[0], 1, oy0, oh))) return r;

    BGeom g;
    g.inv_r = 1.0f / r_sigma;
    const float zm = 1.0f * g.inv_r + 0.5f;
    if (!(zm >= 0.0f && zm < 8192.0f)) {
        return report(uc, halide_error_code_param_too_small,
                      "Parameter r_sigma is %g: the grid would need %g range bins (supported: r_sigma >= 1/8191)", r_sigma, zm);
    }
    const int zmax = (int)zm;
    g.ZH = zmax + 1, g.ZD = zmax + 2;

    DeviceCtx ctx;
    if ((r = acquire_device(uc, &ctx))) return r;
    if ((r = input_to_device(uc, ctx, args[0]))) return r;
    if ((r = output_on_device(uc, ctx, args[1]))) return r;
    if (ow == 0 || oh == 0) {
        mark_output_written(output);
        return 0;
    }
    g.ix0 = input->dim[0].min, g.ix1 = g.ix0 + input->dim[0].extent - 1;
    g.iy0 = input->dim[1].min, g.iy1 = g.iy0 + input->dim[1].extent - 1;
    g.gx0 = floor_div(ox0, S), g.gy0 = floor_div(oy0, S);
    g.GX = floor_div(ox0 + ow - 1, S) + 1 - g.gx0 + 1, g.GY = floor_div(oy0 + oh - 1, S) + 1 - g.gy0 + 1;
    g.HX = g.GX + 4, g.HY = g.GY + 4;

    auto al = [](size_t n) { return (n + 63) & ~(size_t)63; };
    const size_t n_bz = al((size_t)g.HX * g.HY * g.ZD), n_bx = al((size_t)g.GX * g.HY * g.ZD), n_by = al((size_t)g.GX * g.GY * g.ZD);
    void *ws = nullptr;
    if ((r = get_workspace(uc, ctx, (n_bz + n_bx + n_by) * sizeof(float2), &ws))) return r;
    float2 *bz = (float2 *)ws, *bx = bz + n_bz, *by = bx + n_bx;

    const float *din = dev_ptr<float>(input);
    const long in_sy = input->dim[1].stride, out_sy = output->dim[1].stride;
    hipStream_t st = ctx.stream;
    // float4 staging: rows 16-byte aligned at every staged column group (cells start at multiples of 8 minus 4)
    const int vec = ((uintptr_t)din % 16 == 0 && in_sy % 4 == 0 && floor_div(g.ix0, 4) * 4 == g.ix0) ? 1 : 0;
    // 32-bit addressing (see bg_blur_slice): strides and row counts below 2^23, every buffer's byte span below 2^31
    const long lim = 1L << 23, span = 1L << 29;
    const bool a32 = in_sy >= 0 && out_sy >= 0 && in_sy < lim && out_sy < lim && input->dim[1].extent < lim && oh < lim &&
                     in_sy * (long)input->dim[1].extent < span && out_sy * (long)oh < span && (long)g.HX * g.HY < (1L << 24) &&
                     (long)g.HX * g.HY * g.ZD < (span >> 1) && !env_flag("HLMI_BG_NO_A32");
    // round 6: HLMI_BG_ONE_LAUNCH=1 runs grids of at most 16 planes as ONE launch (bg_blur_slice<.., HIST = true>).  Built, bit-exact
    // (the parity tests run both paths) and NOT the default: 0.0297 against 0.0180 ms per call at 1920x1080 — a cell's histogram is
    // rebuilt by every tile whose blur footprint holds it (4.4 x the work) and the histogram is work, not latency, at the clock a
    // call + sync pattern runs at (profiles/r06_bg_one_launch_ab.txt)
    const bool one_launch = g.ZD <= FZ && env_flag("HLMI_BG_ONE_LAUNCH");
    if (one_launch) {
#define HLMI_BG_ONE(ZP_, A32_)                                                                                                       \
    HLMI_LAUNCH(uc, "bg_fused", st, (bg_blur_slice<32, ZP_, A32_, true>), dim3((ow + FPX - 1) / FPX, (oh + 31) / 32), dim3(256), 0, din, \
                in_sy, g, (const float2 *)nullptr, dev_ptr<float>(output), out_sy, ox0, oy0, ow, oh)
        if (g.ZD <= 12) {
            if (a32) HLMI_BG_ONE(12, true);
            else HLMI_BG_ONE(12, false);
        } else {
            if (a32) HLMI_BG_ONE(16, true);
            else HLMI_BG_ONE(16, false);
        }
#undef HLMI_BG_ONE
        mark_output_written(output);
        return 0;
    }
    if (g.ZD <= 12) {
        constexpr int HC = HTH / 12;
        HLMI_LAUNCH(uc, "bg_histogram_blurz", st, bg_histogram_blurz_par<12>, dim3((g.HX + HC - 1) / HC, g.HY), dim3(HC * 12), 0, din,
                    in_sy, g, bz, vec);
    } else if (g.ZD <= 16) {
        constexpr int HC = HTH / 16;
        HLMI_LAUNCH(uc, "bg_histogram_blurz", st, bg_histogram_blurz_par<16>, dim3((g.HX + HC - 1) / HC, g.HY), dim3(HC * 16), 0, din,
                    in_sy, g, bz, vec);
    } else {
        int T = 64;
        while (T > 1 && (size_t)g.ZH * 2 * T * sizeof(float) > 65536) T >>= 1;
        size_t sh = (size_t)g.ZH * 2 * T * sizeof(float);
        HLMI_LAUNCH(uc, "bg_histogram_blurz", st, bg_histogram_blurz, dim3((g.HX + T - 1) / T, g.HY), dim3(T), sh, din, in_sy, g, bz);
    }
    if (g.ZD <= FZ) {   // (grids of more planes — r_sigma < 1/14.5 — take the serial histogram and the three separate launches)
#define HLMI_BG_FUSED(ZP_, A32_)                                                                                                     \
    HLMI_LAUNCH(uc, "bg_blur_slice", st, (bg_blur_slice<32, ZP_, A32_>), dim3((ow + FPX - 1) / FPX, (oh + 31) / 32), dim3(256), 0, din, \
                in_sy, g, bz, dev_ptr<float>(output), out_sy, ox0, oy0, ow, oh)
        if (g.ZD <= 12) {
            if (a32) HLMI_BG_FUSED(12, true);
            else HLMI_BG_FUSED(12, false);
        } else {
            if (a32) HLMI_BG_FUSED(16, true);
            else HLMI_BG_FUSED(16, false);
        }
#undef HLMI_BG_FUSED
    } else {
        HLMI_LAUNCH(uc, "bg_blurx", st, bg_blurx, dim3((g.GX + 63) / 64, g.HY, g.ZD), dim3(64), 0, bz, g, bx);
        HLMI_LAUNCH(uc, "bg_blury", st, bg_blury, dim3((g.GX + 63) / 64, g.GY, g.ZD), dim3(64), 0, bx, g, by);
        HLMI_LAUNCH(uc, "bg_slice", st, bg_slice, dim3((ow + 255) / 256, oh), dim3(256), 0, din, in_sy, g, by, dev_ptr<float>(output),
                    out_sy, ox0, oy0, ow);
    }
    mark_output_written(output);
    return 0;
}

extern "C" int bilateral_grid_argv(void **a) {
    return bilateral_grid((halide_buffer_t *)a[0], *(float *)a[1], (halide_buffer_t *)a[2]);
}
extern "C" const halide_filter_metadata_t *bilateral_grid_metadata(void) { return &bg_md; }
extern "C" int bilateral_grid_auto_schedule(halide_buffer_t *input, float r_sigma, halide_buffer_t *output) {
    return bilateral_grid(input, r_sigma, output);
}
