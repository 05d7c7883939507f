// local_laplacian.hip — gfx950 implementation of the reference's local_laplacian AOT pipeline.
//
// Algorithm: /root/reference/apps/local_laplacian/local_laplacian_generator.cpp:18-87 (+ downsample
// :267-273, upsample :276-282); boundary: `int local_laplacian(halide_buffer_t*, int32_t, float, float,
// halide_buffer_t*)` (:12-16, :287).  pyramid_levels J = 8 is a compile-time GeneratorParam (:10).
//
// Layout in HBM.  All Funcs of the reference are total functions on Z^2 and only the input is
// edge-clamped (:28), so level j must be known a few pixels OUTSIDE ceil(W/2^j).  Because the clamp
// makes level 0 constant beyond the image edge, level j is constant beyond
//     lo_{j+1} = floor((lo_j - 2)/2),   hi_{j+1} = floor((hi_j + 2)/2)      (lo_0, hi_0 = input min/max)
// so each level is stored on a box [so_j, hi_j] x [loy_j, hiy_j] (so_j <= lo_j, see below) and reads are
// clamped to that box — bit-identical to evaluating on the unbounded regions the reference's bounds
// inference demands.
//   level j (1..7):  float G[j][K+1][h_j][ws_j] — planes 0..K-1 = gPyramid[j](.,.,k), plane K = inGPyramid[j]
//                    float OUT[j][h_j][ws_j]    — outGPyramid[j] on R_j
//   row stride ws_j = w_j rounded up to 4 floats, plane stride a multiple of 4 floats: every row starts
//   16-byte aligned.  The storage origin so_j is chosen (one column left of lo_j when necessary) so that the
//   4-column groups the down-sampling waves load and the 2-column groups they store are naturally aligned.
//
// Kernels (DESIGN.md §4 has the measurements).  The common geometry — levels == 8, 8-byte-aligned u16 planes whose width is a
// multiple of 4, three channels, even output origin and width — runs 5 launches per frame:
//   ll_remap_lut    remap LUT (generator :23-25); cached per (device, levels, alpha)
//   ll_down01e      levels 0 -> 1 -> 2 of all K+1 planes in ONE walk; emits outLPyramid[0] (one plane) and three planes of level 1
//                   instead of the K+1-plane level-1 pyramid (round 4's dataflow; ll_down01f = round 3's, which stores them all)
//   ll_down_strip2  levels 3 and 4 from level 2 in one launch (round 5; ll_down_strip = one level per launch, other chains)
//   ll_down_multi   levels 5..7 from level 4 in one launch;  ll_up_multi: outGPyramid[3] from levels 3..7 in one launch
//   ll_up0h         outGPyramid[2] and outGPyramid[1] (LDS tiles) -> outGPyramid[0] = upsample + outLPyramid[0] -> recolour -> u16 store
//                   (HLMI_LL_FUSE_UP2=0: outGPyramid[2] by an ll_up launch of its own)
// Everything else (other `levels`, odd widths or strides, fewer channels, odd output origins) takes the general kernels:
//   ll_down0        level 0 -> 1, any K (chunks of 8 planes), vector or element-wise loads;  ll_down_strip:1
//   ll_top, ll_up   outGPyramid[J-1], outGPyramid[j]: pointwise, data-dependent plane gathers
//   ll_up0 / ll_up0f  outGPyramid[0] + recolour from the materialised level-1 planes
#include "hlmi_device_math.h"
#include "hlmi_internal.h"

#include <atomic>
#include <mutex>
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include <vector>

using namespace hlmi;

namespace {

constexpr int J = 8;         // pyramid_levels (local_laplacian_generator.cpp:10)
constexpr int MAX_K = 32;    // entries of the `Levels` kernel argument (the levels == KCH fast kernels read the first KCH);
                             // `levels` itself is unbounded above, as in the reference (generator :13: Input<int> levels, no range)
constexpr int STRIP = 126;   // level-(j+1) columns one wave produces per row (lanes 0..62 store a float2)
constexpr int KCH = 8;       // planes of gPyramid one ll_down0 pass keeps in registers
constexpr int D0_THREADS = 256;  // ll_down0 workgroup: one wave per SIMD, two workgroups per CU (242 VGPRs)

struct Level {
    int lox, loy;            // absolute coordinates of storage element [0][0]
    int w, h;                // valid extent; reads clamp to [0,w-1] x [0,h-1] (storage coordinates)
    int ws;                  // row stride in floats (multiple of 4)
    size_t ps;               // plane stride in floats (multiple of 4)
    int rx0, rx1, ry0, ry1;  // R_j: region of outGPyramid[j] that is needed (absolute)
    float *g;                // (K+1) planes
    float *out;              // outGPyramid[j]
    // how level j is produced from level j-1 by the strip kernels
    bool odd;                // lane's source columns are 2P-1..2P+2 (odd source origin) instead of 2P-2..2P+1
    int nsx;                 // strips per row
};

// HLMI_LL_PROBE=1 (compile-time, csrc/Makefile VARIANT): s_memrealtime stamps (10 ns ticks) of the phases of the two big kernels, summed over
// workgroups into g_probe; hlmi_debug_ll_probe prints and clears them.  Timing experiments only.
#ifndef HLMI_LL_PROBE
#define HLMI_LL_PROBE 0
#endif
// A/B switch of round 6's 8-byte emission gathers (ll_down01e, em_gather): `make VARIANT=_emb64off EXTRA=-DHLMI_LL_EM_B64=0`
#ifndef HLMI_LL_EM_B64
#define HLMI_LL_EM_B64 1
#endif
#if HLMI_LL_PROBE
__device__ unsigned long long g_probe[32];
#define LL_PROBE_T(var) const unsigned long long var = wall_clock64()   // s_memrealtime: constant 100 MHz
#define LL_PROBE_ADD(slot, val) do { if ((threadIdx.x & 63) == 0) atomicAdd(&g_probe[slot], (unsigned long long)(val)); } while (0)   // only at the very end of a kernel: the atomics are VMEM traffic
#else
#define LL_PROBE_T(var)
#define LL_PROBE_ADD(slot, val)
#endif

// HLMI_LL_RESIDENCY=1 (compile-time, `make VARIANT=_res EXTRA=-DHLMI_LL_RESIDENCY=1`): which kernels' workgroups share a compute unit?
// Workgroups of ll_down01e (kind 0) and ll_up0h (kind 1) count themselves in and out per (XCD, SE, SH, CU) — HW_REG_XCC_ID / HW_REG_HW_ID —
// and record, when they start, how many workgroups of the OTHER kind and of their OWN kind were resident on their CU
// (g_res_hist[8 kind + min(other, 3)], [8 kind + 4 + min(own, 3)]); hlmi_debug_ll_residency reads and clears.  scripts/residency_probe.py.
#ifndef HLMI_LL_RESIDENCY
#define HLMI_LL_RESIDENCY 0
#endif
// experiments: wave priority (s_setprio 0..3) of the two big kernels; `make VARIANT=_prio EXTRA=-DHLMI_LL_D01_PRIO=3`
#ifndef HLMI_LL_D01_PRIO
#define HLMI_LL_D01_PRIO 0
#endif
#ifndef HLMI_LL_UP0_PRIO
#define HLMI_LL_UP0_PRIO 0
#endif
#if HLMI_LL_RESIDENCY
__device__ int g_res[2][4096];
__device__ unsigned long long g_res_hist[16];
struct ResidencyProbe {
    int kind, slot;
    bool on;
    __device__ explicit ResidencyProbe(int k) : kind(k), slot(0), on(threadIdx.x == 0) {
        if (on) {
            unsigned hw, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            slot = (int)(((xcc & 0xfu) << 8) | ((hw >> 8) & 0xffu));   // CU_ID [11:8], SH_ID [12], SE_ID [15:13]
            const int own = atomicAdd(&g_res[kind][slot], 1);
            const int other = atomicAdd(&g_res[1 - kind][slot], 0);
            atomicAdd(&g_res_hist[8 * kind + min(other, 3)], 1ull);
            atomicAdd(&g_res_hist[8 * kind + 4 + min(own, 3)], 1ull);
        }
    }
    __device__ ~ResidencyProbe() {
        if (on) atomicSub(&g_res[kind][slot], 1);
    }
};
#define LL_RESIDENCY(kind) ResidencyProbe residency_probe_(kind)
#else
#define LL_RESIDENCY(kind)
#endif

struct Geometry {
    int K, half;             // levels, (K-1)*256
    float Km1, inv_Km1;
    int ix0, ix1, iy0, iy1;  // clamp box of the input (absolute coordinates)
};

// ---------------------------------------------------------------------------------------------------
// remap(i) = alpha * fx * exp(-fx*fx/2), fx = i/256  (generator :23-25)
__global__ void ll_remap_lut(float *lut, int half, float alpha) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > 2 * half) return;
    float fx = (float)(i - half) * (1.0f / 256.0f);
    lut[i] = (alpha * fx) * dev::halide_exp(((-fx) * fx) * 0.5f);
}

// Workgroups are dispatched round-robin over the 8 XCDs (block b -> XCD b % 8, observed, speed only).  This
// remap hands every XCD a CONTIGUOUS range of logical blocks, so that neighbouring work units (which share
// halo rows) run on the same XCD and find each other's rows in its L2.
__device__ __forceinline__ int xcd_block() {
    const int nb8 = gridDim.x >> 3, b = blockIdx.x;
    return (b < (nb8 << 3)) ? (b & 7) * nb8 + (b >> 3) : b;
}

// `3.0f * (b + c)` has one use and feeds the add: dev::mad contracts the pair under the fma canon (hlmi_device_math.h)
__device__ __forceinline__ float down4(float a, float b, float c, float d) {
    return (dev::mad(3.0f, b + c, a) + d) * 0.125f;  // (:270-271), "/ 8.0f" == "* 0.125f"
}
// The two passes of one level, with the vertical pass's "* 0.125f" deferred: scaling by a power of two commutes
// with every rounding of the chain (no operand is anywhere near the subnormal range), so
//   down4(down4(..), ..) == down4_tail(down4_raw(..), ..)   bit for bit, one multiply less per vertical result.
__device__ __forceinline__ float down4_raw(float a, float b, float c, float d) { return dev::mad(3.0f, b + c, a) + d; }
__device__ __forceinline__ float down4_tail(float a, float b, float c, float d) {
    return (dev::mad(3.0f, b + c, a) + d) * 0.015625f;
}

// gray = 0.299f * floating(0) + 0.587f * floating(1) + 0.114f * floating(2), floating = u16 / 65535.0f (:32, :36), in the
// form the reference's simplifier leaves it in: x / c0 -> x * fold(1 / c0) (src/Simplify_Div.cpp:204), then
// (x * c0) * c1 -> x * fold(c0 * c1) (src/Simplify_Mul.cpp:70; constants fold in double and round to float32,
// src/IRMatch.h:1014-1016) — one multiply per channel by C_c = float(double(float(1.0 / 65535.0)) * double(coef_c)).
// The three constants are pinned against the oracle's by tests/test_local_laplacian.py.
__device__ __forceinline__ float gray_from(uint16_t r, uint16_t g, uint16_t b) {
    constexpr float s = (float)(1.0 / 65535.0);
    constexpr float C0 = (float)((double)s * (double)0.299f), C1 = (float)((double)s * (double)0.587f),
                    C2 = (float)((double)s * (double)0.114f);
    return dev::mad((float)b, C2, dev::mad2((float)r, C0, (float)g, C1));   // (r C0 + g C1) + b C2
}
__device__ __forceinline__ int idx_of(float gray, float Km1, int half) {
    return dev::clampi((int)((gray * Km1) * 256.0f), 0, half);  // (:42-43)
}
// gPyramid[0](x,y,k) = beta*(gray - level) + level + remap(idx - 256*k)   (:41-44); `l` = remap value
// B1: beta == 1.0f exactly, and 1.0f * x == x bit for bit, so the multiply is skipped
template<bool B1 = false>
__device__ __forceinline__ float g0_val(float gray, float level, float beta, float l) {
    return B1 ? ((gray - level) + level) + l : dev::mad(beta, gray - level, level) + l;
}

// ---- lane <-> lane exchange: DPP wave shifts (v_mov_b32_dpp wave_shr:1 / wave_shl:1; gfx9-family only,
// verified on gfx950 by ll_dpp_probe, which tests/ run through hlmi_debug_dpp_probe).
__device__ __forceinline__ float lane_prev(float v) {  // value held by lane-1 (0 for lane 0)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float lane_next(float v) {  // value held by lane+1 (0 for lane 63)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}
__global__ void ll_dpp_probe(int *ok) {
    float v = (float)threadIdx.x;
    float p = lane_prev(v), n = lane_next(v);
    bool good = (threadIdx.x == 0 || p == v - 1.0f) && (threadIdx.x == 63 || n == v + 1.0f);
    unsigned long long m = __ballot(good);
    if (threadIdx.x == 0) *ok = (m == ~0ull) ? 1 : 0;
}

// Horizontal 1-3-3-1 of one output row.  dy[0..3] = vertical results of the lane's 4 adjacent source
// columns q0..q0+3.  Returns the two outputs at destination columns (P, P+1), valid for lanes 0..62.
//   !ODD: q0 = 2P-2: out(P-1) = f(left, dy0, dy1, dy2) and out(P) = f(dy1, dy2, dy3, right) are computed
//         here, out(P+1) is the next lane's out(P-1)
//    ODD: q0 = 2P-1: out(P) = f(dy0..dy3), out(P+1) = f(dy2, dy3, next.dy0, next.dy1)
template<bool ODD>
__device__ __forceinline__ float2 hpair(const float (&dy)[4]) {  // dy = down4_raw results
    if (!ODD) {
        float left = lane_prev(dy[3]);
        float right = lane_next(dy[0]);
        float o0 = down4_tail(left, dy[0], dy[1], dy[2]);
        float o1 = down4_tail(dy[1], dy[2], dy[3], right);
        float o2 = lane_next(o0);
        return make_float2(o1, o2);
    } else {
        float r0 = lane_next(dy[0]), r1 = lane_next(dy[1]);
        float o0 = down4_tail(dy[0], dy[1], dy[2], dy[3]);
        float o1 = down4_tail(dy[2], dy[3], r0, r1);
        return make_float2(o0, o1);
    }
}

// ---------------------------------------------------------------------------------------------------
// Clamped 4-column groups.  A lane needs source columns o..o+3 (storage coordinates, o a multiple of 4 by
// construction) clamped to [0, w-1].  It always loads the ALIGNED quad at oq = clamp(o, 0, (w-1) & ~3) and,
// only in waves that touch an edge (wave-uniform branch), picks element sel[i] = clamp(o+i, 0, w-1) - oq of
// that quad for column i.  Memory-safe when the row is readable up to ((w-1)|3), which the padded level
// rows always are and the input is when its width is a multiple of 4.
struct QuadSel {
    int oq;        // storage column of the quad that is loaded
    int sel[4];    // which of its elements column i takes
    bool plain;    // sel == {0,1,2,3}
};
__device__ __forceinline__ QuadSel quad_sel(int o, int w) {
    QuadSel q;
    q.oq = dev::clampi(o, 0, (w - 1) & ~3);
#pragma unroll
    for (int i = 0; i < 4; i++) q.sel[i] = dev::clampi(o + i, 0, w - 1) - q.oq;
    q.plain = (o >= 0) && (o + 3 <= w - 1);
    return q;
}
template<typename T>
__device__ __forceinline__ T pick4(T x, T y, T z, T w, int s) {
    T lo = (s & 1) ? y : x, hi = (s & 1) ? w : z;
    return (s & 2) ? hi : lo;
}

// ---------------------------------------------------------------------------------------------------
// The u16 frames are touched exactly twice (input: the down and the up kernel; output: written once) and never again, while the
// pyramid planes in between are re-read within tens of microseconds.  NT marks a frame access non-temporal so that it does
// not displace the planes from L2 / Infinity Cache.  Round 3 measured nothing from it (the frame rate was set elsewhere);
// with the re-cut dataflow the frame rate with four frames in flight is set by how the memory system digests the traffic (§4: the three
// parts of a frame add up, whatever runs beside them), and there the hints are worth 6-7 % (84.0 -> 77.9 us per frame,
// five alternating runs each): non-temporal input loads in both kernels, outLPyramid[0] stores and loads, output stores; the
// level-1 and level-2 planes stay cached (non-temporal: slower).  On a stream that owns the device they cost 2-3 %, so the
// kernels take it as a template parameter and the host asks the runtime which kind of stream it is on.
typedef unsigned short us4_t __attribute__((ext_vector_type(4)));
template<bool NT = false>
__device__ __forceinline__ ushort4 ld_frame4(const uint16_t *p) {
    if (NT) {
        us4_t v = __builtin_nontemporal_load(reinterpret_cast<const us4_t *>(p));
        return make_ushort4(v.x, v.y, v.z, v.w);
    }
    return *reinterpret_cast<const ushort4 *>(p);
}
template<bool NT = false>
__device__ __forceinline__ ushort2 ld_frame2(const void *sbase, uint32_t byte_off) {
    const uint32_t *q = reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(sbase) + byte_off);
    const uint32_t w = NT ? __builtin_nontemporal_load(q) : *q;
    return make_ushort2((uint16_t)(w & 0xffffu), (uint16_t)(w >> 16));
}
template<bool NT = false>
__device__ __forceinline__ void st_frame2(void *p, uint16_t a, uint16_t b) {
    const uint32_t w = (uint32_t)a | ((uint32_t)b << 16);
    if (NT) __builtin_nontemporal_store(w, reinterpret_cast<uint32_t *>(p));
    else *reinterpret_cast<uint32_t *>(p) = w;
}

// ---------------------------------------------------------------------------------------------------
// level 0 -> level 1
struct Raw {
    ushort4 c0, c1, c2;  // the lane's 4 columns of the three (clamped) colour channels of one input row
};
struct Levels {
    float v[MAX_K];      // level_k = k * (1 / (levels - 1))  (:41), kernel-argument (scalar) operands
};

template<bool VEC, bool NT = false>
__device__ __forceinline__ void load_raw(Raw &r, const uint16_t *__restrict__ rp, long co0, long co1, long co2,
                                         int oq, const int (&xo)[4]) {
    if (VEC) {
        r.c0 = ld_frame4<NT>(rp + co0 + oq);
        r.c1 = ld_frame4<NT>(rp + co1 + oq);
        r.c2 = ld_frame4<NT>(rp + co2 + oq);
    } else {
        r.c0 = make_ushort4(rp[co0 + xo[0]], rp[co0 + xo[1]], rp[co0 + xo[2]], rp[co0 + xo[3]]);
        r.c1 = make_ushort4(rp[co1 + xo[0]], rp[co1 + xo[1]], rp[co1 + xo[2]], rp[co1 + xo[3]]);
        r.c2 = make_ushort4(rp[co2 + xo[0]], rp[co2 + xo[1]], rp[co2 + xo[2]], rp[co2 + xo[3]]);
    }
}

// 165 VGPRs: 3 waves/SIMD; the scalar-load variant (odd strides / widths) needs more and is not the fast path
// Register budget: 72 (vertical window state) + 16 (double-buffered LUT values) + 32 (gray / LUT positions of this
// and the next row pair) + 12 (raw input in flight) + 18 (results awaiting their stores) + temporaries: two waves per
// SIMD (launch bound); the LUT reads of plane k+1 are issued before the arithmetic of plane k so that the LDS latency
// hides inside the wave itself.
template<bool ODD, bool VEC, bool LUT_LDS, bool B1>
__global__ __launch_bounds__(D0_THREADS, 2) void ll_down0(const uint16_t *__restrict__ in, long in_sy, long co0, long co1,
                                                   long co2, Geometry gm, Levels lev, float beta,
                                                   const float *__restrict__ lut_g, float *__restrict__ g1, int Xs,
                                                   int loy1, int w1, int h1, int ws1, size_t ps1, int nsx, int nsy,
                                                   int nunits) {
    extern __shared__ float slut[];
    if (LUT_LDS) {
        for (int i = threadIdx.x; i <= 2 * gm.half; i += D0_THREADS) slut[i] = lut_g[i];
        __syncthreads();
    }
    const float *lut = LUT_LDS ? slut : lut_g;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform -> scalar control flow
    const int unit = xcd_block() * (D0_THREADS / 64) + wave;
    if (unit >= nunits) return;
    const int lane = threadIdx.x & 63;
    const int sy = unit % nsy, sx = unit / nsy;   // sy fastest: an XCD walks down a strip, halo rows stay in its L2
    const int off = STRIP * sx + 2 * lane;        // destination storage column of the lane's pair
    const int P = Xs + off;                       // absolute level-1 column of the pair
    const int q0 = ODD ? 2 * P - 1 : 2 * P - 2;   // absolute level-0 column of the lane's first source column
    const int iw = gm.ix1 - gm.ix0 + 1, ih = gm.iy1 - gm.iy0;
    const QuadSel qs = quad_sel(q0 - gm.ix0, iw);
    int xo[4];
#pragma unroll
    for (int i = 0; i < 4; i++) xo[i] = qs.oq + qs.sel[i];
    const bool edge_wave = VEC && __any(!qs.plain);
    const bool store_ok = (lane < 63) && (off < w1);
    const int t0 = (int)((long)sy * h1 / nsy), t1 = (int)((long)(sy + 1) * h1 / nsy) - 1;  // level-1 storage rows

    auto row_ptr = [&](int y_abs) -> const uint16_t * {
        return in + (long)(dev::clampi(y_abs - gm.iy0, 0, ih)) * in_sy;
    };

    for (int kb = 0; kb < gm.K; kb += KCH) {
        const int nk = min(KCH, gm.K - kb);
        const bool with_in = (kb == 0);  // plane K (inGPyramid[1], :58-61) rides along with the first chunk
        const int lbase = gm.half - 256 * (kb + KCH - 1);  // lut index of plane kb+7 is idx + lbase (< 0 only if unused)
        float level[KCH];                // level_k of this chunk's planes, pinned in scalar registers
#pragma unroll
        for (int kk = 0; kk < KCH; kk++) {
            // level_k = k * (1 / (levels - 1)) (:41): the same f32 multiply the host does for `lev`, for any number of planes
            const float lk = (float)(kb + kk) * gm.inv_Km1;   // uniform, but computed on the vector unit: back to a scalar register
            level[kk] = kb + kk < MAX_K ? lev.v[kb + kk]
                                        : __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, lk)));
            asm volatile("" : "+s"(level[kk]));
        }

        // vertical-window state: two row pairs' worth of planes, ping-ponged between (a, b) and (a2, b2) so that
        // neither the state nor the prefetched input rows are ever copied between registers (a copy at the loop
        // back-edge would force s_waitcnt vmcnt(0), i.e. a wait for this iteration's STORES, every iteration)
        float a[KCH + 1][4], b[KCH + 1][4], a2[KCH + 1][4], b2[KCH + 1][4];
        // gray and LUT position of the lane's 4 pixels of one input row
        auto gray_row = [&](const Raw &r, float (&gr)[4], int (&li)[4]) {
            const uint16_t rr[4] = {r.c0.x, r.c0.y, r.c0.z, r.c0.w};
            const uint16_t gg[4] = {r.c1.x, r.c1.y, r.c1.z, r.c1.w};
            const uint16_t bb[4] = {r.c2.x, r.c2.y, r.c2.z, r.c2.w};
            float g[4];
#pragma unroll
            for (int i = 0; i < 4; i++) g[i] = gray_from(rr[i], gg[i], bb[i]);
            if (edge_wave) {
#pragma unroll
                for (int i = 0; i < 4; i++) gr[i] = pick4(g[0], g[1], g[2], g[3], qs.sel[i]);
            } else {
#pragma unroll
                for (int i = 0; i < 4; i++) gr[i] = g[i];
            }
#pragma unroll
            for (int i = 0; i < 4; i++) li[i] = idx_of(gr[i], gm.Km1, gm.half) + lbase;
        };
        // remap values of plane kb+kk for the 8 pixels of a row pair
        auto lut_issue = [&](int kk, const int (&l0)[4], const int (&l1)[4], float (&dst)[8]) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                dst[i] = lut[l0[i] + 256 * (KCH - 1 - kk)];
                dst[4 + i] = lut[l1[i] + 256 * (KCH - 1 - kk)];
            }
        };
        // Walk the planes of one row pair (r0 = gray/positions of the first row, r1 of the second).  F(kk, i, v0, v1)
        // receives, for plane slot kk and column i, the gPyramid[0] (or gray, slot KCH) values of the two rows;
        // G(kk) runs once per plane after its four columns.
        auto for_planes = [&](const float (&g0r)[4], const int (&l0)[4], const float (&g1r)[4], const int (&l1)[4],
                              auto &&F, auto &&G) {
            float lv[2][8];
            lut_issue(0, l0, l1, lv[0]);
#pragma unroll
            for (int kk = 0; kk <= KCH; kk++) {
                if (kk + 1 < KCH && kk + 1 < nk) lut_issue(kk + 1, l0, l1, lv[(kk + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
                if (kk < KCH ? (kk < nk) : with_in) {
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        float v0 = kk < KCH ? g0_val<B1>(g0r[i], level[kk < KCH ? kk : 0], beta, lv[kk & 1][i]) : g0r[i];
                        float v1 = kk < KCH ? g0_val<B1>(g1r[i], level[kk < KCH ? kk : 0], beta, lv[kk & 1][4 + i]) : g1r[i];
                        F(kk, i, v0, v1);
                    }
                    G(kk);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };

        const int T0 = loy1 + t0;
        {
            Raw ra, rb;
            load_raw<VEC>(ra, row_ptr(2 * T0 - 1), co0, co1, co2, qs.oq, xo);
            load_raw<VEC>(rb, row_ptr(2 * T0), co0, co1, co2, qs.oq, xo);
            float gra[4], grb[4];
            int la[4], lb[4];
            gray_row(ra, gra, la);
            gray_row(rb, grb, lb);
            for_planes(gra, la, grb, lb, [&](int kk, int i, float v0, float v1) { a[kk][i] = v0, b[kk][i] = v1; },
                       [&](int) {});
        }
        // One output row.  VMEM loads and stores share one in-order-per-type counter (vmcnt) and may complete out of
        // order with respect to EACH OTHER, so waiting for a load means waiting until every store issued before the
        // wait has been acknowledged as well.  The step is therefore ordered so that the only wait sits where all
        // pending VMEM operations are one full step old:
        //   1. arithmetic of all planes (row pair prepared by the previous step), results kept in registers
        //   2. wait for the NEXT row pair's raw input (issued at the end of the previous step) -> gray / LUT positions
        //   3. this row's stores, back to back          4. loads of the row pair after next
        struct RowPair {
            float g0[4], g1[4];  // gray of the two input rows (LUT positions are recomputed from them: 8 VGPRs less)
        };
        Raw rc, rd;
        auto lut_pos = [&](const float (&g)[4], int (&l)[4]) {
#pragma unroll
            for (int i = 0; i < 4; i++) l[i] = idx_of(g[i], gm.Km1, gm.half) + lbase;
        };
        float2 *stage = reinterpret_cast<float2 *>(slut + (LUT_LDS ? ((2 * gm.half + 1 + 1) & ~1) : 0)) + threadIdx.x;
        auto step = [&](int t, const RowPair &cur, RowPair &nxt, float (&ia)[KCH + 1][4], float (&ib)[KCH + 1][4],
                        float (&oa)[KCH + 1][4], float (&ob)[KCH + 1][4]) {
            const int T = loy1 + t;
            int l0[4], l1[4];
            lut_pos(cur.g0, l0);
            lut_pos(cur.g1, l1);
            float dy[4];
            // results wait in LDS (one float2 slot per plane and lane) for the store batch at the end of the step
            for_planes(cur.g0, l0, cur.g1, l1,
                       [&](int kk, int i, float cv, float dv) {
                           dy[i] = down4_raw(ia[kk][i], ib[kk][i], cv, dv);
                           oa[kk][i] = cv;
                           ob[kk][i] = dv;
                       },
                       [&](int kk) { stage[kk * D0_THREADS] = hpair<ODD>(dy); });
            if (t < t1) {
                int unused[4];
                gray_row(rc, nxt.g0, unused);
                gray_row(rd, nxt.g1, unused);
            }
            __builtin_amdgcn_sched_barrier(0);
            float *drow = g1 + (size_t)t * ws1 + off;
#pragma unroll
            for (int kk = 0; kk <= KCH; kk++) {
                if (kk < KCH ? (kk < nk) : with_in) {
                    const int plane = (kk < KCH) ? kb + kk : gm.K;
                    const float2 o = stage[kk * D0_THREADS];
                    if (store_ok) {
                        *reinterpret_cast<float2 *>(drow + (size_t)plane * ps1) = o;
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (t + 1 < t1) {
                load_raw<VEC>(rc, row_ptr(2 * T + 5), co0, co1, co2, qs.oq, xo);
                load_raw<VEC>(rd, row_ptr(2 * T + 6), co0, co1, co2, qs.oq, xo);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        RowPair p0, p1;
        load_raw<VEC>(rc, row_ptr(2 * T0 + 1), co0, co1, co2, qs.oq, xo);
        load_raw<VEC>(rd, row_ptr(2 * T0 + 2), co0, co1, co2, qs.oq, xo);
        {
            int unused[4];
            gray_row(rc, p0.g0, unused);
            gray_row(rd, p0.g1, unused);
        }
        if (t0 < t1) {
            load_raw<VEC>(rc, row_ptr(2 * T0 + 3), co0, co1, co2, qs.oq, xo);
            load_raw<VEC>(rd, row_ptr(2 * T0 + 4), co0, co1, co2, qs.oq, xo);
        }
        for (int t = t0; t <= t1;) {
            step(t, p0, p1, a, b, a2, b2);
            if (++t > t1) break;
            step(t, p1, p0, a2, b2, a, b);
            ++t;
        }
    }
}

// ---- ll_down01f: levels 1 AND 2 (all K+1 planes of both) from the input in one walk, for levels == KCH, a vectorisable input and
// the remap table in LDS — the ll_down_strip:1 launch, its
// read of the level-1 planes (75 MB at 4K) and its launch latency disappear.
//   * A unit owns n level-2 rows [A, B] and the level-1 rows [2A, 2B+1] under them.  gPyramid[2] row Y needs level-1 rows
//     2Y-1 .. 2Y+2, so the walk computes level-1 rows 2A-1 .. 2B+2 — one more above and below than it stores (its
//     vertical neighbours compute the same two rows with the same operations; nothing is exchanged between units).
//     Rows or columns of level 1 outside its box are simply computed from the clamped input: level 1 is constant
//     beyond its box, which is what a clamped read of the stored plane would have returned.
//   * EXCH: the two extra rows per unit (+24 % of the walk at 4K, where a unit is 4-5 level-2 rows) are avoided inside a
//     workgroup: its four waves own vertically adjacent units of one strip, a unit is shifted to the level-1 rows
//     [2A-1, 2B], and what it then lacks for its last level-2 row — rows 2B+1 and 2B+2 — are the FIRST two rows of the
//     wave below, which publishes them in LDS after its first two steps (one workgroup barrier); only the bottom wave of
//     a workgroup walks the two extra rows itself, and gets one level-2 row less to own, so that all four walk 2 n rows.
//   * Vertical 1-3-3-1 of level 1 -> 2 incrementally, in the association of down4_raw: on odd steps pc = a + 3 (b + c),
//     on even steps pc + d.  The two live values per (plane, column) sit in wave-private LDS (the walk has no registers
//     left), moved as float2 = both columns of the lane.
//   * Horizontal pass: the lane's level-1 pair is columns (P, P+1); with P odd (ODD1) the level-2 column (P+1)/2 takes
//     (own.x, own.y, next.x, next.y), with P even column P/2 takes (prev.y, own.x, own.y, next.x): two DPP moves, one
//     level-2 value per lane.  A strip therefore advances by S2 = 62 level-2 columns (61 when both origins are even: lane
//     63's pair is incomplete), i.e. 2 S2 level-1 columns, and starts at Pbase <= so1 far enough left for column so2.
struct D01Args {
    const uint16_t *in;
    long in_sy, co0, co1, co2;
    float beta;
    const float *lut_g;
    float *g1;
    int so1, loy1, w1, h1, ws1;
    size_t ps1;
    float *g2;
    int so2, loy2, w2, h2, ws2;
    size_t ps2;
    int Pbase, S2, nsx, nsy, nunits;
    unsigned nsy_magic;        // floor(2^32 / nsy) + 1: x / nsy == umulhi(x, magic) for x * nsy < 2^32; 0 when nsy == 1
    int rows_base, rows_rem;   // h2 / nsy, h2 % nsy
};
constexpr int D01_STATE = 2 * (KCH + 1) * 64;  // float2 slots of one wave's level-1 -> 2 window state (and of the two rows it publishes)

template<bool ODD0, bool ODD1, bool B1, bool EXCH>
__global__ __launch_bounds__(D0_THREADS, 2) void ll_down01f(D01Args p, Geometry gm, Levels lev) {
    extern __shared__ float slut[];
    LL_PROBE_T(pt0);
    for (int i = threadIdx.x; i <= 2 * gm.half; i += D0_THREADS) slut[i] = p.lut_g[i];
    __syncthreads();
    LL_PROBE_T(pt1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    int sx, A, B;              // strip; level-2 rows [A, B] owned
    bool self_halo = true;     // the walk itself continues over the two rows below the unit
    bool publish = false;      // first two rows go to LDS for the wave above
    if (EXCH) {
        // nsy = workgroups per strip; the workgroup's rows [GA, GB] are cut into 4 runs of n (the last one shorter)
        // no integer divisions here: every wave of the launch runs this before its first load (the host passes the
        // reciprocal of nsy and the quotient / remainder of h2 / nsy; rows are dealt base + 1 to the first `rem` groups)
        const int wg = xcd_block();
        sx = p.nsy_magic ? (int)__umulhi((unsigned)wg, p.nsy_magic) : wg;
        const int gy = wg - sx * p.nsy;
        const int GA = p.loy2 + gy * p.rows_base + min(gy, p.rows_rem);
        const int GB = GA + p.rows_base + (gy < p.rows_rem ? 1 : 0) - 1;
        const int n = (GB - GA + 1 + 1 + 3) >> 2;
        A = GA + wave * n, B = min(A + n - 1, GB);
        if (A > GB) {          // no rows left for this wave: it only keeps the barrier count
            __syncthreads();
            return;
        }
        self_halo = !(wave < 3 && A + n <= GB);
        publish = wave > 0;
    } else {
        const int unit = xcd_block() * (D0_THREADS / 64) + wave;
        if (unit >= p.nunits) return;
        sx = p.nsy_magic ? (int)__umulhi((unsigned)unit, p.nsy_magic) : unit;
        const int sy = unit - sx * p.nsy;
        A = p.loy2 + sy * p.rows_base + min(sy, p.rows_rem);
        B = A + p.rows_base + (sy < p.rows_rem ? 1 : 0) - 1;
    }
    const int P = p.Pbase + 2 * p.S2 * sx + 2 * lane;   // absolute level-1 column of the lane's pair
    const int q0 = ODD0 ? 2 * P - 1 : 2 * P - 2;
    const int iw = gm.ix1 - gm.ix0 + 1, ih = gm.iy1 - gm.iy0;
    const QuadSel qs = quad_sel(q0 - gm.ix0, iw);
    int xo[4];
#pragma unroll
    for (int i = 0; i < 4; i++) xo[i] = qs.oq + qs.sel[i];
    const bool edge_wave = __any(!qs.plain);
    // level-1 rows [T0, T1] computed, [Ts0, Ts1] stored
    const int T0 = 2 * A - 1, T1 = self_halo ? 2 * B + 2 : 2 * B;
    const int Ts0 = max(EXCH ? 2 * A - 1 : 2 * A, p.loy1), Ts1 = min(EXCH ? 2 * B : 2 * B + 1, p.loy1 + p.h1 - 1);
    const int off1 = P - p.so1;
    const bool st1_ok = lane < p.S2 && off1 >= 0 && off1 < p.w1;
    const int X2 = ODD1 ? (P + 1) >> 1 : P >> 1;
    const int off2 = X2 - p.so2;
    const bool st2_ok = (ODD1 ? lane < p.S2 : (lane >= 1 && lane <= p.S2)) && off2 >= 0 && off2 < p.w2;
    float2 *st2 = reinterpret_cast<float2 *>(slut + ((2 * gm.half + 2) & ~1)) + wave * D01_STATE + lane;
    float2 *pub_all = reinterpret_cast<float2 *>(slut + ((2 * gm.half + 2) & ~1)) + (D0_THREADS / 64) * D01_STATE + lane;
    float2 *pub_mine = pub_all + (wave - 1) * D01_STATE, *pub_next = pub_all + wave * D01_STATE;   // [row][plane][64]
    const int lbase = gm.half - 256 * (KCH - 1);
    // level_k in VECTOR registers: a VALU instruction with an SGPR operand does not pair with the other wave's
    // (1.8 instead of 1.1 ns per instruction at two waves per SIMD, scripts/ubench/valu_dep.hip)
    float level[KCH];
#pragma unroll
    for (int kk = 0; kk < KCH; kk++) {
        level[kk] = lev.v[kk];
        asm volatile("" : "+v"(level[kk]));
    }
    auto walk = [&](auto edge_tag) {
    constexpr bool EDGE = decltype(edge_tag)::value;
#if HLMI_LL_PROBE
    unsigned long long pr_a0 = 0, pr_a = 0, pr_3 = 0;
#endif
    auto load_row = [&](Raw &r, int y_abs) {
        const uint16_t *rp = p.in + (long)(dev::clampi(y_abs - gm.iy0, 0, ih)) * p.in_sy;
        load_raw<true>(r, rp, p.co0, p.co1, p.co2, qs.oq, xo);
    };
    auto u16s = [&](const ushort4 &c, uint16_t (&o)[4]) {
        const uint2 w = __builtin_bit_cast(uint2, c);
        o[0] = (uint16_t)(w.x & 0xffffu), o[1] = (uint16_t)(w.x >> 16);
        o[2] = (uint16_t)(w.y & 0xffffu), o[3] = (uint16_t)(w.y >> 16);
    };
    struct Row {
        float g[4];
        int l[4];
    };
    auto prep_row = [&](const Raw &r, Row &o) {
        uint16_t rr[4], gg[4], bb[4];
        u16s(r.c0, rr), u16s(r.c1, gg), u16s(r.c2, bb);
#pragma unroll
        for (int i = 0; i < 4; i++) o.g[i] = gray_from(rr[i], gg[i], bb[i]);
        if (EDGE) {
            const float g0 = o.g[0], g1v = o.g[1], g2v = o.g[2], g3 = o.g[3];
#pragma unroll
            for (int i = 0; i < 4; i++) o.g[i] = pick4(g0, g1v, g2v, g3, qs.sel[i]);
        }
        // l = LDS byte offset of the LUT entry of plane KCH-1; made opaque so that the per-plane constant 1024 (KCH-1-k)
        // stays a separate addend and lands in the ds_read offset field instead of costing one v_add per gather
#pragma unroll
        for (int i = 0; i < 4; i++) {
            o.l[i] = (min((int)((o.g[i] * gm.Km1) * 256.0f), gm.half) + lbase) * 4;
            asm volatile("" : "+v"(o.l[i]));
        }
    };
    auto lut_issue = [&](int kk, const Row &r0, const Row &r1, float (&dst)[8]) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            dst[i] = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(slut) + r0.l[i] + 1024 * (KCH - 1 - kk));
            dst[4 + i] = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(slut) + r1.l[i] + 1024 * (KCH - 1 - kk));
        }
    };
    // In passes over the 8 pixels (not pixel by pixel): a wave issues dependent VALU instructions only every ~3.7 ns but
    // independent ones every ~2.2 ns (scripts/ubench/valu_dep.hip), and with two waves per SIMD nothing else hides it.
    auto plane_vals = [&](int kk, const Row &r0, const Row &r1, const float (&lv)[8], float (&v0)[4], float (&v1)[4]) {
        if (kk < KCH) {
            const float L = level[kk < KCH ? kk : 0];
            float t[8];
#pragma unroll
            for (int i = 0; i < 4; i++) t[i] = r0.g[i] - L, t[4 + i] = r1.g[i] - L;
            if (!B1 && !dev::CANON_FMA) {
#pragma unroll
                for (int i = 0; i < 8; i++) t[i] = p.beta * t[i];
            }
#pragma unroll
            for (int i = 0; i < 8; i++) t[i] = (!B1 && dev::CANON_FMA) ? __builtin_fmaf(p.beta, t[i], L) : t[i] + L;
#pragma unroll
            for (int i = 0; i < 4; i++) v0[i] = t[i] + lv[i], v1[i] = t[4 + i] + lv[4 + i];
        } else {
#pragma unroll
            for (int i = 0; i < 4; i++) v0[i] = r0.g[i], v1[i] = r1.g[i];
        }
    };
    auto vpass = [&](const float (&ia)[4], const float (&ib)[4], const float (&c)[4], const float (&d)[4], float (&o)[4]) {
        float t[4];   // down4_raw, column-parallel
#pragma unroll
        for (int i = 0; i < 4; i++) t[i] = ib[i] + c[i];
        if (!dev::CANON_FMA) {
#pragma unroll
            for (int i = 0; i < 4; i++) t[i] = 3.0f * t[i];
        }
#pragma unroll
        for (int i = 0; i < 4; i++) t[i] = dev::CANON_FMA ? __builtin_fmaf(3.0f, t[i], ia[i]) : ia[i] + t[i];
#pragma unroll
        for (int i = 0; i < 4; i++) o[i] = t[i] + d[i];
    };

    float a[KCH + 1][4], b[KCH + 1][4], a2[KCH + 1][4], b2[KCH + 1][4];
    Raw rc, rd;
    {
        // all four rows of the first step are requested at once: every wave of the launch starts here at the same time
        // and the first (cold) round trip to memory is the longest of the walk
        Raw ra, rb;
        load_row(ra, 2 * T0 - 1);
        load_row(rb, 2 * T0);
        load_row(rc, 2 * T0 + 1);
        load_row(rd, 2 * T0 + 2);
#if HLMI_LL_PROBE
        LL_PROBE_T(ptA0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        LL_PROBE_T(ptA);
        pr_a0 = ptA0, pr_a = ptA;
#endif
        Row r0, r1;
        prep_row(ra, r0);
        prep_row(rb, r1);
        float lv[2][8];
        lut_issue(0, r0, r1, lv[0]);
#pragma unroll
        for (int kk = 0; kk <= KCH; kk++) {
            if (kk + 1 < KCH) lut_issue(kk + 1, r0, r1, lv[(kk + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            plane_vals(kk, r0, r1, lv[kk & 1], a[kk], b[kk]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // One level-1 row T.  PH = 0: T odd relative to the unit (rows "a"/"c" of the level-2 window), PH = 1: rows "b"/"d" —
    // the step that completes level-2 row (T - 2) / 2.
    auto step = [&](auto ph_tag, int T, const Row &c0, const Row &c1, Row &n0, Row &n1, float (&ia)[KCH + 1][4],
                    float (&ib)[KCH + 1][4], float (&oa)[KCH + 1][4], float (&ob)[KCH + 1][4], float2 *pub) {
        constexpr int PH = decltype(ph_tag)::value;
        float lv[2][8], dy[2][4];
        float2 res[KCH + 1];
        float2 sa, sb;  // level-2 window state of the plane whose level-1 row is being finished
        const bool out2 = PH == 1 && T >= 2 * A + 2;   // wave-uniform
        float *d2 = p.g2 + (size_t)(((T - 2) >> 1) - p.loy2) * p.ws2 + off2;
        auto level2 = [&](int k) {
            const float2 c = res[k];
            if (PH == 0) {
                float2 pc;
                pc.x = dev::mad(3.0f, sb.x + c.x, sa.x);
                pc.y = dev::mad(3.0f, sb.y + c.y, sa.y);
                st2[(2 * k + 1) * 64] = pc;
                st2[(2 * k) * 64] = c;
            } else {
                const float rx = sb.x + c.x, ry = sb.y + c.y;   // down4_raw of the lane's two level-1 columns
                float o;
                if (ODD1) {
                    const float nx = lane_next(rx), ny = lane_next(ry);
                    o = down4_tail(rx, ry, nx, ny);
                } else {
                    const float py = lane_prev(ry), nx = lane_next(rx);
                    o = down4_tail(py, rx, ry, nx);
                }
                if (out2 && st2_ok) d2[(size_t)k * p.ps2] = o;
                st2[(2 * k + 1) * 64] = c;
            }
        };
        auto state_issue = [&](int k) {
            if (PH == 0) sa = st2[(2 * k) * 64];
            sb = st2[(2 * k + 1) * 64];
        };
        lut_issue(0, c0, c1, lv[0]);
#pragma unroll
        for (int kk = 0; kk <= KCH; kk++) {
            if (kk > 0) state_issue(kk - 1);   // before the gathers: LDS returns in order, the state must not wait for them
            __builtin_amdgcn_sched_barrier(0);
            if (kk + 1 < KCH) lut_issue(kk + 1, c0, c1, lv[(kk + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            plane_vals(kk, c0, c1, lv[kk & 1], oa[kk], ob[kk]);
            vpass(ia[kk], ib[kk], oa[kk], ob[kk], dy[kk & 1]);
            if (kk > 0) {
                res[kk - 1] = hpair<ODD0>(dy[(kk - 1) & 1]);
                level2(kk - 1);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        state_issue(KCH);
        res[KCH] = hpair<ODD0>(dy[KCH & 1]);
        level2(KCH);
        // unconditional: after the last row rc / rd still hold the previous (valid) rows and the result is unused
        prep_row(rc, n0);
        prep_row(rd, n1);
        __builtin_amdgcn_sched_barrier(0);
        if (T >= Ts0 && T <= Ts1) {
            float *drow = p.g1 + (size_t)(T - p.loy1) * p.ws1 + off1;
            if (st1_ok) {
#pragma unroll
                for (int kk = 0; kk <= KCH; kk++) *reinterpret_cast<float2 *>(drow + (size_t)kk * p.ps1) = res[kk];
            }
        }
        if (EXCH && pub) {
#pragma unroll
            for (int kk = 0; kk <= KCH; kk++) pub[kk * 64] = res[kk];
        }
        __builtin_amdgcn_sched_barrier(0);
        if (T + 1 < T1) {
            load_row(rc, 2 * T + 5);
            load_row(rd, 2 * T + 6);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    Row p0, p1, p2, p3;
    prep_row(rc, p0);
    prep_row(rd, p1);
    load_row(rc, 2 * T0 + 3);   // T0 < T1 always: a unit walks at least 2 rows
    load_row(rd, 2 * T0 + 4);
    LL_PROBE_T(pt2);
    for (int T = T0; T <= T1; T += 2) {
        const bool first = EXCH && T == T0;
        step(std::integral_constant<int, 0>{}, T, p0, p1, p2, p3, a, b, a2, b2, first && publish ? pub_mine : nullptr);
        step(std::integral_constant<int, 1>{}, T + 1, p2, p3, p0, p1, a2, b2, a, b,
             first && publish ? pub_mine + (KCH + 1) * 64 : nullptr);
        if (first) __syncthreads();   // every wave of the workgroup that has rows passes here exactly once
#if HLMI_LL_PROBE
        if (first) { LL_PROBE_T(pt3); pr_3 = pt3; }
#endif
    }
    LL_PROBE_T(pt4);
    LL_PROBE_ADD(0, pt1 - pt0); LL_PROBE_ADD(1, 1); LL_PROBE_ADD(2, pt2 - pt1); LL_PROBE_ADD(3, pr_3 - pt2);
    LL_PROBE_ADD(16, pr_a - pr_a0); LL_PROBE_ADD(17, pr_a0 - pt1);
    LL_PROBE_ADD(4, pt4 - pt2); LL_PROBE_ADD(5, (T1 - T0 + 1)); LL_PROBE_ADD(6, pt4 - pt0);
    if (EXCH && !self_halo) {
        // level-2 row B: rows "c" (2B+1) and "d" (2B+2) are the first two rows of the wave below
        float *d2 = p.g2 + (size_t)(B - p.loy2) * p.ws2 + off2;
#pragma unroll
        for (int k = 0; k <= KCH; k++) {
            const float2 sa = st2[(2 * k) * 64], sb = st2[(2 * k + 1) * 64];
            const float2 c = pub_next[k * 64], d = pub_next[(KCH + 1 + k) * 64];
            const float rx = dev::mad(3.0f, sb.x + c.x, sa.x) + d.x, ry = dev::mad(3.0f, sb.y + c.y, sa.y) + d.y;
            float o;
            if (ODD1) {
                const float nx = lane_next(rx), ny = lane_next(ry);
                o = down4_tail(rx, ry, nx, ny);
            } else {
                const float py = lane_prev(ry), nx = lane_next(rx);
                o = down4_tail(py, rx, ry, nx);
            }
            if (st2_ok) d2[(size_t)k * p.ps2] = o;
        }
    }
    };  // walk
    if (edge_wave) walk(std::true_type{});
    else walk(std::false_type{});
}

// ---------------------------------------------------------------------------------------------------
// ll_down01e: ll_down01f re-cut so that the (K + 1)-plane level-1 pyramid never leaves the chip (round 4).
//
// outLPyramid[0](x, y) (:63-72) — ONE value per pixel — is a pure function of gray(x, y), of gPyramid[0](x, y, li / li + 1)
// (pointwise from gray and the remap table) and of gPyramid[1] at the 2 x 2 coarse pixels of the bilinear footprint (:276-282),
// planes li and li + 1.  A wave holds all of that at the moment it finishes level-1 row T: the gray values and table positions
// of level-0 rows 2T - 1 and 2T (the previous step's row pair, still in registers) and level-1 rows T - 1 and T of every plane
// (the raw rows of its level-1 -> 2 window state in LDS; the data-dependent plane index is an LDS address, the left
// neighbour's pair is the float2 before the lane's own).  So the wave emits outLPyramid[0] rows 2T - 1, 2T (4 bytes per
// pixel) and, of level 1, only what the collapse of level 1 reads at the coarse pixel itself (:63-72): inGPyramid[1] and
// gPyramid[1](., ., li1), gPyramid[1](., ., li1 + 1) for li1 of that coarse pixel — three planes instead of K + 1, and no
// data-dependent plane gathers from memory in the up pass: uniform noise costs what a smooth frame costs.
// The up pass becomes outGPyramid[0] = upsample(outGPyramid[1]) + outLPyramid[0] + recolouring (ll_up0h).  Every Func is
// still evaluated by the same operations on the same values: bit-identical to ll_down01f + ll_up0f.
//   * Level-1 -> 2 window state: the two LDS slots per (plane, lane) hold RAW level-1 rows only (rows T - 1 and T after step T);
//     the partial sum pc = a + 3 (b + c) between an odd step and the even step that completes a level-2 row stays in
//     registers (18; the kernel is launch-bound to two waves per SIMD = 256 VGPRs).  Same LDS footprint as ll_down01f.
//   * Who emits what: the steps T in [2A, 2B + 1] of a unit owning level-2 rows [A, B] emit level-0 rows [4A - 1, 4B + 2]; a
//     wave that takes rows 2B + 1, 2B + 2 from the wave below (EXCH) emits the last pair after its walk, from its own row 2B and
//     the published row 2B + 1.  Lanes 1 .. S2 emit (lane 0 has no left neighbour; strips advance by S2 lanes).
struct D01EArgs {
    D01Args d;
    float *outl0;              // outLPyramid[0] on [ix0, ix1] x [oy0, oy0 + oh - 1], row stride = input width (a multiple of 4)
    int oy0, oh;
    // EXCH: workgroups are numbered strip-fastest (wg = gy * nsx + sx; ll_down01f: row-fastest) so that the contiguous range
    // xcd_block() hands an XCD holds horizontal AND vertical neighbours: a strip's 512-byte row pieces start 16 bytes before a
    // 128-byte line, which its left neighbour reads too — from the same L2 then (68.8 -> 58.8 MB fetched per 4K frame)
    unsigned nsx_magic;        // floor(2^32 / nsx) + 1, 0 when nsx == 1
};
// Packed arithmetic: at two waves per SIMD this kernel is bound by how often ONE wave can issue (a wave issues an
// independent VALU instruction every ~2.1 ns whatever it is, scripts/ubench/valu_pk.hip: v_pk_add / mul / fma_f32 2.4 ns for
// two results against 2.1 ns for one), so every pointwise pass runs on column PAIRS held as <2 x float>: pair A = the
// lane's columns (0, 2), pair B = columns (1, 3) — pixels of equal x parity, which is what the horizontal lerps of the
// emission want.  (The same substitution made ll_up0f slower in round 2: at five waves per SIMD the VALU is throughput-bound
// and a packed instruction costs two issue slots.)  IEEE per component: bit-identical.
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 f2s(float v) { return f2{v, v}; }
__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
// dev::mad / dev::mad2 on column pairs: a * b + c and a * b + c * d in the canonical form the library is built for
__device__ __forceinline__ f2 mad_2(f2 a, f2 b, f2 c) { return dev::CANON_FMA ? fma2(a, b, c) : a * b + c; }
__device__ __forceinline__ f2 mad2_2(f2 a, f2 b, f2 c, f2 d) { return dev::CANON_FMA ? fma2(a, b, c * d) : a * b + c * d; }

template<bool ODD0, bool ODD1, bool B1, bool EXCH, bool NT>
__global__ __launch_bounds__(D0_THREADS, 2) void ll_down01e(D01EArgs pe, Geometry gm, Levels lev) {
    const D01Args &p = pe.d;
    LL_RESIDENCY(0);
#if HLMI_LL_D01_PRIO
    __builtin_amdgcn_s_setprio(HLMI_LL_D01_PRIO);   // experiment: this kernel's waves before the co-resident kernels' in the SIMD's issue arbitration
#endif
    extern __shared__ float slut[];
    for (int i = threadIdx.x; i <= 2 * gm.half; i += D0_THREADS) slut[i] = p.lut_g[i];
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    int sx, A, B;              // strip; level-2 rows [A, B] owned
    bool self_halo = true;     // the walk itself continues over the two rows below the unit
    bool publish = false;      // first two rows go to LDS for the wave above
    if (EXCH) {
        const int wg = xcd_block();
        const int gy = pe.nsx_magic ? (int)__umulhi((unsigned)wg, pe.nsx_magic) : wg;   // one strip: wg / 1
        sx = wg - gy * p.nsx;
        const int GA = p.loy2 + gy * p.rows_base + min(gy, p.rows_rem);
        const int GB = GA + p.rows_base + (gy < p.rows_rem ? 1 : 0) - 1;
        const int n = (GB - GA + 1 + 1 + 3) >> 2;
        A = GA + wave * n, B = min(A + n - 1, GB);
        if (A > GB) {          // no rows left for this wave: it only keeps the barrier count
            __syncthreads();
            return;
        }
        self_halo = !(wave < 3 && A + n <= GB);
        publish = wave > 0;
    } else {
        const int unit = xcd_block() * (D0_THREADS / 64) + wave;
        if (unit >= p.nunits) return;
        sx = p.nsy_magic ? (int)__umulhi((unsigned)unit, p.nsy_magic) : unit;
        const int sy = unit - sx * p.nsy;
        A = p.loy2 + sy * p.rows_base + min(sy, p.rows_rem);
        B = A + p.rows_base + (sy < p.rows_rem ? 1 : 0) - 1;
    }
    const int P = p.Pbase + 2 * p.S2 * sx + 2 * lane;   // absolute level-1 column of the lane's pair
    const int q0 = ODD0 ? 2 * P - 1 : 2 * P - 2;
    const int iw = gm.ix1 - gm.ix0 + 1, ih = gm.iy1 - gm.iy0;
    const QuadSel qs = quad_sel(q0 - gm.ix0, iw);
    int xo[4];
#pragma unroll
    for (int i = 0; i < 4; i++) xo[i] = qs.oq + qs.sel[i];
    const bool edge_wave = __any(!qs.plain);
    // level-1 rows [T0, T1] computed, [Ts0, Ts1] stored (their three planes), steps [2A, 2B + 1] emit outLPyramid[0]
    const int T0 = 2 * A - 1, T1 = self_halo ? 2 * B + 2 : 2 * B;
    const int Ts0 = max(EXCH ? 2 * A - 1 : 2 * A, p.loy1), Ts1 = min(EXCH ? 2 * B : 2 * B + 1, p.loy1 + p.h1 - 1);
    const int off1 = P - p.so1;
    const bool st1_ok = lane < p.S2 && off1 >= 0 && off1 < p.w1;
    const int X2 = ODD1 ? (P + 1) >> 1 : P >> 1;
    const int off2 = X2 - p.so2;
    const bool st2_ok = (ODD1 ? lane < p.S2 : (lane >= 1 && lane <= p.S2)) && off2 >= 0 && off2 < p.w2;
    // the lane's four level-0 columns are one aligned quad of the input's box: inside it entirely or not at all
    const bool em_ok = lane >= 1 && lane <= p.S2 && q0 >= gm.ix0 && q0 + 3 <= gm.ix1;
    float *const em_col = pe.outl0 + (q0 - gm.ix0);
    // LDS: [plane][slot][64 lanes] float2 per wave, then the published rows of waves 1..3 as [plane][row][64 lanes]
    f2 *st2 = reinterpret_cast<f2 *>(slut + ((2 * gm.half + 2) & ~1)) + wave * D01_STATE + lane;
    f2 *pub_all = reinterpret_cast<f2 *>(slut + ((2 * gm.half + 2) & ~1)) + (D0_THREADS / 64) * D01_STATE + lane;
    f2 *pub_mine = pub_all + (wave - 1) * D01_STATE, *pub_next = pub_all + wave * D01_STATE;
    const int lbase = gm.half - 256 * (KCH - 1);
    float level[KCH];
#pragma unroll
    for (int kk = 0; kk < KCH; kk++) {
        level[kk] = lev.v[kk];
        asm volatile("" : "+v"(level[kk]));
    }
    auto walk = [&](auto edge_tag) {
    constexpr bool EDGE = decltype(edge_tag)::value;
    auto load_row = [&](Raw &r, int y_abs) {
        const uint16_t *rp = p.in + (long)(dev::clampi(y_abs - gm.iy0, 0, ih)) * p.in_sy;
        load_raw<true, NT>(r, rp, p.co0, p.co1, p.co2, qs.oq, xo);
    };
    auto u16s = [&](const ushort4 &c, uint16_t (&o)[4]) {
        const uint2 w = __builtin_bit_cast(uint2, c);
        o[0] = (uint16_t)(w.x & 0xffffu), o[1] = (uint16_t)(w.x >> 16);
        o[2] = (uint16_t)(w.y & 0xffffu), o[3] = (uint16_t)(w.y >> 16);
    };
    struct Row {
        f2 g[2];     // gray: g[0] = columns (0, 2), g[1] = columns (1, 3)
        int l[4];    // LDS byte offset of the table entry of plane KCH-1, per column (see ll_down01f)
    };
    auto prep_row = [&](const Raw &r, Row &o) {
        uint16_t rr[4], gg[4], bb[4];
        u16s(r.c0, rr), u16s(r.c1, gg), u16s(r.c2, bb);
        constexpr float s = (float)(1.0 / 65535.0);
        constexpr float C0 = (float)((double)s * (double)0.299f), C1 = (float)((double)s * (double)0.587f),
                        C2 = (float)((double)s * (double)0.114f);   // gray_from's constants
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const f2 fr = {(float)rr[h], (float)rr[h + 2]}, fg = {(float)gg[h], (float)gg[h + 2]}, fb = {(float)bb[h], (float)bb[h + 2]};
            o.g[h] = mad_2(fb, f2s(C2), mad2_2(fr, f2s(C0), fg, f2s(C1)));   // gray_from, column pairs
        }
        if (EDGE) {
            const float g0 = o.g[0].x, g1v = o.g[1].x, g2v = o.g[0].y, g3 = o.g[1].y;
            o.g[0].x = pick4(g0, g1v, g2v, g3, qs.sel[0]), o.g[1].x = pick4(g0, g1v, g2v, g3, qs.sel[1]);
            o.g[0].y = pick4(g0, g1v, g2v, g3, qs.sel[2]), o.g[1].y = pick4(g0, g1v, g2v, g3, qs.sel[3]);
        }
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const f2 t = (o.g[h] * gm.Km1) * 256.0f;
            o.l[h] = (min((int)t.x, gm.half) + lbase) * 4;
            o.l[h + 2] = (min((int)t.y, gm.half) + lbase) * 4;
        }
#pragma unroll
        for (int i = 0; i < 4; i++) asm volatile("" : "+v"(o.l[i]));
    };
    // table values of plane kk for two rows: dst[0], dst[1] = row r0 pairs A, B; dst[2], dst[3] = row r1
    auto lut_issue = [&](int kk, const Row &r0, const Row &r1, f2 (&dst)[4]) {
        auto rd = [&](int l) {
            return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(slut) + l + 1024 * (KCH - 1 - kk));
        };
        dst[0].x = rd(r0.l[0]), dst[1].x = rd(r0.l[1]), dst[0].y = rd(r0.l[2]), dst[1].y = rd(r0.l[3]);
        dst[2].x = rd(r1.l[0]), dst[3].x = rd(r1.l[1]), dst[2].y = rd(r1.l[2]), dst[3].y = rd(r1.l[3]);
    };
    auto plane_vals = [&](int kk, const Row &r0, const Row &r1, const f2 (&lv)[4], f2 (&v0)[2], f2 (&v1)[2]) {
        if (B1 && kk == 0) {
            // level_0 = 0 * (1 / (levels - 1)) = +0 exactly and gray >= +0: (gray - 0) + 0 is gray bit for bit, two of the three
            // passes of this plane are not issued (beta == 1 only: beta * gray would have to be)
            v0[0] = r0.g[0] + lv[0], v0[1] = r0.g[1] + lv[1], v1[0] = r1.g[0] + lv[2], v1[1] = r1.g[1] + lv[3];
        } else if (kk < KCH) {
            const f2 L = f2s(level[kk < KCH ? kk : 0]);
            f2 t[4] = {r0.g[0] - L, r0.g[1] - L, r1.g[0] - L, r1.g[1] - L};
            if (!B1 && !dev::CANON_FMA) {
#pragma unroll
                for (int i = 0; i < 4; i++) t[i] = p.beta * t[i];
            }
#pragma unroll
            for (int i = 0; i < 4; i++) t[i] = (!B1 && dev::CANON_FMA) ? fma2(f2s(p.beta), t[i], L) : t[i] + L;
            v0[0] = t[0] + lv[0], v0[1] = t[1] + lv[1], v1[0] = t[2] + lv[2], v1[1] = t[3] + lv[3];
        } else {
            v0[0] = r0.g[0], v0[1] = r0.g[1], v1[0] = r1.g[0], v1[1] = r1.g[1];
        }
    };
    auto vpass = [&](const f2 (&ia)[2], const f2 (&ib)[2], const f2 (&c)[2], const f2 (&d)[2], f2 (&o)[2]) {
        f2 t[2] = {ib[0] + c[0], ib[1] + c[1]};   // down4_raw, column-parallel
        if (dev::CANON_FMA) {
            t[0] = fma2(f2s(3.0f), t[0], ia[0]), t[1] = fma2(f2s(3.0f), t[1], ia[1]);
        } else {
            t[0] = 3.0f * t[0], t[1] = 3.0f * t[1];
            t[0] = ia[0] + t[0], t[1] = ia[1] + t[1];
        }
        o[0] = t[0] + d[0], o[1] = t[1] + d[1];
    };
    // horizontal 1-3-3-1 (hpair<ODD0>) on the paired layout: dy[0] = columns (0, 2), dy[1] = columns (1, 3)
    auto hpair2 = [&](const f2 (&dy)[2]) {
        const float d4[4] = {dy[0].x, dy[1].x, dy[0].y, dy[1].y};
        const float2 r = hpair<ODD0>(d4);
        return f2{r.x, r.y};
    };
    // ---- outLPyramid[0] of the lane's quad in one level-0 row: rq / rt = the lane's float2 of plane 0 in the level-1 row whose
    // vertical weight is 1/4 / 3/4 (plane stride 128 float2 = 4 x 64 dwords in both the window slots and the published rows).
    // The arithmetic is ll_up0f's stage 2 (hl0 / hl1 / vl: lerps with the exact quarter product folded into an fma, :276-282
    // with the parities known).  Per pixel the pair is (plane li, plane li + 1): one ds_read2st64_b32 fetches both planes of a
    // coarse value (or both table entries), and every pass up to the final blend runs on the two planes at once.  The gathers
    // are issued by hand (the compiler pairs adjacent dwords instead and then shuffles them into plane pairs with v_mov's), five
    // per pixel; the LDS returns them in order, so a pixel's five are complete when at most 5 x (pixels requested after it) are
    // outstanding — em_wait routes the five results through that s_waitcnt.
    struct EmPix {
        f2 lut, qa, qb, ta, tb;
        float lif;
    };
    auto em_gather = [&](const Row &n, int i, const f2 *rq, const f2 *rt, EmPix &g) {
        constexpr int COL[4] = {ODD0 ? -1 : -2, -1, ODD0 ? 0 : -1, 0};   // first float of the coarse column pair, relative to the lane's own .x
        const uint32_t aq0 = (uint32_t)(size_t)rq, at0 = (uint32_t)(size_t)rt, alut = (uint32_t)(size_t)slut + 1024u * (KCH - 2);
        auto rd2 = [](uint32_t addr, auto swap_tag) {   // (dword at addr, dword at addr + 1024 bytes), or swapped
            f2 v;
            if (decltype(swap_tag)::value) asm volatile("ds_read2st64_b32 %0, %1 offset0:4" : "=v"(v) : "v"(addr) : "memory");
            else asm volatile("ds_read2st64_b32 %0, %1 offset1:4" : "=v"(v) : "v"(addr) : "memory");
            return v;
        };
        constexpr std::false_type asc{};
        constexpr std::true_type desc{};
        const int pos = n.l[i] - lbase * 4;                        // 4 x table index of the pixel
        const int li = min(pos >> 10, KCH - 2);                    // (int)(gray * (K-1)), clamped (:66); gray >= 0
        const uint32_t po = (uint32_t)(li << 10) + (uint32_t)(COL[i] * 4);
        g.lut = rd2(alut + (uint32_t)n.l[i] - (uint32_t)(li << 10), desc);   // plane li + 1's entry is the lower address
        if (HLMI_LL_EM_B64 && (COL[i] & 1) == 0) {
            // round 6: the coarse column pair is one lane's float2 — ONE ds_read_b64 per plane fetches both columns, at the full LDS rate:
            // lanes 8 bytes apart are conflict-free for 8-byte reads, while the dword gathers below walk the same rows at a stride of two
            // words (2-way bank conflicts, 8 LDS cycles per instruction instead of 4; the LDS pipe is this kernel's co-limit).  The
            // pairs arrive as (column c, column c + 1) per plane instead of (plane li, plane li + 1) per column: em_arith's `PAIR` form
            auto rd64 = [](uint32_t addr, auto plane1) {
                f2 v;
                if (decltype(plane1)::value) asm volatile("ds_read_b64 %0, %1 offset:1024" : "=v"(v) : "v"(addr) : "memory");
                else asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(addr) : "memory");
                return v;
            };
            g.qa = rd64(aq0 + po, asc), g.qb = rd64(aq0 + po, desc);
            g.ta = rd64(at0 + po, asc), g.tb = rd64(at0 + po, desc);
        } else {
            g.qa = rd2(aq0 + po, asc), g.qb = rd2(aq0 + po + 4u, asc);
            g.ta = rd2(at0 + po, asc), g.tb = rd2(at0 + po + 4u, asc);
        }
        g.lif = (float)li;
    };
    auto em_wait = [&](EmPix &g, auto after_tag) {   // `after` = gathers of how many pixels were requested after this one's
        constexpr int AFTER = decltype(after_tag)::value;
        static_assert(AFTER >= 0 && AFTER <= 3, "");
        if (AFTER == 0) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(g.lut), "+v"(g.qa), "+v"(g.qb), "+v"(g.ta), "+v"(g.tb));
        if (AFTER == 1) asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(g.lut), "+v"(g.qa), "+v"(g.qb), "+v"(g.ta), "+v"(g.tb));
        if (AFTER == 2) asm volatile("s_waitcnt lgkmcnt(10)" : "+v"(g.lut), "+v"(g.qa), "+v"(g.qb), "+v"(g.ta), "+v"(g.tb));
        if (AFTER == 3) asm volatile("s_waitcnt lgkmcnt(15)" : "+v"(g.lut), "+v"(g.qa), "+v"(g.qb), "+v"(g.ta), "+v"(g.tb));
    };
    // YODD: the level-0 row is odd (2T - 1) — its `zero` operand of the vertical lerp is the row weighted 1/4.
    auto em_arith = [&](const Row &n, int i, const EmPix &g, auto yodd_tag) {
        constexpr bool YODD = decltype(yodd_tag)::value;
        const bool xodd = ODD0 ? (i & 1) == 0 : (i & 1) == 1;
        // even x: lerp(f[c], f[c-1], 1/4) = f[c-1]/4 + 3 f[c]/4; odd x: lerp(f[c+1], f[c], 3/4) = f[c+1]/4 + 3 f[c]/4.  lerp(zero, one, w) =
        // zero (1 - w) + one w (:276-282, src/Lerp.cpp:127-128).  Canon 0: the exact quarter product folded into an fma changes
        // nothing; fma canon: the product with `zero` is the contracted one — the 3/4 one at an even coordinate, where the quarter
        // product of `one` is exact anyway, the 1/4 one at an odd coordinate (same instruction as canon 0 there).
        auto hl = [&](f2 fa, f2 fb) {
            if (xodd) return fma2(fb, f2s(0.25f), fa * 0.75f);
            return dev::CANON_FMA ? fma2(fb, f2s(0.75f), fa * 0.25f) : fma2(fa, f2s(0.25f), fb * 0.75f);
        };
        auto vl = [](f2 uq, f2 ut) {
            return (dev::CANON_FMA && !YODD) ? fma2(ut, f2s(0.75f), uq * 0.25f) : fma2(uq, f2s(0.25f), ut * 0.75f);
        };
        const float gr = n.g[i & 1][i >> 1];
        const float lf = gr * gm.Km1 - g.lif;   // `level` has two uses (:64-66): not contracted in either form
        const f2 lev = f2{g.lif, g.lif + 1.0f} * gm.inv_Km1;
        constexpr int COLP[4] = {ODD0 ? -1 : -2, -1, ODD0 ? 0 : -1, 0};
        f2 u;
        if (HLMI_LL_EM_B64 && (COLP[i] & 1) == 0) {
            // PAIR form (em_gather): qa / qb = (column c, column c + 1) of planes li / li + 1 in the row weighted 1/4, ta / tb in the other;
            // the same lerps on scalars (a packed instruction costs two issue slots here: the same issue time)
            auto hls = [&](float fa, float fb) {
                if (xodd) return __builtin_fmaf(fb, 0.25f, fa * 0.75f);
                return dev::CANON_FMA ? __builtin_fmaf(fb, 0.75f, fa * 0.25f) : __builtin_fmaf(fa, 0.25f, fb * 0.75f);
            };
            auto vls = [](float uq, float ut) {
                return (dev::CANON_FMA && !YODD) ? __builtin_fmaf(ut, 0.75f, uq * 0.25f) : __builtin_fmaf(uq, 0.25f, ut * 0.75f);
            };
            u = f2{vls(hls(g.qa.x, g.qa.y), hls(g.ta.x, g.ta.y)), vls(hls(g.qb.x, g.qb.y), hls(g.tb.x, g.tb.y))};
        } else {
            u = vl(hl(g.qa, g.qb), hl(g.ta, g.tb));
        }
        const f2 g2 = f2s(gr);
        const f2 l = (B1 ? ((g2 - lev) + lev) + g.lut : mad_2(f2s(p.beta), g2 - lev, lev) + g.lut) - u;   // g0_val of planes li, li + 1
        if (dev::CANON_FMA) return __builtin_fmaf(1.0f - lf, l.x, lf * l.y);
        const f2 m = f2{1.0f - lf, lf} * l;
        return m.x + m.y;
    };
    constexpr std::true_type row_odd{};
    constexpr std::false_type row_even{};
    constexpr std::integral_constant<int, 0> after0{};
    constexpr std::integral_constant<int, 1> after1{};
    constexpr std::integral_constant<int, 2> after2{};
    constexpr std::integral_constant<int, 3> after3{};
    // one row on its own (the seam pair after the walk): all four pixels requested, then finished in order
    auto emit_row = [&](const Row &n, const f2 *rq, const f2 *rt, float (&r)[4], auto yodd_tag) {
        EmPix g[4];
#pragma unroll
        for (int i = 0; i < 4; i++) em_gather(n, i, rq, rt, g[i]);
        em_wait(g[0], after3), r[0] = em_arith(n, 0, g[0], yodd_tag);
        em_wait(g[1], after2), r[1] = em_arith(n, 1, g[1], yodd_tag);
        em_wait(g[2], after1), r[2] = em_arith(n, 2, g[2], yodd_tag);
        em_wait(g[3], after0), r[3] = em_arith(n, 3, g[3], yodd_tag);
    };
    auto emit_store = [&](int y, const float (&r)[4]) {
        if (em_ok && y >= pe.oy0 && y < pe.oy0 + pe.oh) {
            typedef float f4_t __attribute__((ext_vector_type(4)));
            f4_t *const dst = reinterpret_cast<f4_t *>(em_col + (size_t)(y - pe.oy0) * iw);
            if (NT) __builtin_nontemporal_store(f4_t{r[0], r[1], r[2], r[3]}, dst);
            else *dst = f4_t{r[0], r[1], r[2], r[3]};
        }
    };
#if HLMI_LL_PROBE
    unsigned long long pr_planes = 0, pr_emit = 0, pr_prep = 0, pr_steps = 0;
    LL_PROBE_T(pe0);
    const unsigned long long pc0 = __builtin_readcyclecounter();
#endif
    f2 a[KCH + 1][2], b[KCH + 1][2], a2[KCH + 1][2], b2[KCH + 1][2];
    f2 pcr[KCH + 1];           // a + 3 (b + c) of the level-2 window between its third and fourth row
    Raw rc, rd;
    {
        Raw ra, rb;
        load_row(ra, 2 * T0 - 1);
        load_row(rb, 2 * T0);
        load_row(rc, 2 * T0 + 1);
        load_row(rd, 2 * T0 + 2);
        Row r0, r1;
        prep_row(ra, r0);
        prep_row(rb, r1);
        f2 lv[2][4];
        lut_issue(0, r0, r1, lv[0]);
#pragma unroll
        for (int kk = 0; kk <= KCH; kk++) {
            if (kk + 1 < KCH) lut_issue(kk + 1, r0, r1, lv[(kk + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            plane_vals(kk, r0, r1, lv[kk & 1], a[kk], b[kk]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // One level-1 row T.  PH = 0: the third row of a level-2 window (slot 0 takes it), PH = 1: the fourth (slot 1) — the step
    // that completes level-2 row (T - 2) / 2.  c0 / c1: level-0 rows 2T + 1, 2T + 2; n0 / n1 still hold rows 2T - 1, 2T (the
    // previous step's pair) until this step's end, when they take the rows after c0 / c1.
    auto step = [&](auto ph_tag, int T, const Row &c0, const Row &c1, Row &n0, Row &n1, f2 (&ia)[KCH + 1][2],
                    f2 (&ib)[KCH + 1][2], f2 (&oa)[KCH + 1][2], f2 (&ob)[KCH + 1][2], f2 *pub) {
        constexpr int PH = decltype(ph_tag)::value;
        f2 lv[2][4], dy[2][2];
        f2 res[KCH + 1];
        f2 sa, sb;
        const bool out2 = PH == 1 && T >= 2 * A + 2;   // wave-uniform
        float *d2 = p.g2 + (size_t)(((T - 2) >> 1) - p.loy2) * p.ws2 + off2;
        LL_PROBE_T(ps0);
        auto level2 = [&](int k) {
            const f2 c = res[k];
            if (PH == 0) {
                pcr[k] = mad_2(f2s(3.0f), sb + c, sa);
                st2[(2 * k) * 64] = c;
            } else {
                const f2 r = pcr[k] + c;   // down4_raw of the lane's two level-1 columns
                float o;
                if (ODD1) {
                    const float nx = lane_next(r.x), ny = lane_next(r.y);
                    o = down4_tail(r.x, r.y, nx, ny);
                } else {
                    const float py = lane_prev(r.y), nx = lane_next(r.x);
                    o = down4_tail(py, r.x, r.y, nx);
                }
                if (out2 && st2_ok) d2[(size_t)k * p.ps2] = o;
                st2[(2 * k + 1) * 64] = c;
            }
        };
        auto state_issue = [&](int k) {
            if (PH == 0) sa = st2[(2 * k) * 64], sb = st2[(2 * k + 1) * 64];
        };
        lut_issue(0, c0, c1, lv[0]);
#pragma unroll
        for (int kk = 0; kk <= KCH; kk++) {
            if (kk > 0) state_issue(kk - 1);   // before the gathers: LDS returns in order, the state must not wait for them
            __builtin_amdgcn_sched_barrier(0);
            if (kk + 1 < KCH) lut_issue(kk + 1, c0, c1, lv[(kk + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            plane_vals(kk, c0, c1, lv[kk & 1], oa[kk], ob[kk]);
            vpass(ia[kk], ib[kk], oa[kk], ob[kk], dy[kk & 1]);
            if (kk > 0) {
                res[kk - 1] = hpair2(dy[(kk - 1) & 1]);
                level2(kk - 1);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // Level-1 rows T - 1 and T of the planes 0 .. KCH-1 are in the two slots now (inGPyramid's plane follows below); other
        // lanes' entries are read from here on: the wave runs its LDS instructions in order, the compiler must not move reads above
        // the writes.  The emission's first gathers (level-0 row 2T - 1) are requested NOW — their round trip to the LDS passes under
        // the last plane, the published copy and the level-1 selection instead of stalling the wave (two waves per SIMD hide nothing).
        asm volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_sched_barrier(0);
        f2 *const rowT = st2 + (PH == 0 ? 0 : 64), *const rowP = st2 + (PH == 0 ? 64 : 0);   // rows T and T - 1
        const bool st1_row = T >= Ts0 && T <= Ts1;                                                  // wave-uniform
        const bool em_rows = T >= 2 * A && T <= 2 * B + 1;   // wave-uniform
        EmPix eg[4];
        if (em_rows) {
#pragma unroll
            for (int i = 0; i < 4; i++) em_gather(n0, i, rowT, rowP, eg[i]);   // odd row 2T - 1: coarse row T weighs 1/4
        }
        __builtin_amdgcn_sched_barrier(0);
        state_issue(KCH);
        res[KCH] = hpair2(dy[KCH & 1]);
        level2(KCH);
        if (EXCH && pub) {
#pragma unroll
            for (int kk = 0; kk <= KCH; kk++) pub[(2 * kk) * 64] = res[kk];
        }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        LL_PROBE_T(ps1);
        // Everything that is stored is computed first and stored AFTER prep_row: prep_row waits for the input rows requested at
        // the end of the previous step, and vmcnt counts in order — with this step's (conditional) stores issued before it, the
        // only wait that covers the loads is vmcnt(0), i.e. a round trip of the stores to memory in every step.
        float2 s0 = make_float2(0.0f, 0.0f), s1 = s0;
        const f2 sK = res[KCH];
        if (st1_row) {
            // of level 1 the up pass reads inGPyramid[1] and gPyramid[1](., ., li1 / li1 + 1) at the coarse pixel itself (:63-72)
            const int lx = dev::clampi((int)(sK.x * gm.Km1), 0, KCH - 2), ly = dev::clampi((int)(sK.y * gm.Km1), 0, KCH - 2);
            const float *fx = reinterpret_cast<const float *>(rowT) + (lx << 8), *fy = reinterpret_cast<const float *>(rowT) + (ly << 8) + 1;
            s0 = make_float2(fx[0], fy[0]), s1 = make_float2(fx[256], fy[256]);
        }
        float eo[4] = {0.0f, 0.0f, 0.0f, 0.0f}, ee[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (em_rows) {
            // row 2T - 1's pixels are finished one by one while row 2T's are requested into the registers they free: twenty
            // gathers stay in flight from the first request to the last pixel
            EmPix fg[4];
            em_wait(eg[0], after3), eo[0] = em_arith(n0, 0, eg[0], row_odd);
            __builtin_amdgcn_sched_barrier(0);
            em_gather(n1, 0, rowP, rowT, fg[0]);   // even row 2T
            __builtin_amdgcn_sched_barrier(0);
            em_wait(eg[1], after3), eo[1] = em_arith(n0, 1, eg[1], row_odd);
            __builtin_amdgcn_sched_barrier(0);
            em_gather(n1, 1, rowP, rowT, fg[1]);
            __builtin_amdgcn_sched_barrier(0);
            em_wait(eg[2], after3), eo[2] = em_arith(n0, 2, eg[2], row_odd);
            __builtin_amdgcn_sched_barrier(0);
            em_gather(n1, 2, rowP, rowT, fg[2]);
            __builtin_amdgcn_sched_barrier(0);
            em_wait(eg[3], after3), eo[3] = em_arith(n0, 3, eg[3], row_odd);
            __builtin_amdgcn_sched_barrier(0);
            em_gather(n1, 3, rowP, rowT, fg[3]);
            __builtin_amdgcn_sched_barrier(0);
            em_wait(fg[0], after3), ee[0] = em_arith(n1, 0, fg[0], row_even);
            em_wait(fg[1], after2), ee[1] = em_arith(n1, 1, fg[1], row_even);
            em_wait(fg[2], after1), ee[2] = em_arith(n1, 2, fg[2], row_even);
            em_wait(fg[3], after0), ee[3] = em_arith(n1, 3, fg[3], row_even);
        }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        LL_PROBE_T(ps2);
        // unconditional: after the last row rc / rd still hold the previous (valid) rows and the result is unused
        prep_row(rc, n0);
        prep_row(rd, n1);
        __builtin_amdgcn_sched_barrier(0);
        if (st1_row && st1_ok) {
            float *drow = p.g1 + (size_t)(T - p.loy1) * p.ws1 + off1;
            *reinterpret_cast<float2 *>(drow) = s0;
            *reinterpret_cast<float2 *>(drow + p.ps1) = s1;
            *reinterpret_cast<f2 *>(drow + (size_t)KCH * p.ps1) = sK;
        }
        if (em_rows) {
            emit_store(2 * T - 1, eo);
            emit_store(2 * T, ee);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (T + 1 < T1) {
            load_row(rc, 2 * T + 5);
            load_row(rd, 2 * T + 6);
        }
        __builtin_amdgcn_sched_barrier(0);
#if HLMI_LL_PROBE
        LL_PROBE_T(ps3);
        pr_planes += ps1 - ps0, pr_emit += ps2 - ps1, pr_prep += ps3 - ps2, pr_steps += 1;
#endif
    };
    Row p0, p1, p2, p3;
    prep_row(rc, p0);
    prep_row(rd, p1);
    load_row(rc, 2 * T0 + 3);   // T0 < T1 always: a unit walks at least 2 rows
    load_row(rd, 2 * T0 + 4);
    for (int T = T0; T <= T1; T += 2) {
        const bool first = EXCH && T == T0;
        step(std::integral_constant<int, 0>{}, T, p0, p1, p2, p3, a, b, a2, b2, first && publish ? pub_mine : nullptr);
        step(std::integral_constant<int, 1>{}, T + 1, p2, p3, p0, p1, a2, b2, a, b, first && publish ? pub_mine + 64 : nullptr);
        if (first) __syncthreads();   // every wave of the workgroup that has rows passes here exactly once
    }
#if HLMI_LL_PROBE
    LL_PROBE_T(pe1);
#endif
    if (EXCH && !self_halo) {
        // level-2 row B: rows "c" (2B+1) and "d" (2B+2) are the first two rows of the wave below
        float *d2 = p.g2 + (size_t)(B - p.loy2) * p.ws2 + off2;
#pragma unroll
        for (int k = 0; k <= KCH; k++) {
            const f2 sa = st2[(2 * k) * 64], sb = st2[(2 * k + 1) * 64];
            const f2 c = pub_next[(2 * k) * 64], d = pub_next[(2 * k + 1) * 64];
            const f2 r = mad_2(f2s(3.0f), sb + c, sa) + d;
            float o;
            if (ODD1) {
                const float nx = lane_next(r.x), ny = lane_next(r.y);
                o = down4_tail(r.x, r.y, nx, ny);
            } else {
                const float py = lane_prev(r.y), nx = lane_next(r.x);
                o = down4_tail(py, r.x, r.y, nx);
            }
            if (st2_ok) d2[(size_t)k * p.ps2] = o;
        }
        // outLPyramid[0] rows 4B + 1, 4B + 2 (the pair the last step brought in: p2 / p3) from level-1 rows 2B (slot 1) and 2B + 1
        // (published)
        {
            float eo[4], ee[4];
            emit_row(p2, pub_next, st2 + 64, eo, row_odd);
            emit_row(p3, st2 + 64, pub_next, ee, row_even);
            emit_store(4 * B + 1, eo);
            emit_store(4 * B + 2, ee);
        }
    }
#if HLMI_LL_PROBE
    LL_PROBE_T(pe2);
    LL_PROBE_ADD(27, __builtin_readcyclecounter() - pc0);   // s_memtime: shader-clock cycles of the wave's life (pe2 - pe0: 100 MHz ticks)
    LL_PROBE_ADD(28, pe2 - pe0);
    LL_PROBE_ADD(20, pr_planes); LL_PROBE_ADD(21, pr_emit); LL_PROBE_ADD(22, pr_prep); LL_PROBE_ADD(23, pr_steps);
    LL_PROBE_ADD(24, 1); LL_PROBE_ADD(25, pe1 - pe0); LL_PROBE_ADD(26, pe2 - pe1);
#endif
    };  // walk
    if (edge_wave) walk(std::true_type{});
    else walk(std::false_type{});
}

// ---------------------------------------------------------------------------------------------------
// level j -> j+1 (j >= 1): one wave = (strip of 126 destination columns, TY rows, one plane)
struct StripArgs {
    const float *src;        // level j: (K+1) planes
    int slox, sloy, sw, sh, sws;
    size_t sps;
    float *dst;              // level j+1
    int Xs, dloy, dw, dh, dws;
    size_t dps;
    int nsx, nsy, nunits;
};
// one unit; the whole wave is active (DPP exchanges, wave-uniform control flow)
template<bool ODD>
__device__ __forceinline__ void down_strip_unit(const StripArgs &a, int unit, int lane) {
    const int sy = unit % a.nsy, rest = unit / a.nsy, sx = rest % a.nsx, plane = rest / a.nsx;
    const int off = STRIP * sx + 2 * lane;
    const int P = a.Xs + off;
    const QuadSel qs = quad_sel((ODD ? 2 * P - 1 : 2 * P - 2) - a.slox, a.sw);
    const bool edge_wave = __any(!qs.plain);
    const bool store_ok = (lane < 63) && (off < a.dw);
    const int t0 = (int)((long)sy * a.dh / a.nsy), t1 = (int)((long)(sy + 1) * a.dh / a.nsy) - 1;
    const float *sp = a.src + (size_t)plane * a.sps + qs.oq;
    float *dp = a.dst + (size_t)plane * a.dps + off;
    auto row = [&](int y_abs) -> float4 {
        float4 v = *reinterpret_cast<const float4 *>(sp + (size_t)dev::clampi(y_abs - a.sloy, 0, a.sh - 1) * a.sws);
        if (edge_wave) {
            v = make_float4(pick4(v.x, v.y, v.z, v.w, qs.sel[0]), pick4(v.x, v.y, v.z, v.w, qs.sel[1]),
                            pick4(v.x, v.y, v.z, v.w, qs.sel[2]), pick4(v.x, v.y, v.z, v.w, qs.sel[3]));
        }
        return v;
    };
    const int T0 = a.dloy + t0;
    float4 ra = row(2 * T0 - 1), rb = row(2 * T0), rc = row(2 * T0 + 1), rd = row(2 * T0 + 2);
    for (int t = t0; t <= t1; t++) {
        const int T = a.dloy + t;
        float4 nc = rc, nd = rd;
        if (t < t1) {
            nc = row(2 * T + 3);
            nd = row(2 * T + 4);
        }
        float dy[4] = {down4_raw(ra.x, rb.x, rc.x, rd.x), down4_raw(ra.y, rb.y, rc.y, rd.y), down4_raw(ra.z, rb.z, rc.z, rd.z),
                       down4_raw(ra.w, rb.w, rc.w, rd.w)};
        float2 o = hpair<ODD>(dy);
        if (store_ok) *reinterpret_cast<float2 *>(dp + (size_t)t * a.dws) = o;
        ra = rc, rb = rd, rc = nc, rd = nd;
    }
}
template<bool ODD>
__global__ __launch_bounds__(256) void ll_down_strip(StripArgs a) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int unit = xcd_block() * 4 + wave;
    if (unit >= a.nunits) return;
    down_strip_unit<ODD>(a, unit, threadIdx.x & 63);
}

// ---- ll_down_strip2: levels j+1 AND j+2 of the stored planes of level j in one launch (round 5; j = 2 in the default chain:
// ll_down_strip:2 and ll_down_strip:3 become one launch — one dependent-launch gap of the chain between the two big kernels less).
// ll_down01f's scheme on stored planes: a wave owns RPU rows [A, B] of level j+2 of ONE plane and a strip of S2 of its columns,
// walks the level-(j+1) rows 2A-1 .. 2B+2 under them (stores 2A .. 2B+1: the two outer rows are the neighbours' — recomputed,
// identical operations, identical bits), lane = 4 adjacent level-j columns -> 2 level-(j+1) columns (ll_down_strip's hpair),
// and carries the level-(j+1) -> (j+2) vertical window of its column pair in REGISTERS (one plane per wave: two raw rows and the
// partial sum), the horizontal pass by DPP as in ll_down01f.  Every source row of the unit (2 NT + 2 float4 per lane) is REQUESTED
// before the first is used: the launch is one memory round trip, not one per row — these levels are latency, not bandwidth.
// Rows / columns of level j+1 outside its box are computed from the clamped level-j reads: level j+1 is constant beyond its box,
// which is what a clamped read of the stored plane returns (see ll_down01f).
constexpr int S2_RPU = 2;                 // level-(j+2) rows per unit
constexpr int S2_NT = 2 * S2_RPU + 2;     // level-(j+1) rows a unit walks
constexpr int S2_NSRC = 2 * S2_NT + 2;    // level-j rows it reads
// ll_mid's control words, one per 128-byte line: [0] planes finished, [1 .. MID_FLAGS] replicated "levels are there" flags,
// [1 + MID_FLAGS + plane] producer blocks of that plane finished
constexpr int MID_FLAGS = 64, MID_LINE = 32, MID_PLANES = MAX_K + 1;
constexpr int MID_WORDS = MID_LINE * (1 + MID_FLAGS + MID_PLANES);
struct Strip2Args {
    const float *src;        // level j, (K+1) planes
    int slox, sloy, sw, sh, sws;
    size_t sps;
    float *g1;               // level j+1
    int so1, loy1, w1, h1, ws1;
    size_t ps1;
    float *g2;               // level j+2
    int so2, loy2, w2, h2, ws2;
    size_t ps2;
    int Pbase, S2, nsx, nsy, nunits;   // nunits = planes * nsx * nsy
    unsigned *ctr;           // ll_mid's producer count of this call: zeroed here (nullptr: the chain runs ll_down_multi / ll_up_multi)
};
template<bool ODD0, bool ODD1>
__global__ __launch_bounds__(256) void ll_down_strip2(Strip2Args p) {
    if (p.ctr && blockIdx.x == 0 && threadIdx.x < MID_WORDS / MID_LINE) __hip_atomic_store(p.ctr + MID_LINE * (int)threadIdx.x, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int unit = xcd_block() * 4 + wave;
    if (unit >= p.nunits) return;
    const int lane = threadIdx.x & 63;
    const int sy = unit % p.nsy, rest = unit / p.nsy, sx = rest % p.nsx, plane = rest / p.nsx;   // sy fastest: vertical neighbours share an L2
    const int A = p.loy2 + sy * S2_RPU, B = min(A + S2_RPU - 1, p.loy2 + p.h2 - 1);
    const int P = p.Pbase + 2 * p.S2 * sx + 2 * lane;   // absolute level-(j+1) column of the lane's pair
    const QuadSel qs = quad_sel((ODD0 ? 2 * P - 1 : 2 * P - 2) - p.slox, p.sw);
    const bool edge_wave = __any(!qs.plain);
    const int T0 = 2 * A - 1;                            // first level-(j+1) row walked; the walk always has S2_NT steps
    const int Ts0 = max(2 * A, p.loy1), Ts1 = min(2 * B + 1, p.loy1 + p.h1 - 1);
    const int off1 = P - p.so1;
    const bool st1_ok = lane < p.S2 && off1 >= 0 && off1 < p.w1;
    const int X2 = ODD1 ? (P + 1) >> 1 : P >> 1;
    const int off2 = X2 - p.so2;
    const bool st2_ok = (ODD1 ? lane < p.S2 : (lane >= 1 && lane <= p.S2)) && off2 >= 0 && off2 < p.w2;
    const float *sp = p.src + (size_t)plane * p.sps + qs.oq;
    float4 r[S2_NSRC];                                   // level-j rows 2 T0 - 1 .. 2 T0 + 2 NT (clamped to the level's box)
#pragma unroll
    for (int i = 0; i < S2_NSRC; i++) r[i] = *reinterpret_cast<const float4 *>(sp + (size_t)dev::clampi(2 * T0 - 1 + i - p.sloy, 0, p.sh - 1) * p.sws);
    if (edge_wave) {
#pragma unroll
        for (int i = 0; i < S2_NSRC; i++) {
            const float4 v = r[i];
            r[i] = make_float4(pick4(v.x, v.y, v.z, v.w, qs.sel[0]), pick4(v.x, v.y, v.z, v.w, qs.sel[1]),
                               pick4(v.x, v.y, v.z, v.w, qs.sel[2]), pick4(v.x, v.y, v.z, v.w, qs.sel[3]));
        }
    }
    float *d1 = p.g1 + (size_t)plane * p.ps1 + off1, *d2 = p.g2 + (size_t)plane * p.ps2 + off2;
    float2 s0 = make_float2(0.0f, 0.0f), s1 = s0, pc = s0;   // raw level-(j+1) rows T - 2 / T - 1 of the window; a + 3 (b + c)
#pragma unroll
    for (int t = 0; t < S2_NT; t++) {
        const int T = T0 + t;
        const float4 a = r[2 * t], b = r[2 * t + 1], c = r[2 * t + 2], d = r[2 * t + 3];
        const float dy[4] = {down4_raw(a.x, b.x, c.x, d.x), down4_raw(a.y, b.y, c.y, d.y), down4_raw(a.z, b.z, c.z, d.z),
                             down4_raw(a.w, b.w, c.w, d.w)};
        const float2 res = hpair<ODD0>(dy);
        if (T >= Ts0 && T <= Ts1 && st1_ok) *reinterpret_cast<float2 *>(d1 + (size_t)(T - p.loy1) * p.ws1) = res;
        if ((t & 1) == 0) {      // third row of a level-(j+2) window (the first step's window is incomplete: its result is dropped)
            pc.x = dev::mad(3.0f, s1.x + res.x, s0.x);
            pc.y = dev::mad(3.0f, s1.y + res.y, s0.y);
            s0 = res;
        } else {                 // fourth row: completes level-(j+2) row (T - 2) / 2
            const float rx = pc.x + res.x, ry = pc.y + res.y;   // down4_raw of the lane's two level-(j+1) columns
            float o;
            if (ODD1) {
                const float nx = lane_next(rx), ny = lane_next(ry);
                o = down4_tail(rx, ry, nx, ny);
            } else {
                const float py = lane_prev(ry), nx = lane_next(rx);
                o = down4_tail(py, rx, ry, nx);
            }
            const int Y = (T - 2) >> 1;
            if (T >= 2 * A + 2 && Y <= B && st2_ok) d2[(size_t)(Y - p.loy2) * p.ws2] = o;
            s1 = res;
        }
    }
}

// Agent-coherent accesses (global_load / global_store ... sc1, no cache maintenance): what one workgroup of a launch wrote is what
// another workgroup of the SAME launch on another XCD reads (the XCDs' L2s are not coherent with each other for plain accesses
// inside a launch).  Used by ll_mid for the three small levels its producer blocks hand to its consumer blocks.
__device__ __forceinline__ float ld_f(const float *p, bool coh) {   // `coh` is a constant after unrolling wherever this is called
    if (coh) return __hip_atomic_load(const_cast<float *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *p;
}
template<bool COH>
__device__ __forceinline__ void st_f(float *p, float v) {
    if (COH) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}

// ---------------------------------------------------------------------------------------------------
// upsample(f)(X,Y) (:276-282) of a stored level plane `f` (origin lox/loy, row stride ws)
// The four taps of the bilinear footprint and the lerps on them are separate steps so that a caller can REQUEST the taps of many
// values before it combines the first (ll_up_multi); up_at = both steps, the same operations in the same order.
struct UpTaps {
    float aa, ab, ba, bb;    // f(ya, xa), f(ya, xb), f(yb, xa), f(yb, xb)
};
__device__ __forceinline__ UpTaps up_taps(const float *__restrict__ f, int lox, int loy, int ws, int X, int Y) {
    const int xa = dev::fdiv2(X + 1) - lox, xb = dev::fdiv2(X - 1) - lox;
    const int ya = dev::fdiv2(Y + 1) - loy, yb = dev::fdiv2(Y - 1) - loy;
    UpTaps t;
    t.aa = f[(size_t)ya * ws + xa], t.ab = f[(size_t)ya * ws + xb];
    t.ba = f[(size_t)yb * ws + xa], t.bb = f[(size_t)yb * ws + xb];
    return t;
}
__device__ __forceinline__ UpTaps up_taps_c(const float *__restrict__ f, int lox, int loy, int ws, int X, int Y, bool coh) {
    if (!coh) return up_taps(f, lox, loy, ws, X, Y);
    const int xa = dev::fdiv2(X + 1) - lox, xb = dev::fdiv2(X - 1) - lox;
    const int ya = dev::fdiv2(Y + 1) - loy, yb = dev::fdiv2(Y - 1) - loy;
    UpTaps t;
    t.aa = ld_f(f + ((size_t)ya * ws + xa), true), t.ab = ld_f(f + ((size_t)ya * ws + xb), true);
    t.ba = ld_f(f + ((size_t)yb * ws + xa), true), t.bb = ld_f(f + ((size_t)yb * ws + xb), true);
    return t;
}
__device__ __forceinline__ float up_from(const UpTaps &t, int X, int Y) {
    const float wx = (float)(dev::fmod2(X) * 2 + 1) * 0.25f, wy = (float)(dev::fmod2(Y) * 2 + 1) * 0.25f;
    const float ua = dev::lerpf(t.aa, t.ab, wx);
    const float ub = dev::lerpf(t.ba, t.bb, wx);
    return dev::lerpf(ua, ub, wy);
}
__device__ __forceinline__ float up_at(const float *__restrict__ f, int lox, int loy, int ws, int X, int Y) {
    return up_from(up_taps(f, lox, loy, ws, X, Y), X, Y);
}

// outGPyramid[J-1] = outLPyramid[J-1] (:76, :63-72 with lPyramid[J-1] = gPyramid[J-1], :51); o = element offset
__device__ __forceinline__ float top_value(const float *__restrict__ g, size_t ps, size_t o, int K, float Km1) {
    float level = g[(size_t)K * ps + o] * Km1;
    int li = dev::clampi((int)level, 0, K - 2);
    float lf = level - (float)li;
    return dev::mad2(1.0f - lf, g[(size_t)li * ps + o], lf, g[(size_t)(li + 1) * ps + o]);
}
// outLPyramid[j](X,Y), 0 < j < J-1 (:50-54, :63-72): g = level j (origin lox/loy), gc = level j+1
// SEL: level j was stored by ll_down01e — plane 0 = gPyramid[j](., ., li), plane 1 = gPyramid[j](., ., li + 1) for the pixel's own
// li (the only two values of level j this function reads), plane K = inGPyramid[j]
template<bool SEL = false>
__device__ __forceinline__ float outl_value(const float *__restrict__ g, int ws, size_t ps, int lox, int loy,
                                            const float *__restrict__ gc, int cws, size_t cps, int clox, int cloy,
                                            int X, int Y, int K, float Km1) {
    size_t o = (size_t)(Y - loy) * ws + (X - lox);
    float level = g[(size_t)K * ps + o] * Km1;
    int li = dev::clampi((int)level, 0, K - 2);
    float lf = level - (float)li;
    float l0 = g[(SEL ? 0 : (size_t)li * ps) + o] - up_at(gc + (size_t)li * cps, clox, cloy, cws, X, Y);
    float l1 = g[(SEL ? ps : (size_t)(li + 1) * ps) + o] - up_at(gc + (size_t)(li + 1) * cps, clox, cloy, cws, X, Y);
    return dev::mad2(1.0f - lf, l0, lf, l1);
}

__global__ void ll_top(const float *__restrict__ g, int ws, size_t ps, int lox, int loy, int rx0, int ry0, int rw,
                       int rh, int K, float Km1, float *__restrict__ out) {
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= rw || y >= rh) return;
    size_t o = (size_t)(ry0 + y - loy) * ws + (rx0 + x - lox);
    out[o] = top_value(g, ps, o, K, Km1);
}

// outGPyramid[j] = upsample(outGPyramid[j+1]) + outLPyramid[j], 1 <= j <= J-2 (:50-54, :63-79)
struct UpArgs {
    const float *g;          // level j
    int ws;
    size_t ps;
    int lox, loy;
    const float *gc, *outc;  // level j+1: gPyramid planes, outGPyramid
    int cws;
    size_t cps;
    int clox, cloy, rx0, ry0, rw, rh, K;
    float Km1;
    float *out;              // outGPyramid[j]
};
template<bool SEL>
__device__ __forceinline__ void up_pixel(const UpArgs &a, int x, int y) {   // (x, y) relative to R_j
    if (x >= a.rw || y >= a.rh) return;
    int X = a.rx0 + x, Y = a.ry0 + y;
    size_t o = (size_t)(Y - a.loy) * a.ws + (X - a.lox);
    float outL = outl_value<SEL>(a.g, a.ws, a.ps, a.lox, a.loy, a.gc, a.cws, a.cps, a.clox, a.cloy, X, Y, a.K, a.Km1);
    a.out[o] = up_at(a.outc, a.clox, a.cloy, a.cws, X, Y) + outL;
}
template<bool SEL = false>
__global__ __launch_bounds__(256) void ll_up(UpArgs a) {
    up_pixel<SEL>(a, blockIdx.x * blockDim.x + threadIdx.x, blockIdx.y);
}

// ---------------------------------------------------------------------------------------------------
// The coarse end of the pyramid (levels >= 5: at most a few thousand pixels per plane) is latency, not
// bandwidth: every launch of the chain costs ~4 us of dispatch + dependent-load latency however little it
// computes.  The two kernels below replace 2 (J-1-S) launches by 2: values of intermediate levels that a thread
// needs are RECOMPUTED in registers with exactly the operation order of the single-level kernels (so the
// results are bit-identical) instead of being exchanged through memory between launches.
struct DevLevel {
    float *g, *out;
    int lox, loy, w, h, ws;  // storage box (see Level)
    unsigned ps;
    int rx0, ry0, rw, rh;    // R_j
};
struct CoarseArgs {
    DevLevel lv[J];          // lv[0] = level S (stored), lv[d] = level S+d
    int K;
    float Km1;
};

// ---- ll_down_multi<DEPTH>: levels S+1 .. S+DEPTH of one plane from level S.
// A workgroup owns a TxT tile of the DEEPEST level and everything of the levels in between whose repeated
// floor(x/2) lands in that tile.  Window of level d (all coordinates absolute, clamped to the level's storage box,
// exactly as a read of that level is): [clamp(2 lo_{d+1} - 1), clamp(2 hi_{d+1} + 2)].  Level S+1's window is
// computed from global memory (16 clamped loads per value), the deeper ones from the previous window in LDS;
// window values a workgroup does not own are recomputed by the neighbours (identical operations, identical bits).
#ifndef HLMI_LL_DM_T
#define HLMI_LL_DM_T 4
#endif
constexpr int DM_T = HLMI_LL_DM_T;                           // tile edge at the deepest level (A/B: make VARIANT=_dm8 EXTRA=-DHLMI_LL_DM_T=8)
__host__ __device__ constexpr int dm_win(int depth_below) {  // window edge `depth_below` levels above the deepest
    int w = DM_T;
    for (int i = 0; i < depth_below; i++) w = 2 * w + 2;
    return w;
}
struct Range2 { int x0, x1, y0, y1; };  // inclusive, absolute
__device__ __forceinline__ Range2 clamp_to_box(const DevLevel &L, int x0, int x1, int y0, int y1) {
    Range2 r;
    r.x0 = dev::clampi(x0, L.lox, L.lox + L.w - 1), r.x1 = dev::clampi(x1, L.lox, L.lox + L.w - 1);
    r.y0 = dev::clampi(y0, L.loy, L.loy + L.h - 1), r.y1 = dev::clampi(y1, L.loy, L.loy + L.h - 1);
    return r;
}
template<int DEPTH, bool COH = false>   // COH: the levels it makes are stored agent-coherently (ll_mid)
__device__ __forceinline__ void down_multi_tile(const CoarseArgs &a, int ntx, int nty, int b) {   // b = tile x + ntx (tile y + nty plane)
    constexpr int W1 = dm_win(DEPTH - 1), W2 = DEPTH >= 2 ? dm_win(DEPTH - 2) : 1, W3 = DEPTH >= 3 ? dm_win(DEPTH - 3) : 1,
                  W4 = DEPTH >= 4 ? dm_win(DEPTH - 4) : 1;
    __shared__ float tile1[W1 * W1], tile2[W2 * W2], tile3[W3 * W3], tile4[W4 * W4];  // windows of levels S+1 .. S+4
    float *const tiles[5] = {nullptr, tile1, tile2, tile3, tile4};
    const int ws_[5] = {0, W1, W2, W3, W4};
    const int tx = b % ntx, ty = (b / ntx) % nty, plane = b / (ntx * nty);
    // windows (win[d]) and owned ranges (own[d]) of levels S+d, deepest first
    Range2 win[5], own[5];
    {
        const DevLevel &T = a.lv[DEPTH];
        win[DEPTH] = own[DEPTH] = clamp_to_box(T, T.lox + tx * DM_T, T.lox + tx * DM_T + DM_T - 1, T.loy + ty * DM_T,
                                               T.loy + ty * DM_T + DM_T - 1);
#pragma unroll
        for (int d = DEPTH - 1; d >= 1; d--) {
            const DevLevel &L = a.lv[d];
            win[d] = clamp_to_box(L, 2 * win[d + 1].x0 - 1, 2 * win[d + 1].x1 + 2, 2 * win[d + 1].y0 - 1, 2 * win[d + 1].y1 + 2);
            own[d] = clamp_to_box(L, 2 * own[d + 1].x0, 2 * own[d + 1].x1 + 1, 2 * own[d + 1].y0, 2 * own[d + 1].y1 + 1);
        }
    }
#pragma unroll
    for (int d = 1; d <= DEPTH; d++) {
        const DevLevel &L = a.lv[d], &Sx = a.lv[d - 1];
        const Range2 w = win[d], o = own[d];
        const int nx = w.x1 - w.x0 + 1, n = nx * (w.y1 - w.y0 + 1);
        float *const dst = tiles[d];
        const float *const src_g = Sx.g + (size_t)plane * Sx.ps;
        const float *const src_t = tiles[d - 1];
        const Range2 pw = win[d == 1 ? 1 : d - 1];
        const int pnx = ws_[d == 1 ? 1 : d - 1];
        // two elements per thread at a time: the first level's 16 taps per element come from memory, and both elements' taps are
        // requested before either is summed (one memory round trip for a 22 x 22 window instead of two)
        for (int e0 = threadIdx.x; e0 < n; e0 += 512) {
            float r[2][4][4];
            int X[2], Y[2], yy[2];
            bool ok[2];
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int e = e0 + 256 * u, ec = min(e, n - 1);
                ok[u] = e < n;
                yy[u] = ec / nx, X[u] = w.x0 + (ec - yy[u] * nx), Y[u] = w.y0 + yy[u];
#pragma unroll
                for (int i = 0; i < 4; i++) {
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const int qx = dev::clampi(2 * X[u] - 1 + i, Sx.lox, Sx.lox + Sx.w - 1);
                        const int qy = dev::clampi(2 * Y[u] - 1 + k, Sx.loy, Sx.loy + Sx.h - 1);
                        r[u][i][k] = (d == 1) ? src_g[(size_t)(qy - Sx.loy) * Sx.ws + (qx - Sx.lox)]
                                              : src_t[(qy - pw.y0) * pnx + (qx - pw.x0)];
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < 2; u++) {
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; i++) v[i] = down4_raw(r[u][i][0], r[u][i][1], r[u][i][2], r[u][i][3]);   // vertical pass first (downy, :270), then horizontal (downx, :271)
                const float val = down4_tail(v[0], v[1], v[2], v[3]);
                if (ok[u]) {
                    dst[yy[u] * ws_[d] + (X[u] - w.x0)] = val;
                    if (X[u] >= o.x0 && X[u] <= o.x1 && Y[u] >= o.y0 && Y[u] <= o.y1) {
                        st_f<COH>(L.g + ((size_t)plane * L.ps + (size_t)(Y[u] - L.loy) * L.ws + (X[u] - L.lox)), val);
                    }
                }
            }
        }
        if (d < DEPTH) __syncthreads();
    }
}
template<int DEPTH>
__global__ __launch_bounds__(256) void ll_down_multi(CoarseArgs a, int ntx, int nty) {
    down_multi_tile<DEPTH>(a, ntx, nty, (int)blockIdx.x);
}

// ---- ll_up_multi<TOP>: outGPyramid[S] on R_S from gPyramid / inGPyramid of levels S .. S+TOP (= J-1).
// A workgroup owns a 16x16 tile of R_S.  Everything that needs memory — outLPyramid of every level over the
// region the tile depends on (the top level's outGPyramid is its outLPyramid, :76) — is independent of the
// coarse-to-fine recursion, so it is computed first, by all threads at once, into LDS; the recursion
// outG_j = upsample(outG_{j+1}) + outL_j (:77-79) then runs through LDS, one barrier per level.
constexpr int UM_T = 16;
__host__ __device__ constexpr int um_win(int d) {  // edge of the region of level S+d a tile depends on
    int w = UM_T;
    for (int i = 0; i < d; i++) w = w / 2 + 2;
    return w;
}
__host__ __device__ constexpr int um_off(int d) {  // offset of level S+d's region in the workgroup's LDS array
    int o = 0;
    for (int i = 0; i < d; i++) o += um_win(i) * um_win(i);
    return o;
}
// up_multi_tile handles ONE element per thread and level (act[d] = tid < n): every region of a tile must fit 256 threads
static_assert(UM_T * UM_T <= 256 && um_win(1) * um_win(1) <= 256, "ll_up_multi: a level's region of a tile exceeds the workgroup");
struct NoWait { __device__ void operator()() const {} };
// CF: levels S+CF .. were made by producer blocks of the SAME launch (ll_mid): agent-coherent loads, requested only after `wait()`
// returns (everything that depends on the older levels alone is requested BEFORE it: those round trips run under the wait)
template<int TOP, int CF = TOP + 1, class Wait = NoWait>
__device__ __forceinline__ void up_multi_tile(const CoarseArgs &a, int ntx, int b, Wait wait = Wait()) {
    __shared__ float tl[um_off(TOP + 1)];
    const int tx = b % ntx, ty = b / ntx;
    Range2 reg[TOP + 1];
    {
        const DevLevel &L = a.lv[0];
        reg[0].x0 = L.rx0 + tx * UM_T, reg[0].x1 = min(reg[0].x0 + UM_T - 1, L.rx0 + L.rw - 1);
        reg[0].y0 = L.ry0 + ty * UM_T, reg[0].y1 = min(reg[0].y0 + UM_T - 1, L.ry0 + L.rh - 1);
#pragma unroll
        for (int d = 1; d <= TOP; d++) {
            reg[d].x0 = dev::fdiv2(reg[d - 1].x0 - 1), reg[d].x1 = dev::fdiv2(reg[d - 1].x1 + 1);
            reg[d].y0 = dev::fdiv2(reg[d - 1].y0 - 1), reg[d].y1 = dev::fdiv2(reg[d - 1].y1 + 1);
        }
    }
    // 1. outLPyramid of every level (memory-bound part, all of it independent of the recursion).  A level's region has at most
    //    UM_T x UM_T = 256 elements, one per thread; a value needs its level's inGPyramid first (which planes?) and then ten
    //    gathers — two dependent round trips.  Round 5: THREE passes over all levels — request every inGPyramid value, then every
    //    gather, then combine — so that the tile pays the two round trips once, not once per level (the per-level loops this
    //    replaces ran them level after level: ten dependent trips).  outl_value's / top_value's operations in their order.
    int eX[TOP + 1], eY[TOP + 1];
    bool act[TOP + 1];
    size_t eo[TOP + 1];
    float lvl[TOP + 1];
#pragma unroll
    for (int d = 0; d <= TOP; d++) {
        const DevLevel &L = a.lv[d];
        const Range2 r = reg[d];
        const int nx = r.x1 - r.x0 + 1, n = nx * (r.y1 - r.y0 + 1);
        act[d] = (int)threadIdx.x < n;
        const int e = act[d] ? (int)threadIdx.x : 0, yy = e / nx;   // idle threads re-read element 0 and store nothing
        eX[d] = r.x0 + (e - yy * nx), eY[d] = r.y0 + yy;
        eo[d] = (size_t)(eY[d] - L.loy) * L.ws + (eX[d] - L.lox);
        if (d < CF) lvl[d] = L.g[(size_t)a.K * L.ps + eo[d]];
    }
    __builtin_amdgcn_sched_barrier(0);   // (left alone the scheduler starts a level's gathers as soon as its first load is back)
    float lf[TOP + 1], g0[TOP + 1], g1[TOP + 1];
    int lis[TOP + 1];
    UpTaps t0[TOP + 1], t1[TOP + 1];
    auto planes_of = [&](int d) {
        const float level = lvl[d] * a.Km1;
        lis[d] = dev::clampi((int)level, 0, a.K - 2);
        lf[d] = level - (float)lis[d];
    };
#pragma unroll
    for (int d = 0; d <= TOP; d++) {   // gathers that touch the older levels only
        if (d < CF) {
            const DevLevel &L = a.lv[d];
            planes_of(d);
            const int li = lis[d];
            g0[d] = L.g[(size_t)li * L.ps + eo[d]], g1[d] = L.g[(size_t)(li + 1) * L.ps + eo[d]];
            if (d < TOP && d + 1 < CF) {
                const DevLevel &C = a.lv[d + 1];
                t0[d] = up_taps(C.g + (size_t)li * C.ps, C.lox, C.loy, C.ws, eX[d], eY[d]);
                t1[d] = up_taps(C.g + (size_t)(li + 1) * C.ps, C.lox, C.loy, C.ws, eX[d], eY[d]);
            }
        }
    }
    if (CF <= TOP) {
        __builtin_amdgcn_sched_barrier(0);
        wait();
#pragma unroll
        for (int d = 0; d <= TOP; d++) {
            if (d >= CF) lvl[d] = ld_f(a.lv[d].g + ((size_t)a.K * a.lv[d].ps + eo[d]), true);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int d = 0; d <= TOP; d++) {
            const DevLevel &L = a.lv[d];
            if (d >= CF) {
                planes_of(d);
                g0[d] = ld_f(L.g + ((size_t)lis[d] * L.ps + eo[d]), true), g1[d] = ld_f(L.g + ((size_t)(lis[d] + 1) * L.ps + eo[d]), true);
            }
            if (d < TOP && d + 1 >= CF) {
                const DevLevel &C = a.lv[d + 1];
                t0[d] = up_taps_c(C.g + (size_t)lis[d] * C.ps, C.lox, C.loy, C.ws, eX[d], eY[d], true);
                t1[d] = up_taps_c(C.g + (size_t)(lis[d] + 1) * C.ps, C.lox, C.loy, C.ws, eX[d], eY[d], true);
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int d = 0; d <= TOP; d++) {
        float v;
        if (d == TOP) {
            v = dev::mad2(1.0f - lf[d], g0[d], lf[d], g1[d]);
        } else {
            const float l0 = g0[d] - up_from(t0[d], eX[d], eY[d]);
            const float l1 = g1[d] - up_from(t1[d], eX[d], eY[d]);
            v = dev::mad2(1.0f - lf[d], l0, lf[d], l1);
        }
        if (act[d]) (tl + um_off(d))[(eY[d] - reg[d].y0) * um_win(d) + (eX[d] - reg[d].x0)] = v;
    }
    __syncthreads();
    // 2. collapse: level S+d from level S+d+1 (both in LDS), operation order of up_at
#pragma unroll
    for (int d = TOP - 1; d >= 0; d--) {
        const Range2 r = reg[d], c = reg[d + 1];
        const float *const ct = tl + um_off(d + 1);
        float *const tile = tl + um_off(d);
        const int cw = um_win(d + 1), tw = um_win(d);
        const int nx = r.x1 - r.x0 + 1, n = nx * (r.y1 - r.y0 + 1);
        for (int e = threadIdx.x; e < n; e += 256) {
            const int yy = e / nx, X = r.x0 + (e - yy * nx), Y = r.y0 + yy;
            const int xa = dev::fdiv2(X + 1) - c.x0, xb = dev::fdiv2(X - 1) - c.x0;
            const int ya = dev::fdiv2(Y + 1) - c.y0, yb = dev::fdiv2(Y - 1) - c.y0;
            const float wx = (float)(dev::fmod2(X) * 2 + 1) * 0.25f, wy = (float)(dev::fmod2(Y) * 2 + 1) * 0.25f;
            const float ua = dev::lerpf(ct[ya * cw + xa], ct[ya * cw + xb], wx);
            const float ub = dev::lerpf(ct[yb * cw + xa], ct[yb * cw + xb], wx);
            const float v = dev::lerpf(ua, ub, wy) + tile[yy * tw + (X - r.x0)];
            if (d == 0) {
                const DevLevel &L = a.lv[0];
                L.out[(size_t)(Y - L.loy) * L.ws + (X - L.lox)] = v;
            } else {
                tile[yy * tw + (X - r.x0)] = v;
            }
        }
        if (d > 0) __syncthreads();
    }
}
template<int TOP>
__global__ __launch_bounds__(256) void ll_up_multi(CoarseArgs a, int ntx) {
    up_multi_tile<TOP>(a, ntx, (int)blockIdx.x);
}

// ---- ll_mid<DEPTH, TOP, CF>: ll_down_multi AND ll_up_multi as ONE launch (round 6).  With four frames in flight each of the three
// short launches of the chain costs the frame ~3.2 us whatever it does (profiles/r06_launch_cost_skip_ab.txt: they do not overlap each
// other across queues), so the last dependency that is small enough is carried inside a launch: blocks [0, nD) are ll_down_multi's
// tiles — levels S+1 .. of every plane, stored agent-coherently (write-through) —, each bumps `ctr` once behind its stores; blocks
// [nD, ..) are ll_up_multi's tiles, which wait until the count is full and read those levels agent-coherently.  Blocks are dispatched in
// order, so a resident consumer implies every producer has been dispatched, and producers wait for nothing: no deadlock however few
// workgroups are resident.  What round 5's ll_coarse (the WHOLE chain as tickets) paid — thousands of same-address atomics, megabytes
// through 4-byte coherent accesses — is here nD (486) atomics on one word and the ~100 K values of levels 5-7.  `ctr` is zeroed by
// ll_down_strip2 of the same call.  Same device functions as the two launches: same operations, same bits.
template<int DEPTH, int TOP, int CF>
__global__ __launch_bounds__(256) void ll_mid(CoarseArgs ad, int ntxd, int ntyd, int nD, CoarseArgs au, int ntxu, unsigned *ctr) {
    const int b = (int)blockIdx.x;
    __shared__ int s_last;
    if (b < nD) {
        down_multi_tile<DEPTH, true>(ad, ntxd, ntyd, b);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this thread's write-through stores have been acknowledged
        __syncthreads();
        if (threadIdx.x == 0) {   // count per plane first (its own line), the last block of a plane counts the plane
            const int per_plane = ntxd * ntyd, plane = b / per_plane;
            bool last = __hip_atomic_fetch_add(ctr + MID_LINE * (1 + MID_FLAGS + plane), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)per_plane - 1u;
            if (last) last = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(nD / per_plane) - 1u;
            s_last = last;
        }
        __syncthreads();
        // the last producer raises the flags: consumers never touch the counter's line (same-address operations are served one
        // per ~11 ns, profiles/r05_sync_cost.txt — 510 pollers on the counter itself delayed the producers' own increments by 16 us)
        if (s_last && threadIdx.x < MID_FLAGS) __hip_atomic_store(ctr + MID_LINE * (1 + (int)threadIdx.x), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        const unsigned *flag = ctr + MID_LINE * (1 + (b & (MID_FLAGS - 1)));
        auto wait = [flag]() {
            if (threadIdx.x == 0) {
                while (__hip_atomic_load(const_cast<unsigned *>(flag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) __builtin_amdgcn_s_sleep(8);
            }
            __syncthreads();
        };
        up_multi_tile<TOP, CF>(au, ntxu, b - nD, wait);
    }
}

// Measured and not kept (round 5, git 5e5e508, profiles/r05_ll_coarse_*.txt, profiles/NOTES.md): the five launches between the two big
// kernels as ONE launch (`ll_coarse`: the stages' work items as a ticket queue in stage order — deadlock-free for any number of resident
// workgroups — running these same device functions; bit-exact on the whole suite).  With agent-scope release / acquire fences around
// plain accesses every item wrote back and invalidated a whole L2: 300-600 us for the launch; with agent-coherent (sc1) loads and
// stores and no cache maintenance 125-270 us, growing with the number of workgroups — against 43 us for the five launches: the
// command processor hands a dependent launch to the CUs (~4.5 us) faster than ~2000 same-address atomics and memory-side round trips do.

// ---------------------------------------------------------------------------------------------------
// level 0: outGPyramid[0], colour, u16 (:63-87).  gray / gPyramid[0] recomputed pointwise.
// Workgroup = 4 waves as 2 x 2; a wave covers 128 columns x RU rows, a lane 2 adjacent columns.
struct Up0Args {
    const uint16_t *in;
    long in_sy;
    long gco[3];             // clamped channel offsets feeding `gray`
    long cco[3];             // channel offsets of the (unclamped) colour read, per output channel
    const float *lut_g, *g1, *out1;
    int lox1, loy1, ws1;
    size_t ps1;
    uint16_t *out;
    long out_sy, out_sc;
    int ox0, oy0, ow, oh, nc, RU;
    int same_ch;             // the colour channels are exactly the gray channels: load once
    float beta;
    // level 2, for the fused collapse of level 1 (ll_up0f<.., .., true>): outGPyramid[1] is then produced per workgroup
    // tile in LDS instead of by an ll_up:1 launch
    const float *g2, *out2;
    int lox2, loy2, ws2;
    size_t ps2;
    int rx1_1;               // right end of R_1 (the tiles of the last workgroup column stop there)
    int rx0_1, ry0_1, ry1_1; // the rest of R_1
};

template<bool VEC, bool LUT_LDS>
__global__ __launch_bounds__(256) void ll_up0(Up0Args p, Geometry gm) {
    extern __shared__ float slut[];
    if (LUT_LDS) {
        for (int i = threadIdx.x; i <= 2 * gm.half; i += 256) slut[i] = p.lut_g[i];
        __syncthreads();
    }
    const float *lut = LUT_LDS ? slut : p.lut_g;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int x = blockIdx.x * 256 + (wave & 1) * 128 + 2 * lane;  // output storage column of the lane's pair
    const int y0 = blockIdx.y * (2 * p.RU) + (wave >> 1) * p.RU;
    if (x >= p.ow || y0 >= p.oh) return;
    const int npx = min(2, p.ow - x);
    const int y1 = min(y0 + p.RU, p.oh);
    const int X = p.ox0 + x;
    const bool vec = VEC && npx == 2;
    for (int y = y0; y < y1; y++) {
        const int Y = p.oy0 + y;
        const uint16_t *ip = p.in + (long)(Y - gm.iy0) * p.in_sy + (X - gm.ix0);
        uint16_t *op = p.out + (long)y * p.out_sy + x;
        uint16_t gch[3][2], cch[3][2];
        if (vec) {
#pragma unroll
            for (int c = 0; c < 3; c++) {
                ushort2 v = *reinterpret_cast<const ushort2 *>(ip + p.gco[c]);
                gch[c][0] = v.x, gch[c][1] = v.y;
            }
        } else {
#pragma unroll
            for (int c = 0; c < 3; c++) {
                gch[c][0] = ip[p.gco[c]];
                gch[c][1] = npx == 2 ? ip[p.gco[c] + 1] : gch[c][0];
            }
        }
        if (p.same_ch) {
#pragma unroll
            for (int c = 0; c < 3; c++) cch[c][0] = gch[c][0], cch[c][1] = gch[c][1];
        } else {
#pragma unroll
            for (int c = 0; c < 3; c++) {
                if (c < p.nc) {
                    cch[c][0] = ip[p.cco[c]];
                    cch[c][1] = npx == 2 ? ip[p.cco[c] + 1] : cch[c][0];
                } else {
                    cch[c][0] = cch[c][1] = 0;
                }
            }
        }
        uint16_t res[3][2];
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int Xi = X + i;
            float gray = gray_from(gch[0][i], gch[1][i], gch[2][i]);
            float level = gray * gm.Km1;
            int li = dev::clampi((int)level, 0, gm.K - 2);
            float lf = level - (float)li;
            int idx = idx_of(gray, gm.Km1, gm.half);
            const float *lp = lut + (idx - 256 * li + gm.half);
            float lev0 = (float)li * gm.inv_Km1, lev1 = (float)(li + 1) * gm.inv_Km1;
            const float *gp = p.g1 + (size_t)li * p.ps1;
            float l0 = g0_val(gray, lev0, p.beta, lp[0]) - up_at(gp, p.lox1, p.loy1, p.ws1, Xi, Y);
            float l1 = g0_val(gray, lev1, p.beta, lp[-256]) - up_at(gp + p.ps1, p.lox1, p.loy1, p.ws1, Xi, Y);
            float outL = dev::mad2(1.0f - lf, l0, lf, l1);
            float og = (up_at(p.out1, p.lox1, p.loy1, p.ws1, Xi, Y) + outL) + 0.01f;
            float gr = gray + 0.01f;
#pragma unroll
            for (int c = 0; c < 3; c++) {
                // color = input * (outG0 + eps) / (gray + eps); input is the UNclamped input here (:84)
                float v = ((float)cch[c][i] * og) / gr;
                res[c][i] = (uint16_t)dev::clampf(v, 0.0f, 65535.0f);
            }
        }
        if (vec) {
#pragma unroll
            for (int c = 0; c < 3; c++) {
                if (c < p.nc) {
                    *reinterpret_cast<ushort2 *>(op + (long)c * p.out_sc) = make_ushort2(res[c][0], res[c][1]);
                }
            }
        } else {
#pragma unroll
            for (int c = 0; c < 3; c++) {
                if (c < p.nc) {
                    op[(long)c * p.out_sc] = res[c][0];
                    if (npx == 2) op[(long)c * p.out_sc + 1] = res[c][1];
                }
            }
        }
    }
}

// ---- ll_up0f: the same function as ll_up0<true, *> for the common geometry (even output origin and width,
// three colour channels that are the gray channels, workspace offsets < 4 GB), restructured around what the
// hardware charges for: instruction issue and gather instructions.
//   * a wave's row Y is wave-uniform: row pointers, vertical weights and row parity live in scalar registers
//   * every load is `scalar base + 32-bit lane offset` (no 64-bit vector address arithmetic)
//   * the pair (X even, X+1) of a lane needs coarse columns c-1, c, c+1 (c = X/2): one 8-byte gather per
//     (pixel, plane, coarse row) and one 12-byte gather per coarse row of outGPyramid[1] for the pair
//   * lerp weights are the constants 1/4, 3/4 (:276-282 with x, y parity known), (1 - w) is exact
//   * the three colour quotients share their denominator: one v_rcp_f32 + one Newton step, then the
//     correctly-rounded refinement per numerator (div3_by below)
struct __attribute__((packed, aligned(4))) F2U { float x, y; };
struct __attribute__((packed, aligned(4))) F3U { float x, y, z; };
template<typename T>
__device__ __forceinline__ T ld_su(const void *sbase, uint32_t byte_off) {  // sbase wave-uniform
    return *reinterpret_cast<const T *>(reinterpret_cast<const char *>(sbase) + byte_off);
}

// q[i] = n[i] / d, correctly rounded (identical to IEEE `/` up to the sign of a zero quotient, which the u16 cast
// of the caller discards), for 0.005 <= d <= 4 and n[i] == 0 or
// 2^-100 < |n[i]| < 2^100.  This is the refinement the compiler emits for `/` (rcp, Newton step, quotient, two
// residual corrections) without v_div_scale / v_div_fmas' post-scale / v_div_fixup, which only act on operands
// outside that range (denormal or huge quotients, infinities, NaNs); tests/test_local_laplacian.py checks it
// against `/` on 2^26 random operand pairs of the range (hlmi_debug_div3_check).
__device__ __forceinline__ void div3_by(const float (&n)[3], float d, float (&q)[3]) {
    float r = __builtin_amdgcn_rcpf(d);
    float e = __builtin_fmaf(-d, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
#pragma unroll
    for (int i = 0; i < 3; i++) {
        float qq = n[i] * r;
        float t = __builtin_fmaf(-d, qq, n[i]);
        qq = __builtin_fmaf(t, r, qq);
        t = __builtin_fmaf(-d, qq, n[i]);
        q[i] = __builtin_fmaf(t, r, qq);
    }
}
__global__ void ll_div3_check(const float *n, const float *d, int count, int *bad) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    float nn[3] = {n[i], -n[i], n[i] * 3.0f}, q[3];
    div3_by(nn, d[i], q);
#pragma unroll
    for (int k = 0; k < 3; k++) {
        float ref = nn[k] / d[i];  // a zero quotient may differ in sign (the u16 cast discards it): `==` accepts that
        if (!(ref == q[k]) && !(ref != ref && q[k] != q[k])) atomicAdd(bad, 1);
    }
}

//   * FUSE1: outGPyramid[1] on the part of R_1 this workgroup's 256 x 2 RU outputs read (130 coarse columns x up to
//     RU + 2 coarse rows) is computed HERE, into LDS, by the very device functions ll_up uses (bit-identical), instead
//     of by an ll_up:1 launch that writes it to memory for this kernel to gather back: one launch, the write and the
//     re-read of the plane, and a second pass over level 1's planes less.  The tile halo costs (130 x (RU + 2)) /
//     (128 x RU) - 1 recomputed values (16 % at RU = 16).
constexpr int U0_TW = 130, U0_TS = 131;  // coarse columns of a workgroup's tile / its LDS row stride
template<bool LUT_LDS, bool B1, bool FUSE1>
__global__ __launch_bounds__(256) void ll_up0f(Up0Args p, Geometry gm) {
    extern __shared__ float slut[];
    LL_PROBE_T(pt0);
    float *s_out1 = slut + (LUT_LDS ? ((2 * gm.half + 2) & ~1) : 0);
    if (LUT_LDS) {
        for (int i = threadIdx.x; i <= 2 * gm.half; i += 256) slut[i] = p.lut_g[i];
    }
    int cx0 = 0, cy0 = 0;
    if (FUSE1) {
        const int X0 = p.ox0 + (int)blockIdx.x * 256;                       // even
        const int Yw0 = p.oy0 + (int)blockIdx.y * (2 * p.RU);
        const int Yw1 = min(Yw0 + 2 * p.RU, p.oy0 + p.oh) - 1;
        cx0 = (X0 >> 1) - 1, cy0 = dev::fdiv2(Yw0 - 1);
        const int th = dev::fdiv2(Yw1 + 1) - cy0 + 1;
        for (int e = threadIdx.x; e < U0_TW * th; e += 256) {
            const int ty = e / U0_TW, tx = e - ty * U0_TW;
            const int cx = cx0 + tx, cy = cy0 + ty;
            if (cx > p.rx1_1) continue;
            // outGPyramid[1] = upsample(outGPyramid[2]) + outLPyramid[1]   (:76-79), exactly as ll_up computes it
            const float outL = outl_value(p.g1, p.ws1, p.ps1, p.lox1, p.loy1, p.g2, p.ws2, p.ps2, p.lox2, p.loy2, cx, cy, gm.K, gm.Km1);
            s_out1[ty * U0_TS + tx] = up_at(p.out2, p.lox2, p.loy2, p.ws2, cx, cy) + outL;
        }
    }
    LL_PROBE_T(pt1);
    if (LUT_LDS || FUSE1) __syncthreads();
    LL_PROBE_T(pt2);
    const float *lut = LUT_LDS ? slut : p.lut_g;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int x = blockIdx.x * 256 + (wave & 1) * 128 + 2 * lane;  // output storage column of the lane's pair
    const int y0 = blockIdx.y * (2 * p.RU) + (wave >> 1) * p.RU;
    if (y0 >= p.oh || x >= p.ow) return;
    const int y1 = min(y0 + p.RU, p.oh);
    const int X = p.ox0 + x;                                              // even
    const uint32_t colb = (uint32_t)((X >> 1) - 1 - p.lox1) * 4u;         // coarse column c-1, bytes
    const uint32_t inb = (uint32_t)(X - gm.ix0) * 2u, outb = (uint32_t)x * 2u;
    const uint32_t psb = (uint32_t)p.ps1 * 4u;
    // lerp(zero, one, w) = zero*(1-w) + one*w with w in {1/4, 3/4}: the product by 1/4 is exact, so adding it with
    // an fma rounds exactly like the separate multiply and add of the definition (one instruction less per lerp)
    // fma canon: the contracted product is the one with `zero` (dev::mad2) — f[c] * 3/4 for even X, f[c+1] * 1/4 for odd X
    auto hl0 = [](float rm, float r0) {   // lerp(f[c], f[c-1], 1/4): X even
        return dev::CANON_FMA ? __builtin_fmaf(r0, 0.75f, rm * 0.25f) : __builtin_fmaf(rm, 0.25f, r0 * 0.75f);
    };
    auto hl1 = [](float r0, float rp) { return __builtin_fmaf(rp, 0.25f, r0 * 0.75f); };  // lerp(f[c+1], f[c], 3/4): X odd
    // The row loop is software-pipelined over three rows: a row needs two dependent round trips to memory (its pixels
    // -> which planes to gather from -> the gathers), ~2 us per row if taken one after the other, and a wave walks RU
    // rows — the launch used to last as long as that chain (52 % of all wave-cycles waiting, SQ_WAIT_ANY).  Now, while
    // row y is finished (stage 2), the gathers of row y+1 are in flight (issued by its stage 1) and the pixels of row
    // y+2 are being loaded.  Two register sets per stage, swapped by unrolling (never copied: copying the target of a
    // load in flight would wait for it); rows past the end are clamped (loaded and prepared again, never stored).
    struct Frame {
        ushort2 c0, c1, c2;
    };
    struct Prep {                 // what stage 2 needs of a row besides the gathers
        float chf[3][2];          // colour channels as float
        float gray[2], lf[2], lev0[2], lev1[2];
        float lut0[2], lut1[2];   // remap values of the two planes
        F3U OA, OB;               // outGPyramid[1] rows (q: weight 1/4, t: weight 3/4), columns c-1, c, c+1
        float wq, wt;             // fma canon: OA / A* are row ya (`zero`), OB / B* row yb (`one`); their weights
    };
    struct Gath {
        F2U A0[2], B0[2], A1[2], Bp[2];
    };
    auto load_frame = [&](int y, Frame &f) {
        const uint16_t *irow = p.in + (long)(p.oy0 + min(y, y1 - 1) - gm.iy0) * p.in_sy;
        f.c0 = ld_frame2(irow + p.gco[0], inb), f.c1 = ld_frame2(irow + p.gco[1], inb), f.c2 = ld_frame2(irow + p.gco[2], inb);
    };
    // canon 0: (uq, ut) = the rows weighted 1/4 and 3/4; fma canon: (uq, ut) = (`zero` = row ya, `one` = row yb) with the row's weights
    // wq = 1 - wy, wt = wy (wave-uniform): lerp(ua, ub, wy) = fma(ua, 1 - wy, ub * wy)
    auto vl = [](float uq, float ut, float wq, float wt) {
        return dev::CANON_FMA ? __builtin_fmaf(uq, wq, ut * wt) : __builtin_fmaf(uq, 0.25f, ut * 0.75f);
    };
    auto stage1 = [&](int yy, const Frame &f, Prep &s, Gath &g) {
        const int Y = p.oy0 + min(yy, y1 - 1);
        const int ya = dev::fdiv2(Y + 1) - p.loy1, yb = dev::fdiv2(Y - 1) - p.loy1;
        const bool yodd = dev::fmod2(Y) != 0;  // wave-uniform
        // lerp(ua, ub, wy) (:280), ua from coarse row ya, ub from yb: wy = 3/4 for odd Y, 1/4 for even Y.  The row
        // whose weight is 1/4 (an exact product) is called q, the other t — a scalar choice of row pointers.
        const int yq = (dev::CANON_FMA || yodd) ? ya : yb, yt = (dev::CANON_FMA || yodd) ? yb : ya;
        s.wq = yodd ? 0.25f : 0.75f, s.wt = yodd ? 0.75f : 0.25f;
        const float *ga = p.g1 + (size_t)yq * p.ws1, *gb = p.g1 + (size_t)yt * p.ws1;
        const uint16_t ch[3][2] = {{f.c0.x, f.c0.y}, {f.c1.x, f.c1.y}, {f.c2.x, f.c2.y}};
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const float gray = gray_from(ch[0][i], ch[1][i], ch[2][i]);
            const float level = gray * gm.Km1;
            // gray >= 0: the lower bounds of clamp(.., 0, ..) (:43, :66) can never bind
            const int li = min((int)level, gm.K - 2);
            const float lif = (float)li;
            const int idx = min((int)(level * 256.0f), gm.half);
            const float *lp = lut + (idx - 256 * li + gm.half);
            const uint32_t pb = (uint32_t)li * psb + colb + 4u * i;
            g.A0[i] = ld_su<F2U>(ga, pb), g.B0[i] = ld_su<F2U>(gb, pb);
            g.A1[i] = ld_su<F2U>(ga, pb + psb), g.Bp[i] = ld_su<F2U>(gb, pb + psb);
            s.lut0[i] = lp[0], s.lut1[i] = lp[-256];
            s.gray[i] = gray, s.lf[i] = level - lif;
            s.lev0[i] = lif * gm.inv_Km1, s.lev1[i] = (lif + 1.0f) * gm.inv_Km1;
#pragma unroll
            for (int c = 0; c < 3; c++) s.chf[c][i] = (float)ch[c][i];
        }
        if (FUSE1) {  // rows of the workgroup's LDS tile; column c - 1 - cx0 = 64 (wave & 1) + lane
            const float *oa = s_out1 + (yq + p.loy1 - cy0) * U0_TS + (wave & 1) * 64 + lane;
            const float *ob = s_out1 + (yt + p.loy1 - cy0) * U0_TS + (wave & 1) * 64 + lane;
            s.OA.x = oa[0], s.OA.y = oa[1], s.OA.z = oa[2];
            s.OB.x = ob[0], s.OB.y = ob[1], s.OB.z = ob[2];
        } else {
            const float *oa = p.out1 + (size_t)yq * p.ws1, *ob = p.out1 + (size_t)yt * p.ws1;
            s.OA = ld_su<F3U>(oa, colb), s.OB = ld_su<F3U>(ob, colb);
        }
    };
    auto stage2 = [&](int y, const Prep &s, const Gath &g) {
        const float uo[2] = {vl(hl0(s.OA.x, s.OA.y), hl0(s.OB.x, s.OB.y), s.wq, s.wt), vl(hl1(s.OA.y, s.OA.z), hl1(s.OB.y, s.OB.z), s.wq, s.wt)};
        uint16_t res[3][2];
#pragma unroll
        for (int i = 0; i < 2; i++) {
            float u0, u1;
            if (i == 0) {
                u0 = vl(hl0(g.A0[i].x, g.A0[i].y), hl0(g.B0[i].x, g.B0[i].y), s.wq, s.wt), u1 = vl(hl0(g.A1[i].x, g.A1[i].y), hl0(g.Bp[i].x, g.Bp[i].y), s.wq, s.wt);
            } else {
                u0 = vl(hl1(g.A0[i].x, g.A0[i].y), hl1(g.B0[i].x, g.B0[i].y), s.wq, s.wt), u1 = vl(hl1(g.A1[i].x, g.A1[i].y), hl1(g.Bp[i].x, g.Bp[i].y), s.wq, s.wt);
            }
            const float l0 = g0_val<B1>(s.gray[i], s.lev0[i], p.beta, s.lut0[i]) - u0;
            const float l1 = g0_val<B1>(s.gray[i], s.lev1[i], p.beta, s.lut1[i]) - u1;
            const float outL = dev::mad2(1.0f - s.lf[i], l0, s.lf[i], l1);
            const float og = (uo[i] + outL) + 0.01f;
            const float gr = s.gray[i] + 0.01f;
            const float n[3] = {s.chf[0][i] * og, s.chf[1][i] * og, s.chf[2][i] * og};
            float q[3];
            div3_by(n, gr, q);
#pragma unroll
            for (int c = 0; c < 3; c++) res[c][i] = (uint16_t)__builtin_amdgcn_fmed3f(q[c], 0.0f, 65535.0f);  // q is never NaN
        }
        if (y < y1) {
            uint16_t *orow = p.out + (long)y * p.out_sy;
#pragma unroll
            for (int c = 0; c < 3; c++) st_frame2(reinterpret_cast<char *>(orow + (long)c * p.out_sc) + outb, res[c][0], res[c][1]);
        }
    };
    Frame f0, f1;
    Prep s0, s1;
    Gath g0, g1;
    load_frame(y0, f0);
    load_frame(y0 + 1, f1);
    stage1(y0, f0, s0, g0);
    for (int y = y0; y < y1; y += 2) {
        // the scheduling barriers keep the order loads -> gathers of the next row -> arithmetic of this row; left
        // alone the scheduler moves the arithmetic up and the gathers down to where they are needed
        load_frame(y + 2, f0);
        __builtin_amdgcn_sched_barrier(0);
        stage1(y + 1, f1, s1, g1);
        __builtin_amdgcn_sched_barrier(0);
        stage2(y, s0, g0);
        __builtin_amdgcn_sched_barrier(0);
        load_frame(y + 3, f1);
        __builtin_amdgcn_sched_barrier(0);
        stage1(y + 2, f0, s0, g0);
        __builtin_amdgcn_sched_barrier(0);
        stage2(y + 1, s1, g1);
        __builtin_amdgcn_sched_barrier(0);
    }
    LL_PROBE_T(pt3);
    LL_PROBE_ADD(8, pt1 - pt0); LL_PROBE_ADD(9, pt2 - pt1); LL_PROBE_ADD(10, 1);
    LL_PROBE_ADD(11, pt3 - pt2); LL_PROBE_ADD(12, pt3 - pt0);
}

// ---------------------------------------------------------------------------------------------------
// ll_up0h: the up pass of the re-cut dataflow (ll_down01e): outGPyramid[0] = upsample(outGPyramid[1]) + outLPyramid[0]
// (:76-79) + recolouring (:82-87), with outLPyramid[0] read as ONE stored plane and outGPyramid[1] collapsed per workgroup tile
// in LDS from the three level-1 planes ll_down01e stored (phase 1 of ll_up0f<.., .., true>, the two data-dependent plane
// reads of level 1 replaced by planes 0 and 1).  No remap table, no plane gathers: per pixel pair and row three u16 pairs, one
// float2 and six LDS reads.  Arithmetic: ll_up0f's (hl0 / hl1 / vl, div3_by).
struct Up0HArgs {
    Up0Args u;
    const float *outl0;        // outLPyramid[0] on [ix0, ix1] x [oy0, oy0 + oh - 1]
    int l0_ws;                 // its row stride in floats (= input width)
    // FUSE2: outGPyramid[2] on the part of R_2 the workgroup's level-1 tile reads is produced here too (phase 0, an LDS tile of at
    // most 68 x (RU / 2 + 3) values, by the expression of ll_up) instead of by an ll_up:2 launch: one dependent launch less in the
    // chain between the two big kernels, which costs a frame queue 8 us of every frame
    int fuse2;
    const float *g3, *out3;
    int lox3, loy3, ws3;
    size_t ps3;
};
constexpr int U0H_PF = 4;      // rows in flight per wave
constexpr int U0H_T2 = 68;     // row stride of the level-2 tile
template<bool NT, int CH = 2>  // CH: tile values a thread requests at a time in the two tile phases (1: 102.0, 2: 101.1, 4: 102.4 us per
                               // frame on one stream — 4 costs occupancy: 124 VGPRs against 66)
__global__ __launch_bounds__(256) void ll_up0h(Up0HArgs ph, Geometry gm) {
    const Up0Args &p = ph.u;
    LL_RESIDENCY(1);
#if HLMI_LL_UP0_PRIO
    __builtin_amdgcn_s_setprio(HLMI_LL_UP0_PRIO);
#endif
    extern __shared__ float s_out1[];
    // tiles in row-major order, a contiguous run of them per XCD (blocks are dealt round-robin over the 8 XCDs): a tile's
    // 130 x (RU + 2) coarse window overlaps its neighbours' by two columns / rows, and its 130-float rows start one float before a
    // 512-byte boundary — 6 lines for 4 of payload, the outer two shared with the neighbour tile: same L2 now (131.5 -> 118.1 MB
    // fetched per 4K frame)
    int bx, by;
    {
        const int b = (int)blockIdx.y * (int)gridDim.x + (int)blockIdx.x, nb8 = (int)(gridDim.x * gridDim.y) >> 3;
        const int lb = b < (nb8 << 3) ? (b & 7) * nb8 + (b >> 3) : b;
        by = lb / (int)gridDim.x, bx = lb - by * (int)gridDim.x;
    }
    const int X0 = p.ox0 + bx * 256;                       // even
    const int Yw0 = p.oy0 + by * (2 * p.RU);
    const int Yw1 = min(Yw0 + 2 * p.RU, p.oy0 + p.oh) - 1;
    const int cx0 = (X0 >> 1) - 1, cy0 = dev::fdiv2(Yw0 - 1);
    const int th = dev::fdiv2(Yw1 + 1) - cy0 + 1;
    float *const s_out2 = s_out1 + U0_TS * (p.RU + 2);
    const int cx1 = min(cx0 + U0_TW - 1, p.rx1_1);                               // last level-1 column of the tile
    const int c2x0 = dev::fdiv2(cx0 - 1), c2y0 = dev::fdiv2(cy0 - 1);             // level-2 window the level-1 tile's upsampling reads
    // The two tile phases below are each a handful of values per thread, and a value takes two DEPENDENT memory round trips
    // (inGPyramid -> which planes -> their bilinear taps).  As per-element loops they ran those trips element after element — and
    // every workgroup of the launch (one round of resident workgroups) sat in that prologue at the same time, the memory system idle.
    // Round 5: CH elements per thread at a time, all first loads, then all gathers, then the arithmetic (outl_value's and up_at's
    // operations in their order): two trips per chunk.  Worth ~1 % of the frame: the resident workgroups of a CU already interleave
    // their prologues.  Every load is `uniform base + 32-bit byte offset` (the workspace offsets of this path are < 4 GB): as
    // 64-bit element indices the address arithmetic was 60 % of the 148 VALU instructions a tile value cost, and the tile values
    // 45 % of the launch's instructions.
    // The four taps of an upsampled value are f(yb, xb), f(yb, xb + 1) and the same pair one row on (fdiv2(v + 1) = fdiv2(v - 1) + 1):
    // two unaligned 8-byte loads from two byte offsets.  Rows are numbered from the tile's first one, so every row x stride product
    // is (small) x (stride < 2^24, the host checks) = one full-rate v_mul_u32_u24 on top of a scalar base; e / tile width is a
    // multiplication by ceil(2^22 / width), exact while e x width < 2^22 (130 x 302 rows x 130 at the largest tile LDS can hold).
    struct Taps32 {
        uint32_t b, a;             // byte offsets of f(yb, xb) and f(yb + 1, xb) inside a plane
    };
    auto mul24 = [](uint32_t a, uint32_t b) { return (uint32_t)__umul24(a, b); };
    auto tap_bytes = [&](uint32_t row0, int y0, int lox, int ws, int X, int Y) {   // row0 = (y0 - loy) * ws (uniform), y0 <= fdiv2(Y - 1)
        Taps32 t;
        t.b = (row0 + mul24((uint32_t)(dev::fdiv2(Y - 1) - y0), (uint32_t)ws) + (uint32_t)(dev::fdiv2(X - 1) - lox)) << 2;
        t.a = t.b + ((uint32_t)ws << 2);
        return t;
    };
    auto taps_at = [](const float *base, uint32_t plane_bytes, const Taps32 &t) {
        const F2U lo = ld_su<F2U>(base, plane_bytes + t.b), hi = ld_su<F2U>(base, plane_bytes + t.a);
        UpTaps r;
        r.bb = lo.x, r.ba = lo.y, r.ab = hi.x, r.aa = hi.y;
        return r;
    };
    const uint32_t ps1b = (uint32_t)p.ps1 * 4u, ps2b = (uint32_t)p.ps2 * 4u, ps3b = (uint32_t)ph.ps3 * 4u;
    if (ph.fuse2) {
        const int n2x = dev::fdiv2(cx1 + 1) - c2x0 + 1, n2y = dev::fdiv2(cy0 + th) - c2y0 + 1, n2 = n2x * n2y;
        const uint32_t m2 = (4194304u + (uint32_t)n2x - 1u) / (uint32_t)n2x;
        const int c3y0 = dev::fdiv2(c2y0 - 1);
        const uint32_t row2 = (uint32_t)((c2y0 - p.loy2) * p.ws2), row3 = (uint32_t)((c3y0 - ph.loy3) * ph.ws3);
        for (int e0 = threadIdx.x; e0 < n2; e0 += 256 * CH) {
            int X2[CH], Y2[CH];
            uint32_t ti[CH];
            bool ok[CH];
            uint32_t ob[CH];
            Taps32 t3[CH];
            float inG[CH], lf[CH], ga[CH], gb[CH];
            UpTaps o3[CH], t0[CH], t1[CH];
#pragma unroll
            for (int i = 0; i < CH; i++) {
                const int e = e0 + 256 * i;
                const uint32_t ec = (uint32_t)min(e, n2 - 1), ty = mul24(ec, m2) >> 22, tx = ec - mul24(ty, (uint32_t)n2x);
                ok[i] = e < n2, ti[i] = mul24(ty, U0H_T2) + tx;
                X2[i] = c2x0 + (int)tx, Y2[i] = c2y0 + (int)ty;
                ob[i] = (row2 + mul24(ty, (uint32_t)p.ws2) + (uint32_t)(X2[i] - p.lox2)) << 2;
                inG[i] = ld_su<float>(p.g2, (uint32_t)gm.K * ps2b + ob[i]);
                t3[i] = tap_bytes(row3, c3y0, ph.lox3, ph.ws3, X2[i], Y2[i]);
                o3[i] = taps_at(ph.out3, 0u, t3[i]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < CH; i++) {
                const float level = inG[i] * gm.Km1;
                const int li = dev::clampi((int)level, 0, gm.K - 2);
                lf[i] = level - (float)li;
                ga[i] = ld_su<float>(p.g2, (uint32_t)li * ps2b + ob[i]), gb[i] = ld_su<float>(p.g2, (uint32_t)(li + 1) * ps2b + ob[i]);
                t0[i] = taps_at(ph.g3, (uint32_t)li * ps3b, t3[i]);
                t1[i] = taps_at(ph.g3, (uint32_t)(li + 1) * ps3b, t3[i]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < CH; i++) {
                // outGPyramid[2] = upsample(outGPyramid[3]) + outLPyramid[2]   (:76-79), exactly as ll_up computes it
                const float l0 = ga[i] - up_from(t0[i], X2[i], Y2[i]), l1 = gb[i] - up_from(t1[i], X2[i], Y2[i]);
                const float outL = dev::mad2(1.0f - lf[i], l0, lf[i], l1);
                if (ok[i]) s_out2[ti[i]] = up_from(o3[i], X2[i], Y2[i]) + outL;
            }
        }
        __syncthreads();
    }
    {
        const int n1 = U0_TW * th;
        constexpr uint32_t M1 = (4194304u + U0_TW - 1) / U0_TW;
        const uint32_t row1 = (uint32_t)((cy0 - p.loy1) * p.ws1), row2 = (uint32_t)((c2y0 - p.loy2) * p.ws2);
        for (int e0 = threadIdx.x; e0 < n1; e0 += 256 * CH) {
            int cx[CH], cy[CH];
            uint32_t ti[CH];
            bool ok[CH];
            Taps32 t2[CH];
            float inG[CH], lf[CH], ga[CH], gb[CH];
            UpTaps o2[CH], t0[CH], t1[CH];
#pragma unroll
            for (int i = 0; i < CH; i++) {
                const int e = e0 + 256 * i;
                const uint32_t ec = (uint32_t)min(e, n1 - 1), ty = mul24(ec, M1) >> 22, tx = ec - mul24(ty, U0_TW);
                ok[i] = e < n1 && cx0 + (int)tx <= p.rx1_1, ti[i] = mul24(ty, U0_TS) + tx;
                cx[i] = min(cx0 + (int)tx, p.rx1_1), cy[i] = cy0 + (int)ty;      // columns right of R_1 are not tile values: their threads re-read its last one
                const uint32_t ob = (row1 + mul24(ty, (uint32_t)p.ws1) + (uint32_t)(cx[i] - p.lox1)) << 2;
                // level 1 as ll_down01e stored it: plane 0 / 1 = gPyramid[1](., ., li / li + 1) of the pixel's own li, plane K = inGPyramid[1]
                inG[i] = ld_su<float>(p.g1, (uint32_t)gm.K * ps1b + ob), ga[i] = ld_su<float>(p.g1, ob), gb[i] = ld_su<float>(p.g1, ps1b + ob);
                t2[i] = tap_bytes(row2, c2y0, p.lox2, p.ws2, cx[i], cy[i]);
                if (!ph.fuse2) o2[i] = taps_at(p.out2, 0u, t2[i]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < CH; i++) {
                const float level = inG[i] * gm.Km1;
                const int li = dev::clampi((int)level, 0, gm.K - 2);
                lf[i] = level - (float)li;
                t0[i] = taps_at(p.g2, (uint32_t)li * ps2b, t2[i]);
                t1[i] = taps_at(p.g2, (uint32_t)(li + 1) * ps2b, t2[i]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < CH; i++) {
                // outGPyramid[1] = upsample(outGPyramid[2]) + outLPyramid[1]   (:76-79), exactly as ll_up computes it
                const float l0 = ga[i] - up_from(t0[i], cx[i], cy[i]), l1 = gb[i] - up_from(t1[i], cx[i], cy[i]);
                const float outL = dev::mad2(1.0f - lf[i], l0, lf[i], l1);
                float up2;
                if (ph.fuse2) {   // up_at on the LDS tile: the same taps, weights and lerps
                    const float *q = s_out2 + mul24((uint32_t)(dev::fdiv2(cy[i] - 1) - c2y0), U0H_T2) + (uint32_t)(dev::fdiv2(cx[i] - 1) - c2x0);
                    UpTaps t;
                    t.bb = q[0], t.ba = q[1], t.ab = q[U0H_T2], t.aa = q[U0H_T2 + 1];
                    up2 = up_from(t, cx[i], cy[i]);
                } else {
                    up2 = up_from(o2[i], cx[i], cy[i]);
                }
                if (ok[i]) s_out1[ti[i]] = up2 + outL;
            }
        }
    }
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int x = bx * 256 + (wave & 1) * 128 + 2 * lane;  // output storage column of the lane's pair
    const int y0 = by * (2 * p.RU) + (wave >> 1) * p.RU;
    if (y0 >= p.oh || x >= p.ow) return;
    const int y1 = min(y0 + p.RU, p.oh);
    const int X = p.ox0 + x;                                              // even
    uint32_t inb = (uint32_t)(X - gm.ix0) * 2u, outb = (uint32_t)x * 2u, l0b = (uint32_t)(X - gm.ix0) * 4u;
    // fma canon: the contracted product is the one with `zero` (dev::mad2) — f[c] * 3/4 for even X, f[c+1] * 1/4 for odd X
    auto hl0 = [](float rm, float r0) {   // lerp(f[c], f[c-1], 1/4): X even
        return dev::CANON_FMA ? __builtin_fmaf(r0, 0.75f, rm * 0.25f) : __builtin_fmaf(rm, 0.25f, r0 * 0.75f);
    };
    auto hl1 = [](float r0, float rp) { return __builtin_fmaf(rp, 0.25f, r0 * 0.75f); };  // lerp(f[c+1], f[c], 3/4): X odd
    auto vl = [](float uq, float ut, float wq, float wt) {   // as in ll_up0f
        return dev::CANON_FMA ? __builtin_fmaf(uq, wq, ut * wt) : __builtin_fmaf(uq, 0.25f, ut * 0.75f);
    };
    struct Frame {
        ushort2 c0, c1, c2;
        float2 l0;
    };
    // The lane's byte offsets are loop invariants: hoisted out of the row loop as 64-bit values they cost a v_lshl_add_u64 per load
    // and store (7 per row); redefined in place per use (an empty asm) they stay 32-bit and every access is `row base in SGPRs +
    // VGPR offset`.
    auto fresh = [](uint32_t &v) {
        asm volatile("" : "+v"(v));
        return v;
    };
    auto load_frame = [&](int y, Frame &f) {
        const int yc = min(y, y1 - 1);
        const uint16_t *irow = p.in + (long)(p.oy0 + yc - gm.iy0) * p.in_sy;
        const uint32_t ib = fresh(inb);
        f.c0 = ld_frame2<NT>(irow + p.gco[0], ib), f.c1 = ld_frame2<NT>(irow + p.gco[1], ib), f.c2 = ld_frame2<NT>(irow + p.gco[2], ib);
        {
            const f2 *const lp = reinterpret_cast<const f2 *>(reinterpret_cast<const char *>(ph.outl0 + (size_t)yc * ph.l0_ws) + fresh(l0b));
            const f2 v = NT ? __builtin_nontemporal_load(lp) : *lp;
            f.l0 = make_float2(v.x, v.y);
        }
    };
    auto finish = [&](int y, const Frame &f) {
        const int Y = p.oy0 + min(y, y1 - 1);
        const int ya = dev::fdiv2(Y + 1), yb = dev::fdiv2(Y - 1);
        const bool yodd = dev::fmod2(Y) != 0;  // wave-uniform
        // q: the coarse row whose weight is 1/4; fma canon: q = row ya (`zero`), t = row yb (`one`) with weights wq, wt (see ll_up0f)
        const int yq = (dev::CANON_FMA || yodd) ? ya : yb, yt = (dev::CANON_FMA || yodd) ? yb : ya;
        const float wq = yodd ? 0.25f : 0.75f, wt = yodd ? 0.75f : 0.25f;
        const float *oa = s_out1 + (yq - cy0) * U0_TS + (wave & 1) * 64 + lane;
        const float *ob = s_out1 + (yt - cy0) * U0_TS + (wave & 1) * 64 + lane;
        const float ax = oa[0], ay = oa[1], az = oa[2], bx = ob[0], by = ob[1], bz = ob[2];
        const float uo[2] = {vl(hl0(ax, ay), hl0(bx, by), wq, wt), vl(hl1(ay, az), hl1(by, bz), wq, wt)};
        const float outL[2] = {f.l0.x, f.l0.y};
        const uint16_t ch[3][2] = {{f.c0.x, f.c0.y}, {f.c1.x, f.c1.y}, {f.c2.x, f.c2.y}};
        uint16_t res[3][2];
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const float gray = gray_from(ch[0][i], ch[1][i], ch[2][i]);
            const float og = (uo[i] + outL[i]) + 0.01f;
            const float gr = gray + 0.01f;
            const float n[3] = {(float)ch[0][i] * og, (float)ch[1][i] * og, (float)ch[2][i] * og};
            float q[3];
            div3_by(n, gr, q);
#pragma unroll
            for (int c = 0; c < 3; c++) res[c][i] = (uint16_t)__builtin_amdgcn_fmed3f(q[c], 0.0f, 65535.0f);  // q is never NaN
        }
        if (y < y1) {
            uint16_t *orow = p.out + (long)y * p.out_sy;
#pragma unroll
            for (int c = 0; c < 3; c++) st_frame2<NT>(reinterpret_cast<char *>(orow + (long)c * p.out_sc) + fresh(outb), res[c][0], res[c][1]);
        }
    };
    Frame f[U0H_PF];
#pragma unroll
    for (int i = 0; i < U0H_PF; i++) load_frame(y0 + i, f[i]);
    for (int y = y0; y < y1; y += U0H_PF) {
#pragma unroll
        for (int i = 0; i < U0H_PF; i++) {
            const Frame cur = f[i];
            load_frame(y + U0H_PF + i, f[i]);
            finish(y + i, cur);
        }
    }
}

// ---------------------------------------------------------------------------------------------------
const int64_t est_zero = 0, est_w = 1536, est_h = 2560, est_c = 3;
const int64_t *const buf_est[6] = {&est_zero, &est_w, &est_zero, &est_h, &est_zero, &est_c};
const halide_scalar_value_t est_levels = [] { halide_scalar_value_t v{}; v.u.i32 = 8; return v; }();
const halide_scalar_value_t est_one = [] { halide_scalar_value_t v{}; v.u.f32 = 1.0f; return v; }();
const halide_type_t ty_u16 = {(decltype(halide_type_t::code))1, 16, 0};
const halide_type_t ty_i32 = {(decltype(halide_type_t::code))0, 32, 0};
const halide_type_t ty_f32 = {(decltype(halide_type_t::code))2, 32, 0};
// estimates: generator :92-98
const halide_filter_argument_t ll_args[5] = {
    {"input", halide_argument_kind_input_buffer, 3, ty_u16, nullptr, nullptr, nullptr, nullptr, buf_est},
    {"levels", halide_argument_kind_input_scalar, 0, ty_i32, nullptr, nullptr, nullptr, &est_levels, nullptr},
    {"alpha", halide_argument_kind_input_scalar, 0, ty_f32, nullptr, nullptr, nullptr, &est_one, nullptr},
    {"beta", halide_argument_kind_input_scalar, 0, ty_f32, nullptr, nullptr, nullptr, &est_one, nullptr},
    {"output", halide_argument_kind_output_buffer, 3, ty_u16, nullptr, nullptr, nullptr, nullptr, buf_est},
};
const halide_filter_metadata_t ll_md = {1, 5, ll_args, kTargetString, "local_laplacian"};

// ---- run-time switches (all default to the fast path; the env overrides exist for A/B measurements)
int env_int(const char *name, int dflt) {
    const char *e = getenv(name);
    return (e && *e) ? atoi(e) : dflt;
}

// rows per wave of the up kernels: ll_up0h turns tile element numbers into rows with multiplications that are exact for at most
// RU + 2 = 302 tile rows (tests/test_index_arithmetic.py), so the switch is clamped to the proven range
int dev_clamp_ru(int ru) { return ru < 1 ? 1 : (ru > 300 ? 300 : ru); }

// last call's level table, for hlmi_debug_local_laplacian_outg (tests only)
thread_local Level t_dbg_lv[J];
thread_local hipStream_t t_dbg_stream = nullptr;
thread_local bool t_dbg_out1_pending = false;  // the last call fused level 1's collapse: outGPyramid[1] was never stored
thread_local bool t_dbg_out2_pending = false;  // ... and level 2's (ll_up0h phase 0)
thread_local bool t_dbg_emit = false;         // ... by ll_up0h: level 1 holds its three planes only (ll_down01e)
thread_local int t_dbg_K = 0;
thread_local float t_dbg_Km1 = 0;

// ---- cache of remap tables: the LUT is a function of (levels, alpha) only — a video stream calls with the same pair frame
// after frame — so it is kept in memory of its own per (device, levels, alpha) and ll_remap_lut runs when the pair is new
// (one 4.5 us launch less per frame).  An entry is used by other streams behind the event recorded after its launch.
struct LutImage {
    int device = -1, levels = 0;
    uint32_t alpha_bits = 0;
    bool valid = false;
    float *dev = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ready = nullptr;
    uint64_t used = 0;
};
std::mutex g_lut_mu;
LutImage g_lut[8];
uint64_t g_lut_clock = 0;

// ---- HIP graphs.  One frame is a chain of 8 dependent launches whose arguments are a pure function of (buffers, shape,
// parameters, workspace, switches); callers that process a stream of equally shaped frames into the same buffers (bench.py,
// apps/local_laplacian/process.cpp:38's benchmark loop, a video pipeline with a ring of frames) repeat the same chain.  The
// SECOND call with a given key captures the chain on the call's stream (hipStreamBeginCapture, thread-local mode) and
// instantiates it; that call and every later one replay it with one hipGraphLaunch instead of 8 launches' worth of host
// work.  Keys that differ in any pointer, extent, stride, parameter, switch or in the workspace address never match, so a
// graph can only replay launches that the eager path would have issued with identical arguments.  The capture runs on a
// private stream of the calling thread (capturing records, it does not execute — and a capture on the caller's own stream would
// swallow or be invalidated by whatever another host thread enqueues there meanwhile, e.g. a halide_copy_to_host); the
// instantiated graph is then launched on the caller's stream.
// OPT-IN (HLMI_LL_GRAPH=1): measured on MI355X it buys nothing — 84.2 vs 83.4 Gpx/s on four frame-queue streams (inside the
// box-to-box noise) and 69.1 vs 72.7 Gpx/s on one stream (profiles/r03a_*): the GPU-side gap between dependent launches is the
// same for a graph and for eager launches, and the host (44 us of enqueue per 99 us frame) is not the bottleneck.
struct GraphKey {
    int device;
    hipStream_t stream;
    const void *in, *out, *ws, *lut;
    int32_t idim[3][3], odim[3][3];   // min, extent, stride
    int32_t levels;
    uint32_t alpha_bits, beta_bits;
    uint64_t env_sig;
    bool operator==(const GraphKey &o) const { return memcmp(this, &o, sizeof(GraphKey)) == 0; }
};
struct GraphEntry {
    GraphKey key;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    bool failed = false;       // capture or instantiation was refused once: stay eager for this key
    bool out1_pending = false, out2_pending = false; // debug bookkeeping of the captured call (hlmi_debug_local_laplacian_outg)
    uint64_t used = 0;
};
std::mutex g_graph_mu;
std::vector<GraphEntry> g_graphs;   // small (<= 64): linear search
uint64_t g_graph_clock = 0;

uint64_t ll_env_signature() {
    static const char *const names[] = {"HLMI_LL_NO_LUT_CACHE", "HLMI_LL_UNITS0", "HLMI_LL_NO_VEC",
                                        "HLMI_LL_D01_EXCH", "HLMI_LL_FUSE_FROM", "HLMI_LL_UPCHAIN_FROM", "HLMI_LL_RU", "HLMI_LL_EMIT", "HLMI_LL_NT",
                                        "HLMI_LL_FUSE_UP2"};
    uint64_t h = 1469598103934665603ull;
    for (const char *n : names) {
        const char *e = getenv(n);
        for (const char *c = e ? e : ""; ; c++) {
            h = (h ^ (uint8_t)*c) * 1099511628211ull;
            if (!*c) break;
        }
    }
    return h;
}

}  // namespace

extern "C" int local_laplacian(halide_buffer_t *input, int32_t levels, float alpha, float beta, halide_buffer_t *output) {
    void *uc = nullptr;
    BufArg args[2] = {{"input", input, T_U16, 3, false}, {"output", output, T_U16, 3, true}};
    int r = check_not_null(uc, args, 2);
    if (r) return r;
    if ((r = check_type_and_dims(uc, args, 2))) return r;
    if (any_bounds_query(args, 2)) {
        // input is needed exactly on the output's region: every other tap goes through repeat_edge (:28, :84)
        int mins[3], ext[3];
        for (int d = 0; d < 3; d++) mins[d] = output->dim[d].min, ext[d] = output->dim[d].extent;
        answer_query(input, mins, ext);
        answer_query(output, mins, ext);
        return 0;
    }
    if ((r = check_shape(uc, args[0])) || (r = check_shape(uc, args[1]))) return r;
    for (int d = 0; d < 3; d++) {
        if ((r = check_covers(uc, args[0], d, output->dim[d].min, output->dim[d].extent))) return r;
    }
    // The reference declares `Input<int> levels` without a range (generator :13) and divides by levels - 1 (:41): any value
    // >= 2 is accepted (the remap table lives in LDS up to 15 levels, in global memory beyond; memory grows with levels + 1
    // planes per pyramid level and an impossible request fails with device_malloc_failed like any other allocation).  levels
    // < 2 makes 1 / (levels - 1) infinite or negative — the one value range with no defined result — and is rejected.
    if (levels < 2) {
        return report(uc, halide_error_code_param_too_small, "Parameter levels is %d but must be at least 2", levels);
    }
    if (levels > (1 << 20)) {   // (levels - 1) * 256 table entries and plane counts must stay inside 32-bit index arithmetic
        return report(uc, halide_error_code_param_too_large, "Parameter levels is %d but must be at most %d", levels, 1 << 20);
    }
    const int ow = output->dim[0].extent, oh = output->dim[1].extent, nc = output->dim[2].extent;
    if (nc > 3) {
        // `color` is defined for c in [0,3) only (:84 reads input(x,y,c), whose extent the check above bounds)
        return report(uc, halide_error_code_constraint_violated, "Output buffer output has %d channels, at most 3 supported", nc);
    }

    DeviceCtx ctx;
    if ((r = acquire_device(uc, &ctx))) return r;
    if ((r = input_to_device(uc, ctx, args[0]))) return r;
    if ((r = output_on_device(uc, ctx, args[1]))) return r;
    if (ow == 0 || oh == 0 || nc == 0) {
        mark_output_written(output);
        return 0;
    }

    Geometry gm;
    gm.K = levels;
    gm.half = (levels - 1) * 256;
    gm.Km1 = (float)(levels - 1);
    gm.inv_Km1 = 1.0f / gm.Km1;
    gm.ix0 = input->dim[0].min, gm.ix1 = gm.ix0 + input->dim[0].extent - 1;
    gm.iy0 = input->dim[1].min, gm.iy1 = gm.iy0 + input->dim[1].extent - 1;
    const int ic0 = input->dim[2].min, ic1 = ic0 + input->dim[2].extent - 1;

    // per-level boxes.  lo/hi: beyond them the level is constant; so: storage origin (<= lo) whose parity
    // makes the strip kernels' 16-byte loads / 8-byte stores aligned (see ll_down0 / hpair).
    Level lv[J];
    int lox = gm.ix0, hix = gm.ix1, loy = gm.iy0, hiy = gm.iy1, so = gm.ix0;
    int rx0 = output->dim[0].min, rx1 = rx0 + ow - 1, ry0 = output->dim[1].min, ry1 = ry0 + oh - 1;
    size_t ws_floats = (size_t)(2 * gm.half + 1 + 63) & ~(size_t)63;
    size_t off_g[J], off_out[J];
    for (int j = 0; j < J; j++) {
        Level &L = lv[j];
        L.lox = so, L.loy = loy, L.w = hix - so + 1, L.h = hiy - loy + 1;
        L.ws = (L.w + 3) & ~3;
        L.ps = ((size_t)L.ws * L.h + 3) & ~(size_t)3;
        L.rx0 = rx0, L.rx1 = rx1, L.ry0 = ry0, L.ry1 = ry1;
        if (j >= 1) {
            off_g[j] = ws_floats;
            ws_floats += ((size_t)(levels + 1) * L.ps + 63) & ~(size_t)63;
            off_out[j] = ws_floats;
            ws_floats += (L.ps + 63) & ~(size_t)63;
        }
        // next level
        const bool odd = (so & 1) != 0;
        const int par = odd ? (floor_div(so + 1, 2) & 1) : ((floor_div(so, 2) + 1) & 1);
        lox = floor_div(lox - 2, 2), hix = floor_div(hix + 2, 2);
        loy = floor_div(loy - 2, 2), hiy = floor_div(hiy + 2, 2);
        so = lox - ((lox - par) & 1);
        if (j + 1 < J) {
            lv[j + 1].odd = odd;
            lv[j + 1].nsx = (hix - so + 1 + STRIP - 1) / STRIP;
        }
        rx0 = floor_div(rx0 - 1, 2), rx1 = floor_div(rx1 + 1, 2);
        ry0 = floor_div(ry0 - 1, 2), ry1 = floor_div(ry1 + 1, 2);
    }
    const uint16_t *din = dev_ptr<uint16_t>(input);
    uint16_t *dout = dev_ptr<uint16_t>(output);
    const long in_sy = input->dim[1].stride, in_sc = input->dim[2].stride;
    const long out_sy = output->dim[1].stride, out_sc = output->dim[2].stride;
    // clamped channels feeding `gray` (repeat_edge clamps the channel coordinate too, :28)
    long gco[3];
    for (int c = 0; c < 3; c++) gco[c] = (long)((c < ic0 ? ic0 : (c > ic1 ? ic1 : c)) - ic0) * in_sc;
    hipStream_t st = ctx.stream;
    const bool lut_lds = levels <= 15;  // LUT must fit the default 64 KB dynamic-LDS window
    const int nlut = 2 * gm.half + 1;
    const size_t lut_sh = lut_lds ? sizeof(float) * nlut : 0;

    // levels >= S are produced / collapsed by the two multi-level kernels (S = 4: 2 launches instead of 7)
    const int S = [&] {
        int v = env_int("HLMI_LL_FUSE_FROM", 4);
        return (v >= J - 5 && v <= J - 2) ? v : J;
    }();
    // the collapse (outGPyramid[J-1] .. outGPyramid[SU]) is ONE launch (ll_up_multi); opt-in: on the large levels its per-pixel overhead exceeds the saved launches
    // Default 3 when the down pass fuses from 4: outGPyramid[3] joins the collapse launch (one ll_up launch and its ~4.5 us of
    // dependent-launch latency less: 110.8 -> 108.9 us per frame on one stream); from level 2 the kernel's per-pixel overhead
    // costs more than the launch it saves (114.2).  0: SU = S.
    const int SU = [&] {
        int v = env_int("HLMI_LL_UPCHAIN_FROM", S == 4 ? 3 : 0);   // (S == 3: SU = S)
        return (v >= 1 && v <= J - 2) ? v : S;
    }();
    // ---- which kernels run: decided before the workspace is sized (the re-cut dataflow's outLPyramid[0] plane is only requested
    // by the calls that fill it)
    Up0Args p;
    bool vec, fast, fuse1;
    const int stream_cus = stream_cu_count(ctx.device, ctx.stream);
    const bool partitioned = stream_cus < stream_cu_count(ctx.device, nullptr);
    {
        p.in = din, p.in_sy = in_sy;
        const int oc0 = output->dim[2].min;
        bool same = (nc == 3);
        for (int ch = 0; ch < 3; ch++) {
            p.gco[ch] = gco[ch];
            p.cco[ch] = ch < nc ? (long)(oc0 + ch - ic0) * in_sc : 0;
            if (p.cco[ch] != p.gco[ch]) same = false;
        }
        p.same_ch = same ? 1 : 0;
        p.out = dout, p.out_sy = out_sy, p.out_sc = out_sc;
        p.ox0 = output->dim[0].min, p.oy0 = output->dim[1].min, p.ow = ow, p.oh = oh, p.nc = nc;
        p.beta = beta;
        vec = ((uintptr_t)din % 4 == 0) && ((uintptr_t)dout % 4 == 0) && in_sy % 2 == 0 && out_sy % 2 == 0 &&
              out_sc % 2 == 0 && ((p.ox0 - gm.ix0) % 2 == 0) && !env_int("HLMI_LL_NO_VEC", 0);
        for (int ch = 0; ch < 3; ch++) vec = vec && p.gco[ch] % 2 == 0 && p.cco[ch] % 2 == 0;
        fast = vec && same && nc == 3 && (ow & 1) == 0 && (p.ox0 & 1) == 0 &&
               (double)(levels + 1) * (4.0 * (double)lv[1].ps) < 4.0e9;
        // the fused collapse needs level 2 to be a stored level of its own (SU >= 2 always holds: SU >= S >= 4 or the
        // opt-in up-chain, which starts at >= 1 and then owns level 1 itself)
        fuse1 = fast && SU >= 2;
        // rows per wave: taller tiles re-read less of level 1 (18 coarse rows per 16 output rows, 34 per 32) but keep a wave
        // busy longer.  On a frame-queue stream (`partitioned`: one of several library queues with frames in flight, runtime.cpp),
        // where several frames share the memory system and the frame rate is set by
        // bytes, 32 rows measure 2.7 % faster (84.8 vs 82.6 Gpx/s); on a stream that owns the device 16 rows do (72.7 vs 68.2).
        p.RU = dev_clamp_ru(env_int("HLMI_LL_RU", fuse1 ? (partitioned ? 32 : 16) : 8));
    }
    // ll_down01f / ll_down01e: levels 1 and 2 from the input in one walk (levels == KCH planes in registers, 8-byte input loads)
    const bool d01_possible = levels == KCH && lut_lds &&
                              ((uintptr_t)din % 8 == 0) && in_sy % 4 == 0 && gco[0] % 4 == 0 && gco[1] % 4 == 0 && gco[2] % 4 == 0 &&
                              (gm.ix1 - gm.ix0 + 1) % 4 == 0 && !env_int("HLMI_LL_NO_VEC", 0);
    // The default for the common geometry: ll_down01e emits outLPyramid[0] and three planes of level 1, ll_up0h collapses
    // (HLMI_LL_EMIT=0: the round-3 pair ll_down01f / ll_up0f with the materialised K + 1 level-1 planes)
    const bool emit = d01_possible && fast && fuse1 && lv[1].ws < (1 << 24) && env_int("HLMI_LL_EMIT", 1);   // ws: ll_up0h's 24-bit row products
    // ll_up0h has no data-dependent gathers to amortise over a tall tile: short tiles (more, smaller workgroups) are faster on a
    // stream that owns the device (31.7 us at 8 rows per wave against 33.5 / 38.4 at 16 / 32); with four frames in flight 8 / 12 / 16 /
    // 24 / 32 rows measure 107.8 / 110.1 / 111.2 / 112.3 / 112.3 Gpx/s (profiles/r06_frame_queue_geometry.txt), and next to ONE
    // resident ll_down01e workgroup per CU 32 rows beat 24 / 40 / 48 / 64 (profiles/r06_coresidency_ab.txt)
    if (emit) p.RU = dev_clamp_ru(env_int("HLMI_LL_RU", partitioned ? 32 : 8));
    // non-temporal frame / outLPyramid[0] accesses: +6-7 % frames per second with four frames in flight, -2-3 % on a stream that owns the device
    const bool nt = env_int("HLMI_LL_NT", partitioned ? 1 : 0) != 0;
    // ll_up0h also collapses level 2 (into an LDS tile) when level 3 is a stored level of its own: the ll_up:2 launch goes
    // (with four frames in flight: 79.4 -> 76.4 us per frame; on a stream that
    // owns the device the tile redundancy used to cost what the launch saved — 109 -> 111 us in round 4 — until round 5's batched
    // tile phases: 104.1 -> 98.7 us per frame, 115 -> 110.6 for one call + sync)
    const bool fuse2 = emit && SU >= 3 && SU < J && env_int("HLMI_LL_FUSE_UP2", 1);

    // OPT-IN (HLMI_LL_FUSE_MID=1): ll_down_multi:4 and ll_up_multi:3 as ONE launch (ll_mid: levels 5-7 handed over inside the launch).
    // Bit-exact; with the consumers' older-level loads requested before they wait it is 1.9 us per frame FASTER on one stream (96.0
    // against 97.9 back to back, 111.5 against 113.0 for one call + sync) and 0.6 us per frame SLOWER with four frames in flight (71.4
    // against 70.8: its consumer workgroups sit on the CUs while the producers run) — profiles/r06_ll_mid_ab.txt.  Not the default on
    // any stream: ONE such launch cannot deadlock (ll_mid's comment), but the waiting consumers of SEVERAL of them in flight on
    // different queues could in principle fill an XCD's workgroup slots while a producer of each still waits for one there.
    const bool mid = emit && S == 4 && SU == 3 && J == 8 && env_int("HLMI_LL_FUSE_MID", 0);
    // ---- workspace: the levels and outLPyramid[0] of the re-cut dataflow (input width x output rows)
    const size_t off_l0 = ws_floats;
    if (emit) ws_floats += ((size_t)(gm.ix1 - gm.ix0 + 1) * (size_t)oh + 63) & ~(size_t)63;
    const size_t off_ctr = ws_floats;   // ll_mid's producer count (one word, a cache line of its own)
    if (mid) ws_floats += (MID_WORDS + 63) & ~63;
    void *ws = nullptr;
    if ((r = get_workspace(uc, ctx, ws_floats * sizeof(float), &ws))) return r;
    float *wsf = (float *)ws;
    float *lut = wsf;
    float *outl0 = wsf + off_l0;
    for (int j = 1; j < J; j++) lv[j].g = wsf + off_g[j], lv[j].out = wsf + off_out[j];
    lv[0].g = lv[0].out = nullptr;
    for (int j = 0; j < J; j++) t_dbg_lv[j] = lv[j];
    t_dbg_stream = ctx.stream;

    if (!env_int("HLMI_LL_NO_LUT_CACHE", 0)) {
        uint32_t abits;
        memcpy(&abits, &alpha, 4);
        std::unique_lock<std::mutex> lock(g_lut_mu);
        LutImage *hit = nullptr, *slot = &g_lut[0];
        for (auto &e : g_lut) {
            if (e.valid && e.device == ctx.device && e.levels == levels && e.alpha_bits == abits) hit = &e;
        }
        if (hit) {
            hit->used = ++g_lut_clock;
            if (hit->stream != st) HLMI_HIP(uc, wait_done(st, hit->ready));
            lut = hit->dev;
        } else {
            for (auto &e : g_lut) {
                if (!e.dev) { slot = &e; break; }
                if (e.used < slot->used) slot = &e;
            }
            slot->valid = false;
            if (slot->dev) {   // evicting (more than 8 (levels, alpha) pairs in use): launches on any stream may still read it
                HLMI_HIP(uc, hipDeviceSynchronize());
                (void)hipFree(slot->dev);
                slot->dev = nullptr;
            }
            HLMI_HIP(uc, hipMalloc((void **)&slot->dev, sizeof(float) * ((size_t)nlut + 64)));
            if (!slot->ready) HLMI_HIP(uc, hipEventCreateWithFlags(&slot->ready, hipEventDisableTiming));
            slot->device = ctx.device, slot->levels = levels, slot->alpha_bits = abits, slot->stream = st, slot->used = ++g_lut_clock;
            lut = slot->dev;
            HLMI_LAUNCH(uc, "ll_remap_lut", st, ll_remap_lut, dim3((nlut + 255) / 256), dim3(256), 0, lut, gm.half, alpha);
            HLMI_HIP(uc, record_done(slot->ready, st));
            slot->valid = true;
        }
    } else {
        HLMI_LAUNCH(uc, "ll_remap_lut", st, ll_remap_lut, dim3((nlut + 255) / 256), dim3(256), 0, lut, gm.half, alpha);
    }
    {
        const Level &c = lv[1];
        p.lut_g = lut, p.g1 = c.g, p.out1 = c.out;
        p.lox1 = c.lox, p.loy1 = c.loy, p.ws1 = c.ws, p.ps1 = c.ps;
        p.g2 = lv[2].g, p.out2 = lv[2].out, p.lox2 = lv[2].lox, p.loy2 = lv[2].loy, p.ws2 = lv[2].ws, p.ps2 = lv[2].ps;
        p.rx1_1 = c.rx1, p.rx0_1 = c.rx0, p.ry0_1 = c.ry0, p.ry1_1 = c.ry1;
    }
    // the stages of the chain between the two big kernels
    auto strip_args = [&](int j, int target_units) {   // level j -> j + 1
        const Level &sl = lv[j], &d = lv[j + 1];
        const int cols = d.nsx * (levels + 1);
        StripArgs a;
        a.src = sl.g, a.slox = sl.lox, a.sloy = sl.loy, a.sw = sl.w, a.sh = sl.h, a.sws = sl.ws, a.sps = sl.ps;
        a.dst = d.g, a.Xs = d.lox, a.dloy = d.loy, a.dw = d.w, a.dh = d.h, a.dws = d.ws, a.dps = d.ps;
        a.nsx = d.nsx;
        a.nsy = max(1, min(max(target_units / cols, (d.h + 31) / 32), max(1, d.h / 2)));
        a.nunits = cols * a.nsy;
        return a;
    };
    auto coarse_args = [&](int from) {
        CoarseArgs ca;
        for (int dl = 0; from + dl < J; dl++) {
            const Level &L = lv[from + dl];
            DevLevel &D = ca.lv[dl];
            D.g = L.g, D.out = L.out, D.lox = L.lox, D.loy = L.loy, D.w = L.w, D.h = L.h, D.ws = L.ws, D.ps = (unsigned)L.ps;
            D.rx0 = L.rx0, D.ry0 = L.ry0, D.rw = L.rx1 - L.rx0 + 1, D.rh = L.ry1 - L.ry0 + 1;
        }
        ca.K = levels, ca.Km1 = gm.Km1;
        return ca;
    };
    auto up_args = [&](int j) {
        const Level &a = lv[j], &c = lv[j + 1];
        UpArgs u;
        u.g = a.g, u.ws = a.ws, u.ps = a.ps, u.lox = a.lox, u.loy = a.loy, u.gc = c.g, u.outc = c.out, u.cws = c.ws, u.cps = c.ps;
        u.clox = c.lox, u.cloy = c.loy, u.rx0 = a.rx0, u.ry0 = a.ry0, u.rw = a.rx1 - a.rx0 + 1, u.rh = a.ry1 - a.ry0 + 1;
        u.K = levels, u.Km1 = gm.Km1, u.out = a.out;
        return u;
    };
    bool fuse1_out = false, fuse2_out = false;
    auto enqueue = [&]() -> int {   // the launch chain of one frame (everything below depends only on what GraphKey holds)
    bool fuse_d2 = false;
    {
        const Level &d = lv[1];
        // two waves per SIMD with (almost) equal row counts: the kernel is VALU-bound, so balance is what counts
        const int target = env_int("HLMI_LL_UNITS0", 8 * stream_cu_count(ctx.device, nullptr));
        const int nsy = max(1, min(max(target / d.nsx, (d.h + 63) / 64), max(1, d.h / 2)));
        const int nunits = d.nsx * nsy;
        const int iw = gm.ix1 - gm.ix0 + 1;
        const bool vec = ((uintptr_t)din % 8 == 0) && in_sy % 4 == 0 && gco[0] % 4 == 0 && gco[1] % 4 == 0 &&
                         gco[2] % 4 == 0 && iw % 4 == 0 && !env_int("HLMI_LL_NO_VEC", 0);
        Levels lev;
        for (int k = 0; k < MAX_K; k++) lev.v[k] = (float)k * gm.inv_Km1;
        const bool b1 = (beta == 1.0f);
        const int variant = (d.odd ? 8 : 0) | (vec ? 4 : 0) | (lut_lds ? 2 : 0) | (b1 ? 1 : 0);
        constexpr int WPB = D0_THREADS / 64;
        dim3 grid((nunits + WPB - 1) / WPB), block(D0_THREADS);
        const size_t d0_sh = (lut_lds ? sizeof(float) * ((nlut + 1) & ~1) : 0) + sizeof(float2) * D0_THREADS * (KCH + 1);
        // algorithmic bytes: the input read once (u16 x 3 channels), the K+1 level-1 planes written once
        const double d0_bytes = 6.0 * iw * (gm.iy1 - gm.iy0 + 1) + 4.0 * (levels + 1) * d.w * d.h;
        auto launch_d0 = [&](auto kern) -> int {
            timing_note_bytes(d0_bytes);
            HLMI_LAUNCH(uc, "ll_down0", st, kern, grid, block, d0_sh, din, in_sy, gco[0], gco[1], gco[2], gm, lev, beta,
                        lut, d.g, d.lox, d.loy, d.w, d.h, d.ws, d.ps, d.nsx, nsy, nunits);
            return 0;
        };
#define LL_D0(O, V, L, B)                                             \
    case ((O ? 8 : 0) | (V ? 4 : 0) | (L ? 2 : 0) | (B ? 1 : 0)):      \
        r = launch_d0(&ll_down0<O, V, L, B>);                         \
        break;
        fuse_d2 = levels == KCH && vec && lut_lds;
        if (fuse_d2) {
            // ---- levels 1 AND 2 from the input in one walk (ll_down01f); the ll_down_strip:1 launch below is skipped
            const Level &e = lv[2];
            D01Args a;
            a.in = din, a.in_sy = in_sy, a.co0 = gco[0], a.co1 = gco[1], a.co2 = gco[2], a.beta = beta, a.lut_g = lut;
            a.g1 = d.g, a.so1 = d.lox, a.loy1 = d.loy, a.w1 = d.w, a.h1 = d.h, a.ws1 = d.ws, a.ps1 = d.ps;
            a.g2 = e.g, a.so2 = e.lox, a.loy2 = e.loy, a.w2 = e.w, a.h2 = e.h, a.ws2 = e.ws, a.ps2 = e.ps;
            const bool odd0 = d.odd, odd1 = e.odd;   // e.odd == (d.lox & 1)
            a.S2 = (odd0 || odd1) ? 62 : 61;
            const int lim = odd1 ? 2 * e.lox - 1 : 2 * e.lox - 2;    // leftmost pair must reach level-2 column so2
            a.Pbase = min(d.lox, lim);                                // same parity as so1 in either case
            const int hi1 = d.lox + d.w - 1, hi2 = e.lox + e.w - 1;
            const int x2_first = odd1 ? (a.Pbase + 1) / 2 + 0 : a.Pbase / 2 + 1;   // Pbase + 1 (resp. Pbase) is even: exact
            a.nsx = max((hi2 - x2_first + a.S2) / a.S2, (hi1 - a.Pbase + 2 * a.S2) / (2 * a.S2));
            // ll_down01f: sized for the whole device on every stream (fewer, taller units measured 2-3 % slower: 101.5 vs 98.9 us per
            // frame).  ll_down01e on a frame-queue stream (several frames in flight, the launches of different frames fill the device
            // together): fewer and taller units — fewer seam rows walked twice, less per-workgroup set-up — measure 10 % more frames
            // per second than the 2048 units of a launch that has the device to itself (round 6, four queues: 2048 / 1536 / 1024 /
            // 896..384 / 256 units -> 109.4 / 110.2 / 111.3 / 112.1-112.5 / 103.3 Gpx/s, profiles/r06_frame_queue_geometry.txt; with
            // one workgroup per CU — below — 640 units = 160 workgroups of 55 level-2 rows measure best),
            // while on a stream that owns the device one round of resident waves is what counts (52.7 us against 57.1)
            const int target2 = env_int("HLMI_LL_UNITS0", emit && partitioned ? 10 * stream_cus : 8 * stream_cu_count(ctx.device, nullptr));
            // EXCH: a workgroup = 4 vertically adjacent units exchanging their seam rows through LDS.  With n level-2 rows
            // per wave a workgroup owns R = 4 n - 1 rows (the bottom wave walks the two seam rows of the next workgroup
            // itself and owns one row less); n = the smallest that keeps the launch within `target2` resident waves.
            bool exch = env_int("HLMI_LL_D01_EXCH", 1) != 0;
            int nwy = 0;
            auto ceil_div = [](int x, int y) { return (x + y - 1) / y; };
            if (exch) {
                const int nwy_max = max(1, target2 / (WPB * a.nsx));
                const int n = max(2, (ceil_div(e.h, nwy_max) + 1 + 3) / 4);
                nwy = ceil_div(e.h, 4 * n - 1);
                exch = e.h / nwy >= 4;     // every workgroup gets at least 4 rows; smaller images take the plain units
            }
            if (exch) {
                a.nsy = nwy;
                a.nunits = a.nsx * nwy * WPB;
            } else {
                a.nsy = max(1, min(max(target2 / a.nsx, (e.h + 31) / 32), e.h));
                a.nunits = a.nsx * a.nsy;
            }
            a.nsy_magic = a.nsy == 1 ? 0u : (unsigned)((1ull << 32) / (unsigned)a.nsy + 1ull);   // 0: nsy == 1
            a.rows_base = e.h / a.nsy, a.rows_rem = e.h % a.nsy;
            dim3 grid2((a.nunits + WPB - 1) / WPB);
            size_t sh2 = sizeof(float) * ((nlut + 1) & ~1) + sizeof(float2) * D01_STATE * (WPB + (exch ? WPB - 1 : 0));
            // With frames in flight (a frame-queue stream) ll_down01e asks for so much LDS that only ONE of its workgroups fits a
            // CU (2 x 77 KB would): the other half of the CU's registers and 77 KB of its LDS stay free for the workgroups of the
            // OTHER frames' kernels — ll_up0h above all, which waits for memory while this one computes.  Four frames in flight,
            // 40 steps, alternating A/B on three boxes: 110.3-114.5 -> 113.5-114.1 Gpx/s with the 512 units / 24 rows above,
            // 115.8-121.8 with 640 units and 32 rows per ll_up0h wave (profiles/r06_coresidency_ab.txt).  On a stream that owns
            // the device the second workgroup is what hides this kernel's own latencies (84.7 -> 74.7 Gpx/s without it).
            // HLMI_LL_D01_PAD_LDS: bytes of unused LDS to add instead (experiments; 0 = two workgroups per CU).
            {
                const int pad = env_int("HLMI_LL_D01_PAD_LDS", emit && partitioned ? -1 : 0);
                constexpr size_t kHalfCuLds = 160 * 1024 / 2;   // gfx950: 160 KB per CU
                if (pad < 0) sh2 = max(sh2, kHalfCuLds + 2048);
                else sh2 += (size_t)pad;
            }
            if (emit) {
                // input read once; outLPyramid[0] (4 B per output pixel), three level-1 planes and K + 1 level-2 planes written
                timing_note_bytes(6.0 * iw * (gm.iy1 - gm.iy0 + 1) + 4.0 * iw * oh + 4.0 * 3.0 * d.w * d.h + 4.0 * (levels + 1) * e.w * e.h);
                D01EArgs ae;
                ae.d = a, ae.outl0 = outl0, ae.oy0 = output->dim[1].min, ae.oh = oh;
                ae.nsx_magic = a.nsx == 1 ? 0u : (unsigned)((1ull << 32) / (unsigned)a.nsx + 1ull);
#define LL_D01E(O0, O1, B)                                                                                                    \
    do {                                                                                                                      \
        if (exch && nt) HLMI_LAUNCH(uc, "ll_down01", st, (ll_down01e<O0, O1, B, true, true>), grid2, block, sh2, ae, gm, lev);  \
        else if (exch) HLMI_LAUNCH(uc, "ll_down01", st, (ll_down01e<O0, O1, B, true, false>), grid2, block, sh2, ae, gm, lev);   \
        else if (nt) HLMI_LAUNCH(uc, "ll_down01", st, (ll_down01e<O0, O1, B, false, true>), grid2, block, sh2, ae, gm, lev);    \
        else HLMI_LAUNCH(uc, "ll_down01", st, (ll_down01e<O0, O1, B, false, false>), grid2, block, sh2, ae, gm, lev);          \
    } while (0)
                switch ((odd0 ? 4 : 0) | (odd1 ? 2 : 0) | (b1 ? 1 : 0)) {
                    case 0: LL_D01E(false, false, false); break;
                    case 1: LL_D01E(false, false, true); break;
                    case 2: LL_D01E(false, true, false); break;
                    case 3: LL_D01E(false, true, true); break;
                    case 4: LL_D01E(true, false, false); break;
                    case 5: LL_D01E(true, false, true); break;
                    case 6: LL_D01E(true, true, false); break;
                    default: LL_D01E(true, true, true); break;
                }
#undef LL_D01E
            } else {
            timing_note_bytes(d0_bytes + 4.0 * (levels + 1) * e.w * e.h);
#define LL_D01(O0, O1, B)                                                                                              \
    do {                                                                                                               \
        if (exch) HLMI_LAUNCH(uc, "ll_down01", st, (ll_down01f<O0, O1, B, true>), grid2, block, sh2, a, gm, lev);      \
        else HLMI_LAUNCH(uc, "ll_down01", st, (ll_down01f<O0, O1, B, false>), grid2, block, sh2, a, gm, lev);          \
    } while (0)
            switch ((odd0 ? 4 : 0) | (odd1 ? 2 : 0) | (b1 ? 1 : 0)) {
                case 0: LL_D01(false, false, false); break;
                case 1: LL_D01(false, false, true); break;
                case 2: LL_D01(false, true, false); break;
                case 3: LL_D01(false, true, true); break;
                case 4: LL_D01(true, false, false); break;
                case 5: LL_D01(true, false, true); break;
                case 6: LL_D01(true, true, false); break;
                default: LL_D01(true, true, true); break;
            }
#undef LL_D01
            }
        } else {
            switch (variant) {
                LL_D0(false, false, false, false) LL_D0(false, false, false, true) LL_D0(false, false, true, false)
                LL_D0(false, false, true, true) LL_D0(false, true, false, false) LL_D0(false, true, false, true)
                LL_D0(false, true, true, false) LL_D0(false, true, true, true) LL_D0(true, false, false, false)
                LL_D0(true, false, false, true) LL_D0(true, false, true, false) LL_D0(true, false, true, true)
                LL_D0(true, true, false, false) LL_D0(true, true, false, true) LL_D0(true, true, true, false)
                LL_D0(true, true, true, true)
            }
        }
#undef LL_D0
        if (r) return r;
    }
    fuse2_out = fuse2;
    // levels 3 and 4 from level 2 in one launch (ll_down_strip2) when the chain below would run ll_down_strip:2 and :3
    // (one stream: 111.6 -> 105.1 us per frame back to back, 123 -> 116.6 for one call + sync; four frame queues 75-77 -> 72-76)
    const bool strip2 = fuse_d2 && S == 4;
    if (strip2) {
        const Level &sl = lv[2], &d = lv[3], &e = lv[4];
        Strip2Args a;
        a.src = sl.g, a.slox = sl.lox, a.sloy = sl.loy, a.sw = sl.w, a.sh = sl.h, a.sws = sl.ws, a.sps = sl.ps;
        a.g1 = d.g, a.so1 = d.lox, a.loy1 = d.loy, a.w1 = d.w, a.h1 = d.h, a.ws1 = d.ws, a.ps1 = d.ps;
        a.g2 = e.g, a.so2 = e.lox, a.loy2 = e.loy, a.w2 = e.w, a.h2 = e.h, a.ws2 = e.ws, a.ps2 = e.ps;
        const bool odd0 = d.odd, odd1 = e.odd;                  // as for ll_down01f: e.odd == (d.lox & 1)
        a.S2 = (odd0 || odd1) ? 62 : 61;
        const int lim = odd1 ? 2 * e.lox - 1 : 2 * e.lox - 2;   // leftmost pair must reach level-(j+2) column so2
        a.Pbase = min(d.lox, lim);
        const int hi1 = d.lox + d.w - 1, hi2 = e.lox + e.w - 1;
        const int x2_first = odd1 ? (a.Pbase + 1) / 2 + 0 : a.Pbase / 2 + 1;
        a.nsx = max((hi2 - x2_first + a.S2) / a.S2, (hi1 - a.Pbase + 2 * a.S2) / (2 * a.S2));
        a.nsy = (e.h + S2_RPU - 1) / S2_RPU;
        a.nunits = (levels + 1) * a.nsx * a.nsy;
        a.ctr = mid ? reinterpret_cast<unsigned *>(wsf + off_ctr) : nullptr;
        timing_note_bytes(4.0 * (levels + 1) * ((double)sl.w * sl.h + (double)d.w * d.h + (double)e.w * e.h));
        dim3 grid((a.nunits + 3) / 4), block(256);
        switch ((odd0 ? 2 : 0) | (odd1 ? 1 : 0)) {
            case 0: HLMI_LAUNCH(uc, "ll_down_strip2:2", st, (ll_down_strip2<false, false>), grid, block, 0, a); break;
            case 1: HLMI_LAUNCH(uc, "ll_down_strip2:2", st, (ll_down_strip2<false, true>), grid, block, 0, a); break;
            case 2: HLMI_LAUNCH(uc, "ll_down_strip2:2", st, (ll_down_strip2<true, false>), grid, block, 0, a); break;
            default: HLMI_LAUNCH(uc, "ll_down_strip2:2", st, (ll_down_strip2<true, true>), grid, block, 0, a); break;
        }
    }
    for (int j = 1; j + 1 < J; j++) {
        if (strip2 && (j == 2 || j == 3)) continue;
        if (j == S && mid && strip2) break;   // ll_mid below makes levels S+1 .. J-1 itself
        if (j == S) {
            const CoarseArgs ca = coarse_args(S);
            long total = 0;
            for (int dl = 1; S + dl < J; dl++) total += (long)(levels + 1) * lv[S + dl].w * lv[S + dl].h;
            const int ntx = (lv[J - 1].w + DM_T - 1) / DM_T, nty = (lv[J - 1].h + DM_T - 1) / DM_T;
            dim3 grid((unsigned)(ntx * nty * (levels + 1))), block(256);
            char nm[32];
            snprintf(nm, sizeof nm, "ll_down_multi:%d", S);
            timing_note_bytes(4.0 * ((double)(levels + 1) * lv[S].w * lv[S].h + (double)total));
            if (J - 1 - S == 4) HLMI_LAUNCH(uc, nm, st, (ll_down_multi<4>), grid, block, 0, ca, ntx, nty);
            else if (J - 1 - S == 3) HLMI_LAUNCH(uc, nm, st, (ll_down_multi<3>), grid, block, 0, ca, ntx, nty);
            else if (J - 1 - S == 2) HLMI_LAUNCH(uc, nm, st, (ll_down_multi<2>), grid, block, 0, ca, ntx, nty);
            else HLMI_LAUNCH(uc, nm, st, (ll_down_multi<1>), grid, block, 0, ca, ntx, nty);
            break;
        }
        if (j == 1 && fuse_d2) continue;   // level 2 came out of ll_down01f
        // enough waves to fill the chip on the big levels, short strips on the small ones
        const StripArgs sa = strip_args(j, 16 * stream_cu_count(ctx.device, nullptr));
        dim3 grid((sa.nunits + 3) / 4), block(256);
        char nm[32];
        snprintf(nm, sizeof nm, "ll_down_strip:%d", j);
        timing_note_bytes(4.0 * (levels + 1) * ((double)lv[j].w * lv[j].h + (double)lv[j + 1].w * lv[j + 1].h));
        if (lv[j + 1].odd) HLMI_LAUNCH(uc, nm, st, (ll_down_strip<true>), grid, block, 0, sa);
        else HLMI_LAUNCH(uc, nm, st, (ll_down_strip<false>), grid, block, 0, sa);
    }
    if (mid && strip2) {
        const CoarseArgs cd = coarse_args(S), cu = coarse_args(SU);
        const int ntxd = (lv[J - 1].w + DM_T - 1) / DM_T, ntyd = (lv[J - 1].h + DM_T - 1) / DM_T, nD = ntxd * ntyd * (levels + 1);
        const int ntxu = (cu.lv[0].rw + UM_T - 1) / UM_T, ntyu = (cu.lv[0].rh + UM_T - 1) / UM_T;
        long total = 0;
        for (int dl = 1; S + dl < J; dl++) total += (long)(levels + 1) * lv[S + dl].w * lv[S + dl].h;
        // ll_down_multi:4's bytes (level 4 read, levels 5-7 written) + ll_up_multi:3's (as below)
        timing_note_bytes(4.0 * ((double)(levels + 1) * lv[S].w * lv[S].h + (double)total) + 4.0 * 4.0 * (double)cu.lv[0].rw * cu.lv[0].rh * 4.0 / 3.0);
        HLMI_LAUNCH(uc, "ll_mid:4", st, (ll_mid<3, 4, 2>), dim3((unsigned)(nD + ntxu * ntyu)), dim3(256), 0, cd, ntxd, ntyd, nD, cu, ntxu,
                    reinterpret_cast<unsigned *>(wsf + off_ctr));
    } else if (SU < J) {
        const CoarseArgs cu = coarse_args(SU);
        const int ntx = (cu.lv[0].rw + UM_T - 1) / UM_T, nty = (cu.lv[0].rh + UM_T - 1) / UM_T;
        dim3 grid((unsigned)(ntx * nty)), block(256);
        char nm[32];
        snprintf(nm, sizeof nm, "ll_up_multi:%d", SU);
        {   // per output pixel of level SU: 2 planes of g + inG read, outG written; coarser levels add 1/3
            const double px = (double)cu.lv[0].rw * cu.lv[0].rh;
            timing_note_bytes(4.0 * 4.0 * px * (SU + 1 < J ? 4.0 / 3.0 : 1.0));
        }
        switch (J - 1 - SU) {
#define LL_UM(T) case T: HLMI_LAUNCH(uc, nm, st, (ll_up_multi<T>), grid, block, 0, cu, ntx); break;
            LL_UM(1) LL_UM(2) LL_UM(3) LL_UM(4) LL_UM(5) LL_UM(6)
#undef LL_UM
        }
    } else {
        const Level &t = lv[J - 1];
        int rw = t.rx1 - t.rx0 + 1, rh = t.ry1 - t.ry0 + 1;
        HLMI_LAUNCH(uc, "ll_top", st, ll_top, dim3((rw + 63) / 64, rh), dim3(64), 0, t.g, t.ws, t.ps, t.lox, t.loy, t.rx0,
                    t.ry0, rw, rh, levels, gm.Km1, t.out);
    }
    for (int j = min(SU, J - 1) - 1; j >= (fuse1 ? (fuse2 ? 3 : 2) : 1); j--) {
        const UpArgs ua = up_args(j);
        char nm[32];
        snprintf(nm, sizeof nm, "ll_up:%d", j);
        // per output: 2 planes of g_j + inG_j read, outG_j written; per coarse pixel: 2 planes of g_{j+1} + outG_{j+1}
        timing_note_bytes(4.0 * (4.0 * ua.rw * ua.rh + 3.0 * (lv[j + 1].rx1 - lv[j + 1].rx0 + 1) * (lv[j + 1].ry1 - lv[j + 1].ry0 + 1)));
        HLMI_LAUNCH(uc, nm, st, ll_up<false>, dim3((ua.rw + 255) / 256, ua.rh), dim3(256), 0, ua);
    }
    fuse1_out = fuse1;
    {
        const Level &c = lv[1];
        dim3 grid((ow + 255) / 256, (oh + 2 * p.RU - 1) / (2 * p.RU)), block(256);
        // input read + output written (u16 x nc channels), 2 selected planes of g_1 read; unfused: + outG_1 read;
        // fused: + inG_1 and, per level-2 pixel, 2 planes of g_2 + outG_2
        const double n1 = (double)(c.rx1 - c.rx0 + 1) * (c.ry1 - c.ry0 + 1), n2 = (double)(lv[2].rx1 - lv[2].rx0 + 1) * (lv[2].ry1 - lv[2].ry0 + 1);
        const double u0_bytes = 2.0 * (3 + nc) * ow * oh + 4.0 * 3.0 * n1 + (fuse1 ? 4.0 * 3.0 * n2 : 0.0);
        if (emit) {
            // input read + output written (u16 x 3 channels), outLPyramid[0] read, three planes of level 1, per level-2 pixel two
            // planes of g_2 + outG_2
            timing_note_bytes(2.0 * (3 + nc) * ow * oh + 4.0 * ow * oh + 4.0 * 3.0 * n1 + 4.0 * 3.0 * n2);
            Up0HArgs ph;
            ph.u = p, ph.outl0 = outl0, ph.l0_ws = gm.ix1 - gm.ix0 + 1;
            ph.fuse2 = fuse2 ? 1 : 0;
            ph.g3 = lv[3].g, ph.out3 = lv[3].out, ph.lox3 = lv[3].lox, ph.loy3 = lv[3].loy, ph.ws3 = lv[3].ws, ph.ps3 = lv[3].ps;
            const size_t sh_h = sizeof(float) * ((size_t)U0_TS * (p.RU + 2) + (size_t)U0H_T2 * (p.RU / 2 + 4)) +
                                (size_t)env_int("HLMI_LL_UP0_PAD_LDS", 0);   // experiment: unused LDS (fewer workgroups per CU)
            if (nt) HLMI_LAUNCH(uc, "ll_up0", st, ll_up0h<true>, grid, block, sh_h, ph, gm);
            else HLMI_LAUNCH(uc, "ll_up0", st, ll_up0h<false>, grid, block, sh_h, ph, gm);
            return 0;
        }
        timing_note_bytes(u0_bytes);
        if (fast) {
            const bool b1 = (beta == 1.0f);
            const size_t sh = lut_sh + (fuse1 ? ((lut_lds && (nlut & 1)) ? 4 : 0) + sizeof(float) * U0_TS * (p.RU + 2) : 0);
#define LL_U0(L, B, F) HLMI_LAUNCH(uc, "ll_up0", st, (ll_up0f<L, B, F>), grid, block, sh, p, gm)
            if (fuse1) {
                if (lut_lds && b1) LL_U0(true, true, true);
                else if (lut_lds) LL_U0(true, false, true);
                else if (b1) LL_U0(false, true, true);
                else LL_U0(false, false, true);
            } else {
                if (lut_lds && b1) LL_U0(true, true, false);
                else if (lut_lds) LL_U0(true, false, false);
                else if (b1) LL_U0(false, true, false);
                else LL_U0(false, false, false);
            }
#undef LL_U0
        } else if (vec) {
            if (lut_lds) HLMI_LAUNCH(uc, "ll_up0", st, (ll_up0<true, true>), grid, block, lut_sh, p, gm);
            else HLMI_LAUNCH(uc, "ll_up0", st, (ll_up0<true, false>), grid, block, lut_sh, p, gm);
        } else {
            if (lut_lds) HLMI_LAUNCH(uc, "ll_up0", st, (ll_up0<false, true>), grid, block, lut_sh, p, gm);
            else HLMI_LAUNCH(uc, "ll_up0", st, (ll_up0<false, false>), grid, block, lut_sh, p, gm);
        }
    }
    return 0;
    };  // enqueue

    t_dbg_K = levels, t_dbg_Km1 = gm.Km1;
    t_dbg_emit = emit;   // decided outside `enqueue`: a graph replay leaves it right too
    // ---- replay / capture / eager
    GraphEntry *ge = nullptr;
    const bool graphs = env_int("HLMI_LL_GRAPH", 0) && !stream_is_special(st) && !timing_enabled() && !env_int("HLMI_LL_NO_LUT_CACHE", 0);
    bool capture = false;
    GraphKey key;
    memset(&key, 0, sizeof key);   // padding bytes too: keys are compared with memcmp
    if (graphs) {
        key.device = ctx.device, key.stream = st, key.in = din, key.out = dout, key.ws = ws, key.lut = lut;
        for (int d = 0; d < 3; d++) {
            key.idim[d][0] = input->dim[d].min, key.idim[d][1] = input->dim[d].extent, key.idim[d][2] = input->dim[d].stride;
            key.odim[d][0] = output->dim[d].min, key.odim[d][1] = output->dim[d].extent, key.odim[d][2] = output->dim[d].stride;
        }
        key.levels = levels;
        memcpy(&key.alpha_bits, &alpha, 4);
        memcpy(&key.beta_bits, &beta, 4);
        key.env_sig = ll_env_signature();
        std::lock_guard<std::mutex> lock(g_graph_mu);
        for (auto &e : g_graphs) {
            if (e.key == key) {
                ge = &e;
                break;
            }
        }
        if (ge && ge->exec) {
            ge->used = ++g_graph_clock;
            t_dbg_out1_pending = ge->out1_pending, t_dbg_out2_pending = ge->out2_pending;
            HLMI_HIP(uc, hipGraphLaunch(ge->exec, st));   // under the lock: an entry cannot be evicted while it is launched
            mark_output_written(output);
            return 0;
        }
        if (!ge) {   // first sight of this key: remember it, run eagerly
            if (g_graphs.size() >= 64) {
                size_t victim = 0;
                for (size_t i = 1; i < g_graphs.size(); i++) if (g_graphs[i].used < g_graphs[victim].used) victim = i;
                GraphEntry &v = g_graphs[victim];
                if (v.exec) {   // may still be executing on its stream
                    (void)hipStreamSynchronize(v.key.stream);
                    (void)hipGraphExecDestroy(v.exec);
                    (void)hipGraphDestroy(v.graph);
                    (void)hipGetLastError();
                }
                g_graphs.erase(g_graphs.begin() + victim);
            }
            GraphEntry e;
            e.key = key, e.used = ++g_graph_clock;
            g_graphs.push_back(e);
        } else if (!ge->failed) {
            capture = true;   // second sight
        }
    }
    if (capture) {
        // the entry may move when another thread pushes (`ge` is not used beyond this point): capture into locals, publish
        // under the lock by key
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        static thread_local hipStream_t t_cap[64] = {};   // per thread and device: used for captures only, never executes
        hipStream_t &cap = t_cap[ctx.device & 63];
        bool ok = cap != nullptr || hipStreamCreateWithFlags(&cap, hipStreamNonBlocking) == hipSuccess;
        int er = 0;
        if (ok) ok = hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal) == hipSuccess;
        if (ok) {
            const hipStream_t real = st;
            st = cap;            // `enqueue` launches on `st`
            er = enqueue();
            st = real;
            ok = hipStreamEndCapture(cap, &graph) == hipSuccess && graph != nullptr && er == 0;
        }
        if (ok) ok = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess;
        if (!ok) {
            (void)hipGetLastError();
            if (exec) (void)hipGraphExecDestroy(exec);
            if (graph) (void)hipGraphDestroy(graph);
            exec = nullptr, graph = nullptr;
        }
        {
            std::lock_guard<std::mutex> lock(g_graph_mu);
            for (auto &e : g_graphs) {
                if (e.key == key) {
                    if (ok && !e.exec) {
                        e.graph = graph, e.exec = exec, e.out1_pending = fuse1_out, e.out2_pending = fuse2_out, e.used = ++g_graph_clock;
                        graph = nullptr, exec = nullptr;
                    } else if (!ok) {
                        e.failed = true;
                    }
                    break;
                }
            }
        }
        if (exec) {   // another thread published the same key meanwhile (or the entry was evicted): use ours once, drop it
            HLMI_HIP(uc, hipGraphLaunch(exec, st));
            (void)hipStreamSynchronize(st);
            (void)hipGraphExecDestroy(exec);
            (void)hipGraphDestroy(graph);
            t_dbg_out1_pending = fuse1_out, t_dbg_out2_pending = fuse2_out;
            mark_output_written(output);
            return 0;
        }
        if (ok) {
            std::lock_guard<std::mutex> lock(g_graph_mu);
            for (auto &e : g_graphs) {
                if (e.key == key && e.exec) {
                    t_dbg_out1_pending = e.out1_pending, t_dbg_out2_pending = e.out2_pending;
                    HLMI_HIP(uc, hipGraphLaunch(e.exec, st));
                    mark_output_written(output);
                    return 0;
                }
            }
        }
        if (er) return er;   // a launch error inside the capture: reported as the eager path would
        // capture refused: fall through to the eager path
    }
    if ((r = enqueue())) return r;
    t_dbg_out1_pending = fuse1_out, t_dbg_out2_pending = fuse2_out;
    mark_output_written(output);
    return 0;
}

extern "C" int local_laplacian_argv(void **a) {
    return local_laplacian((halide_buffer_t *)a[0], *(int32_t *)a[1], *(float *)a[2], *(float *)a[3],
                           (halide_buffer_t *)a[4]);
}
extern "C" const halide_filter_metadata_t *local_laplacian_metadata(void) { return &ll_md; }
extern "C" int local_laplacian_auto_schedule(halide_buffer_t *input, int32_t levels, float alpha, float beta,
                                             halide_buffer_t *output) {
    return local_laplacian(input, levels, alpha, beta, output);
}

// Test hook: runs the DPP wave-shift probe on the current device; 1 = wave_shr:1 / wave_shl:1 behave as the
// strip kernels assume, 0 = they do not, < 0 = HIP error.
extern "C" int hlmi_debug_ll_probe(unsigned long long *out32) {
#if HLMI_LL_PROBE
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_probe), sizeof(unsigned long long) * 32) != hipSuccess) return -1;
    unsigned long long zero[32] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_probe), zero, sizeof zero) != hipSuccess) return -1;
    return 1;
#else
    (void)out32;
    return 0;
#endif
}

extern "C" int hlmi_debug_ll_residency(unsigned long long *out16) {
#if HLMI_LL_RESIDENCY
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_res_hist), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
    unsigned long long zero[16] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_res_hist), zero, sizeof zero) != hipSuccess) return -1;
    return 1;
#else
    (void)out16;
    return 0;
#endif
}

extern "C" int hlmi_debug_dpp_probe(void) {
    int *flag = nullptr, h = 0;
    if (hipMalloc(&flag, sizeof(int)) != hipSuccess) return -1;
    if (hipMemset(flag, 0, sizeof(int)) != hipSuccess) return -1;
    hipLaunchKernelGGL(ll_dpp_probe, dim3(1), dim3(64), 0, 0, flag);
    if (hipMemcpy(&h, flag, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    (void)hipFree(flag);
    return h;
}

// Test hook: div3_by against the correctly-rounded `/` on `count` operand pairs (host pointers); returns the
// number of quotients that differ bitwise (0 expected), < 0 = HIP error.
extern "C" int hlmi_debug_div3_check(const float *n, const float *d, int count) {
    float *dn = nullptr, *dd = nullptr;
    int *bad = nullptr, h = -1;
    if (hipMalloc(&dn, sizeof(float) * count) != hipSuccess || hipMalloc(&dd, sizeof(float) * count) != hipSuccess ||
        hipMalloc(&bad, sizeof(int)) != hipSuccess) {
        return -1;
    }
    if (hipMemcpy(dn, n, sizeof(float) * count, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(dd, d, sizeof(float) * count, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemset(bad, 0, sizeof(int)) != hipSuccess) {
        return -1;
    }
    hipLaunchKernelGGL(ll_div3_check, dim3((count + 255) / 256), dim3(256), 0, 0, dn, dd, count, bad);
    if (hipMemcpy(&h, bad, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) h = -1;
    (void)hipFree(dn), (void)hipFree(dd), (void)hipFree(bad);
    return h;
}

// Test hook (tests/ only; not part of the reference ABI): copies outGPyramid[level] of the calling thread's
// LAST local_laplacian call, restricted to R_level, to `dst` (row-major, rw x rh floats); returns 0, or -1.
extern "C" int hlmi_debug_local_laplacian_outg(int level, float *dst, int cap_floats, int *rw_out, int *rh_out) {
    auto dbg_up_args = [](const Level &a, const Level &c) {
        UpArgs u;
        u.g = a.g, u.ws = a.ws, u.ps = a.ps, u.lox = a.lox, u.loy = a.loy, u.gc = c.g, u.outc = c.out, u.cws = c.ws, u.cps = c.ps;
        u.clox = c.lox, u.cloy = c.loy, u.rx0 = a.rx0, u.ry0 = a.ry0, u.rw = a.rx1 - a.rx0 + 1, u.rh = a.ry1 - a.ry0 + 1;
        u.K = t_dbg_K, u.Km1 = t_dbg_Km1, u.out = a.out;
        return u;
    };
    if (level < 1 || level >= J || !t_dbg_lv[level].out) return -1;
    const Level &L = t_dbg_lv[level];
    int rw = L.rx1 - L.rx0 + 1, rh = L.ry1 - L.ry0 + 1;
    if (rw_out) *rw_out = rw;
    if (rh_out) *rh_out = rh;
    if (!dst) return 0;
    if ((long)rw * rh > cap_floats) return -1;
    if ((level == 1 || level == 2) && t_dbg_out2_pending) {
        // ll_up0h kept outGPyramid[2] in LDS tiles: produce the plane with the stand-alone kernel (levels 2, 3 and outGPyramid[3] are
        // still in the arena); level 1's own stand-alone collapse below reads it
        const Level &a = t_dbg_lv[2], &c = t_dbg_lv[3];
        const int rw2 = a.rx1 - a.rx0 + 1, rh2 = a.ry1 - a.ry0 + 1;
        hipLaunchKernelGGL(ll_up<false>, dim3((rw2 + 255) / 256, rh2), dim3(256), 0, t_dbg_stream, dbg_up_args(a, c));
        if (hipGetLastError() != hipSuccess) return -1;
        t_dbg_out2_pending = false;
    }
    if (level == 1 && t_dbg_out1_pending) {
        // the fused ll_up0f / ll_up0h kept outGPyramid[1] in LDS: produce the plane now with the stand-alone kernel (its inputs are
        // still in the arena) so that the tests can compare every level
        const Level &a = t_dbg_lv[1], &c = t_dbg_lv[2];
        if (t_dbg_emit) {   // level 1 holds its three planes only (ll_down01e)
            hipLaunchKernelGGL(ll_up<true>, dim3((rw + 255) / 256, rh), dim3(256), 0, t_dbg_stream, dbg_up_args(a, c));
        } else {
            hipLaunchKernelGGL(ll_up<false>, dim3((rw + 255) / 256, rh), dim3(256), 0, t_dbg_stream, dbg_up_args(a, c));
        }
        if (hipGetLastError() != hipSuccess) return -1;
        t_dbg_out1_pending = false;
    }
    if (hipStreamSynchronize(t_dbg_stream) != hipSuccess) return -1;
    const float *src = L.out + (size_t)(L.ry0 - L.loy) * L.ws + (L.rx0 - L.lox);
    if (hipMemcpy2D(dst, sizeof(float) * rw, src, sizeof(float) * L.ws, sizeof(float) * rw, rh, hipMemcpyDeviceToHost) !=
        hipSuccess) {
        return -1;
    }
    return 0;
}
