// iir_blur.hip — gfx950 implementation of the reference's iir_blur AOT pipeline (first-order IIR low pass down/up the
// columns, then along the rows).  Algorithm: /root/reference/apps/iir_blur/iir_blur_generator.cpp:13-31,
// 146-156; boundary: `int iir_blur(halide_buffer_t *input, float alpha, halide_buffer_t *output)`, f32 [W,H,C] planar.
// The reference pins 1536 x 2560 x 3 (:158-163); this entry point takes any extents (superset) with mins 0.
//
// The scans are sequential BY DEFINITION (float recurrences do not re-associate), so the only parallelism is across
// columns and channels: one workgroup owns 64 adjacent columns of one channel, and ONE of its waves (the scanner) runs the
// recurrence — a dependent multiply + add per row, ~16 cycles.  With a single wave doing everything (the first version:
// 0.74 ms) each step also paid its share of a ~1 us HBM round trip, because one wave cannot keep more than 63 vector-memory
// operations in flight.  So the memory traffic moves to eight helper waves and the scanner only ever touches LDS:
//   tile p = 64 rows x 64 columns, four LDS slots in rotation:  slot (p+2)%4 is being filled with tile p+2 (its rows were
//   requested four iterations earlier and waited for in registers), slot (p+1)%4 is complete and is READ by the scanner while
//   it scans slot p%4 in place — the reads ride in the issue slots the dependent multiply-add chain leaves empty
//   (reading a tile's 64 values BEFORE its chain, as round 1 did, cost as much again per tile) — and slot (p-1)%4 is
//   written out.  The steady-state scanner loop is now nothing but the chain: 128 dependent VALU operations and 32 LDS
//   instructions per tile, 7.8 ns per row against the 7.4 ns two dependent operations cost a lone wave
//   (scripts/ubench/valu_dep.hip).  Floor of this decomposition: 2 directions x (2560 + 1536) rows x 7.4 ns = 61 us.
//   forward   b = (1-a) b + a in(x, y), top to bottom; tiles go to a scratch plane as 64-float rows
//   backward  b = (1-a) b + a scratch(x, y), bottom to top; tiles are written out TRANSPOSED (the generator transposes
//             after each column blur, :31): lane = y, 64 consecutive floats per store
// Launched twice (columns of the input, then columns of the transposed intermediate = rows of the input).
#include "hlmi_device_math.h"
#include "hlmi_internal.h"

#include <type_traits>

using namespace hlmi;

namespace {

#ifndef HLMI_IIR_PROBE
#define HLMI_IIR_PROBE 0
#endif
#if HLMI_IIR_PROBE
__device__ unsigned long long g_iprobe[32];
#define IP_T(v) const unsigned long long v = wall_clock64()
#define IP_ACC(a, d) a += (d)
#else
#define IP_T(v)
#define IP_ACC(a, d)
#endif
constexpr int TR = 64;                    // tile rows
// LDS tile layout: groups of 4 rows, a lane's 4 values of a group contiguous (one ds_*_b128), groups 260 floats apart:
//   at(r, l) = (r / 4) * TG + 4 l + (r % 4).  The scanner — the critical path: it may have only 15 LDS operations in
// flight (lgkmcnt), and with one dword per row it spent more time waiting on that counter than on the recurrence — moves
// a group per instruction (32 LDS instructions per tile instead of 128); the transposed reads of the backward pass
// (lane = row, fixed column) fall on bank (4 j + lane) mod 32: conflict-free because TG = 4 (mod 32).
constexpr int TG = 260, TSZ = (TR / 4) * TG;
__device__ __forceinline__ int at(int r, int l) { return (r >> 2) * TG + 4 * l + (r & 3); }
constexpr int NHELP = 8, NLOAD = 4, LR = TR / NLOAD, SR = TR / (NHELP - NLOAD);   // helper waves: loaders / storers and the tile rows each moves
constexpr int NSET = 4;                   // tiles a helper has requested and not yet put into LDS (+1 being filled)
constexpr int SLOTS = 8;                  // LDS tile slots (133 KB): four in rotation + the tiles kept for the turn
constexpr int KEEP = 7;                   // forward tiles that stay in LDS for the backward pass (see `turn`)

// src: [C][Hd][Wd] (row stride s_sy, plane stride s_sc); scratch: same shape, dense; dst: [C][Wd][Hd] (d_sy, d_sc)
__global__ __launch_bounds__(64 * (1 + NHELP)) void iir_cols_T(const float *__restrict__ src, long s_sy, long s_sc, int Wd, int Hd,
                                                               float alpha, float *__restrict__ scratch, float *__restrict__ dst,
                                                               long d_sy, long d_sc) {
    extern __shared__ __attribute__((aligned(16))) float tiles_raw[];   // [SLOTS][TSZ]: 133 KB, above the static limit
    auto tiles = reinterpret_cast<float (*)[TSZ]>(tiles_raw);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int x0 = blockIdx.x * 64, c = blockIdx.y;
    const int x = min(x0 + lane, Wd - 1);                    // lanes past the edge shadow the last column (never stored)
    const float c1 = 1.0f - alpha;   // (1 - alpha) b + alpha in: the first product is the one fused under the fma canon (dev::mad), alpha in
                                     // stays a multiply — made by the helper waves, off the scanner's dependent chain
    const float *sbase = src + (long)c * s_sc, *tbase = scratch + ((long)c * Hd) * Wd;   // uniform; the loaders add xoff
    const uint32_t xoff = 4u * (uint32_t)x;
    float *t = scratch + ((long)c * Hd) * Wd + x;
    float *dcol = dst + (long)c * d_sc;
    const int NT = (Hd + TR - 1) / TR;
    float b = 0.0f;                                          // scanner: the recurrence state
    // THE TURN.  The backward pass starts where the forward pass ends, so its first tiles are the forward pass's last ones —
    // which are still in LDS.  Sending them through the scratch plane put a store round trip (the last tiles must be
    // visible) and a load round trip (~3 us before tile 0 is back) between the two passes of every launch.  With `turn`
    // (whole tiles only) the last KEEP forward tiles are not written out: the helpers reverse them in place (row r -> step
    // 63 - r, times alpha: what a loader would have put there) between two barriers, the scanner goes straight on, and the
    // loaders' first loads (tile KEEP) have KEEP - 2 tiles of recurrence to arrive.
    const bool turn = (Hd % TR) == 0 && NT >= 16;

    // one pass over the column: tile p (processing order) holds steps 0..n-1; forward: step st = row 64p + st of `s`,
    // backward: row Hd-1 - 64p - st of the scratch plane.  The scanner and the helpers run DIFFERENT loops that meet at the
    // same barriers, and the helpers' steady-state iteration is straight-line code (clamped indices instead of guards):
    // any branch around the loads makes the compiler drain all outstanding loads at the join, which is the latency
    // this structure exists to hide.
    auto pass = [&](auto back_tag) {
        constexpr bool BACK = decltype(back_tag)::value;
        // slot of tile p (processing order).  With the turn the backward pass walks the slots downwards from the one the
        // forward pass ended in: backward tile p IS forward tile NT-1-p, found where the forward pass left it
        auto sl = [&](int p) { return (BACK && turn) ? ((NT - 1 - p) & (SLOTS - 1)) : (p & (SLOTS - 1)); };
        auto rows_of = [&](int p) { return min(TR, Hd - p * TR); };
        auto row_at = [&](int p, int st) { return BACK ? Hd - 1 - p * TR - st : p * TR + st; };
        // the turn, helpers' side: wave hw of the eight takes groups 2 hw, 2 hw + 1 (rows 8 hw .. 8 hw + 7) of each kept tile,
        // everybody reads, barrier, everybody writes the mirrored groups: step st = 63 - r holds alpha * forward(r), the
        // pass's very first step (tile 0, step 0) the raw value — exactly what fill() makes of the scratch rows
        auto reverse_kept = [&](int hw) {
            const int g0 = 2 * hw, g1 = 2 * hw + 1;
            float4 ka[KEEP], kb[KEEP];
#pragma unroll
            for (int q = 0; q < KEEP; q++) {
                const float4 *tl = reinterpret_cast<const float4 *>(tiles[sl(q)] + 4 * lane);
                ka[q] = tl[g0 * (TG / 4)], kb[q] = tl[g1 * (TG / 4)];
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < KEEP; q++) {
                float4 *tl = reinterpret_cast<float4 *>(tiles[sl(q)] + 4 * lane);
                float4 wa, wb;
                wa.x = alpha * ka[q].w, wa.y = alpha * ka[q].z, wa.z = alpha * ka[q].y, wa.w = alpha * ka[q].x;
                wb.x = (q == 0 && g1 == TR / 4 - 1) ? kb[q].w : alpha * kb[q].w;
                wb.y = alpha * kb[q].z, wb.z = alpha * kb[q].y, wb.w = alpha * kb[q].x;
                tl[(TR / 4 - 1 - g0) * (TG / 4)] = wa;
                tl[(TR / 4 - 1 - g1) * (TG / 4)] = wb;
            }
        };
        if (wave == 0) {
            // ---- the scanner: b = c1 * b + (alpha * in), in place in the tile's LDS slot
            auto slow = [&](int p) {                         // first tile (its first row is taken as it is, :21, :26-28) and a short last tile
                float *tl = tiles[sl(p)];
                const int n = rows_of(p);
                int st = 0;
                if (p == 0) {
                    b = tl[at(0, lane)];
                    st = 1;
                }
                for (; st < n; st++) {
                    b = dev::mad(c1, b, tl[at(st, lane)]);
                    tl[at(st, lane)] = b;
                }
            };
            // a full tile whose 64 inputs are already in registers; the NEXT tile's inputs are read meanwhile
            auto fast = [&](int p, const float4 (&ucur)[TR / 4], float4 (&unxt)[TR / 4]) {
                float4 *tl = reinterpret_cast<float4 *>(tiles[sl(p)] + 4 * lane);
                const float4 *tn = reinterpret_cast<const float4 *>(tiles[sl(p + 1)] + 4 * lane);
#pragma unroll
                for (int gq = 0; gq < TR / 4; gq++) {
                    float4 r;
                    b = dev::mad(c1, b, ucur[gq].x), r.x = b;
                    b = dev::mad(c1, b, ucur[gq].y), r.y = b;
                    b = dev::mad(c1, b, ucur[gq].z), r.z = b;
                    b = dev::mad(c1, b, ucur[gq].w), r.w = b;
                    tl[gq * (TG / 4)] = r;
                    unxt[gq] = tn[gq * (TG / 4)];
                }
            };
            // the first and the last tile of a pass, when full: all 64 inputs in registers before the chain starts (one LDS
            // round trip), as in the steady state — the row-at-a-time loop of slow() pays that round trip per row (~45 ns
            // against 7.4), 2 x 2.4 us per pass
            auto first_full = [&]() {
                float4 *tl = reinterpret_cast<float4 *>(tiles[sl(0)] + 4 * lane);
                float4 u[TR / 4];
#pragma unroll
                for (int gq = 0; gq < TR / 4; gq++) u[gq] = tl[gq * (TG / 4)];
#pragma unroll
                for (int gq = 0; gq < TR / 4; gq++) {
                    float4 r;
                    if (gq == 0) b = u[0].x;                 // the pass's first row is taken as it is (:21, :26-28)
                    else b = dev::mad(c1, b, u[gq].x);
                    r.x = b;
                    b = dev::mad(c1, b, u[gq].y), r.y = b;
                    b = dev::mad(c1, b, u[gq].z), r.z = b;
                    b = dev::mad(c1, b, u[gq].w), r.w = b;
                    tl[gq * (TG / 4)] = r;
                }
            };
            auto last_full = [&](int p, const float4 (&ucur)[TR / 4]) {
                float4 *tl = reinterpret_cast<float4 *>(tiles[sl(p)] + 4 * lane);
#pragma unroll
                for (int gq = 0; gq < TR / 4; gq++) {
                    float4 r;
                    b = dev::mad(c1, b, ucur[gq].x), r.x = b;
                    b = dev::mad(c1, b, ucur[gq].y), r.y = b;
                    b = dev::mad(c1, b, ucur[gq].z), r.z = b;
                    b = dev::mad(c1, b, ucur[gq].w), r.w = b;
                    tl[gq * (TG / 4)] = r;
                }
            };
            float4 ua[TR / 4], ub[TR / 4];
            IP_T(w0);
            if (BACK && turn) __syncthreads();               // the helpers' reversal of the kept tiles (read | write)
            __syncthreads();
            IP_T(w1);
            if (rows_of(0) == TR) first_full();
            else slow(0);
            if (NT > 2) {                                    // tile 1 is full and was filled before the barrier above
                const float4 *tn = reinterpret_cast<const float4 *>(tiles[sl(1)] + 4 * lane);
#pragma unroll
                for (int gq = 0; gq < TR / 4; gq++) ua[gq] = tn[gq * (TG / 4)];
            }
            __syncthreads();
            IP_T(w2);
            int p = 1;
#if HLMI_IIR_PROBE
            unsigned long long pr_chain = 0, pr_bar = 0;
#endif
            while (p <= NT - 2) {                            // tiles 1 .. NT-2 are full
                IP_T(q0);
                fast(p, ua, ub);
                IP_T(q1);
                __syncthreads();
                IP_T(q2);
                IP_ACC(pr_chain, q1 - q0); IP_ACC(pr_bar, q2 - q1);
                if (++p > NT - 2) break;
                IP_T(q3);
                fast(p, ub, ua);
                IP_T(q4);
                __syncthreads();
                IP_T(q5);
                IP_ACC(pr_chain, q4 - q3); IP_ACC(pr_bar, q5 - q4);
                ++p;
            }
#if HLMI_IIR_PROBE
            if (lane == 0) atomicAdd(&g_iprobe[0], pr_chain), atomicAdd(&g_iprobe[1], pr_bar), atomicAdd(&g_iprobe[2], 1ull);
#endif
            IP_T(w3);
            if (NT > 1) {
                if (NT > 2 && rows_of(NT - 1) == TR) {       // its inputs were read during tile NT-2: odd tiles sit in ua
                    if ((NT - 1) & 1) last_full(NT - 1, ua);
                    else last_full(NT - 1, ub);
                } else {
                    slow(NT - 1);
                }
                __syncthreads();
            }
            __syncthreads();
#if HLMI_IIR_PROBE
            {
                IP_T(w4);
                const int o = BACK ? 21 : 16;
                if (lane == 0) atomicAdd(&g_iprobe[o], w1 - w0), atomicAdd(&g_iprobe[o + 1], w2 - w1), atomicAdd(&g_iprobe[o + 2], w3 - w2),
                               atomicAdd(&g_iprobe[o + 3], w4 - w3), atomicAdd(&g_iprobe[o + 4], 1ull);
            }
#endif
            return;
        }
        // ---- the helpers: waves 1..NLOAD only LOAD (global -> registers -> LDS), the others only STORE (LDS -> global).  A wave
        // that does both has to wait for loads it issued four tiles ago with stores of the last tiles still in flight, and the
        // one in-order vmcnt counter of gfx9 turns that into "wait for (almost) everything": the memory round trip came back
        // into every iteration (~0.9 us per tile against a 0.33 us recurrence).  Loaders wait for loads only; storers never wait.
        if (wave <= NLOAD) {
            const int l0 = (wave - 1) * LR;                  // first tile row this loader moves
            float set[NSET][LR];                             // rows of NSET tiles in flight
            auto issue = [&](int p, float (&v)[LR]) {        // p past the end re-reads the last tile, rows past a short tile its last row
                const int pc = min(p, NT - 1), last = rows_of(pc) - 1;
                // the rows are the same for the whole wave: a uniform row pointer (scalar arithmetic) + the lane's 32-bit byte
                // offset is ONE vector instruction per load, and the pointer ADVANCES by one row stride per load (a 64-bit
                // product per row made an issue() 330 ns of scalar arithmetic — 2 us of every pass's start-up)
                const long rs = BACK ? -(long)Wd : s_sy;
                const float *rowp = BACK ? tbase + (long)row_at(pc, min(l0, last)) * Wd : sbase + (long)row_at(pc, min(l0, last)) * s_sy;
#pragma unroll
                for (int i = 0; i < LR; i++) {
                    v[i] = *(const float *)((const char *)rowp + xoff);
                    rowp += (l0 + i < last) ? rs : 0;        // rows past a short tile repeat its last row
                }
            };
            // the loaders also do the recurrence's independent product: LDS holds alpha * in (the pass's very first row stays
            // raw).  Filling a slot for a tile past the end is harmless (that slot's tile has been written out already) except in a
            // forward pass with the turn, whose last tiles stay: iter_tail guards it.
            auto fill = [&](int p, const float (&v)[LR]) {
                float4 *tl = reinterpret_cast<float4 *>(tiles[sl(p)] + at(l0, lane));   // l0 is a multiple of 4
#pragma unroll
                for (int gq = 0; gq < LR / 4; gq++) {
                    float4 w;
                    w.x = (p == 0 && l0 + 4 * gq == 0) ? v[4 * gq] : alpha * v[4 * gq];
                    w.y = alpha * v[4 * gq + 1], w.z = alpha * v[4 * gq + 2], w.w = alpha * v[4 * gq + 3];
                    tl[gq * (TG / 4)] = w;
                }
            };
#if HLMI_IIR_PROBE
            unsigned long long pr_fill = 0, pr_issue = 0, pr_lbar = 0;
#endif
            auto iter = [&](int p, auto ph_tag) {            // steady state, 1 <= p: no branches
                constexpr int PH = decltype(ph_tag)::value;  // p % NSET: the register set indices must be compile-time
                IP_T(q0);
                fill(p + 2, set[(PH + 2) % NSET]);
                IP_T(q1);
                issue(p + 2 + NSET, set[(PH + 2) % NSET]);
                IP_T(q2);
                __syncthreads();
                IP_T(q3);
                IP_ACC(pr_fill, q1 - q0); IP_ACC(pr_issue, q2 - q1); IP_ACC(pr_lbar, q3 - q2);
            };
            // the last iterations of a pass: a forward pass with the turn must not fill tiles past the end (the slots hold
            // the tiles it keeps); the branch costs the loads' latency hiding, which no longer matters there
            auto iter_tail = [&](int p, auto ph_tag) {
                constexpr int PH = decltype(ph_tag)::value;
                if (BACK || !turn || p + 2 < NT) fill(p + 2, set[(PH + 2) % NSET]);
                if (p + 2 + NSET < NT) issue(p + 2 + NSET, set[(PH + 2) % NSET]);   // nothing in flight when the pass ends
                __syncthreads();
            };
            static_assert(NSET == 4 && SLOTS == 8 && KEEP == 7, "the unrolled tile loop below names the phases; tile t travels in set t % 4");
            int p = 1;
            if (BACK && turn) {
                // tiles 0 .. KEEP-1 are in LDS already; the first loads are tiles KEEP .. KEEP+3
                // the reversal first — the scanner waits for it —, then the loads: they have KEEP - 2 tiles to arrive.  (Only
                // three sets before the next barrier: the 64th load in flight would stall this wave — vmcnt counts to 63 —
                // until the first one is back, with the scanner waiting at the barrier behind it.)
                reverse_kept(wave - 1);
                __syncthreads();
                issue(KEEP, set[3]);
                issue(KEEP + 1, set[0]);
                issue(KEEP + 2, set[1]);
                __syncthreads();
                issue(KEEP + 3, set[2]);
                for (; p < KEEP - 2; p++) __syncthreads();   // iterations whose tile p + 2 needs no fill
            } else {
                issue(0, set[0]);
                issue(1, set[1]);
                issue(2, set[2]);
                issue(3, set[3]);
                fill(0, set[0]);
                fill(1, set[1]);
                __syncthreads();                             // the scanner starts on tile 0 as early as possible
                issue(4, set[0]);
                issue(5, set[1]);
                fill(2, set[2]);
                issue(6, set[2]);
                __syncthreads();
            }
            // p = 1 or KEEP - 2 = 5: p % 4 == 1 either way
            for (; p + 5 < NT; p += 4) {
                iter(p, std::integral_constant<int, 1>{});
                iter(p + 1, std::integral_constant<int, 2>{});
                iter(p + 2, std::integral_constant<int, 3>{});
                iter(p + 3, std::integral_constant<int, 0>{});
            }
            if (p < NT) iter_tail(p++, std::integral_constant<int, 1>{});
            if (p < NT) iter_tail(p++, std::integral_constant<int, 2>{});
            if (p < NT) iter_tail(p++, std::integral_constant<int, 3>{});
            if (p < NT) iter_tail(p++, std::integral_constant<int, 0>{});
            if (p < NT) iter_tail(p++, std::integral_constant<int, 1>{});
            __syncthreads();                                 // the storers' last tile
#if HLMI_IIR_PROBE
            if (lane == 0 && wave == 1) atomicAdd(&g_iprobe[4], pr_fill), atomicAdd(&g_iprobe[5], pr_issue), atomicAdd(&g_iprobe[6], pr_lbar), atomicAdd(&g_iprobe[7], 1ull);
#endif
            return;
        }
        const int h0 = (wave - 1 - NLOAD) * SR;              // storer: first tile row (forward) / tile column (backward) it moves
        auto drain_full = [&](int p) {                       // tile p (64 rows) leaves LDS
            const float *tl = tiles[sl(p)];
#pragma unroll
            for (int i = 0; i < SR; i++) {
                if (!BACK) {
                    t[(long)row_at(p, h0 + i) * Wd] = tl[at(h0 + i, lane)];
                } else {
                    // transposed: dst[c][x0 + j][y], y = the row of step `lane` (64 consecutive floats, descending with the
                    // lane); columns past the edge repeat the last one (same value to the same address)
                    const int j = min(h0 + i, Wd - 1 - x0);
                    dcol[(long)(x0 + j) * d_sy + row_at(p, lane)] = tl[at(lane, j)];
                }
            }
        };
        auto drain_last = [&](int p) {                       // the (possibly short) last tile
            const float *tl = tiles[sl(p)];
            const int n = rows_of(p);
#pragma unroll
            for (int i = 0; i < SR; i++) {
                if (!BACK) {
                    if (h0 + i < n) t[(long)row_at(p, h0 + i) * Wd] = tl[at(h0 + i, lane)];
                } else {
                    const int j = min(h0 + i, Wd - 1 - x0);
                    if (lane < n) dcol[(long)(x0 + j) * d_sy + row_at(p, lane)] = tl[at(lane, j)];
                }
            }
        };
        if (BACK && turn) reverse_kept(wave - 1);
        __syncthreads();
        __syncthreads();                                     // iteration 0: nothing to write out yet
#if HLMI_IIR_PROBE
        unsigned long long pr_drain = 0, pr_sbar = 0;
#endif
        for (int p = 1; p < NT; p++) {
            IP_T(q0);
            if (BACK || !turn || p - 1 < NT - KEEP) drain_full(p - 1);   // a forward pass with the turn keeps its last tiles
            IP_T(q1);
            __syncthreads();
            IP_T(q2);
            IP_ACC(pr_drain, q1 - q0); IP_ACC(pr_sbar, q2 - q1);
        }
#if HLMI_IIR_PROBE
        if (lane == 0 && wave == NLOAD + 1) atomicAdd(&g_iprobe[8], pr_drain), atomicAdd(&g_iprobe[9], pr_sbar), atomicAdd(&g_iprobe[10], 1ull);
#endif
        if (BACK || !turn) drain_last(NT - 1);
        __syncthreads();                                     // the scratch rows are visible to the whole workgroup
    };
    pass(std::false_type{});
    pass(std::true_type{});
}

const int64_t e0 = 0, ew = 1536, eh = 2560, ec = 3;
const int64_t *const est[6] = {&e0, &ew, &e0, &eh, &e0, &ec};
const halide_scalar_value_t est_alpha = [] { halide_scalar_value_t v{}; v.u.f32 = 0.1f; return v; }();
const halide_type_t ty_f32 = {(decltype(halide_type_t::code))2, 32, 0};
// estimates: generator :166-173
const halide_filter_argument_t ib_args[3] = {
    {"input", halide_argument_kind_input_buffer, 3, ty_f32, nullptr, nullptr, nullptr, nullptr, est},
    {"alpha", halide_argument_kind_input_scalar, 0, ty_f32, nullptr, nullptr, nullptr, &est_alpha, nullptr},
    {"output", halide_argument_kind_output_buffer, 3, ty_f32, nullptr, nullptr, nullptr, nullptr, est},
};
const halide_filter_metadata_t ib_md = {1, 3, ib_args, kTargetString, "iir_blur"};

}  // namespace

extern "C" int hlmi_debug_iir_probe(unsigned long long *out32) {   // 32 counters (probe builds only)
#if HLMI_IIR_PROBE
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_iprobe), sizeof(unsigned long long) * 32) != hipSuccess) return -1;
    unsigned long long zero[32] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_iprobe), zero, sizeof zero) != hipSuccess) return -1;
    return 1;
#else
    (void)out32;
    return 0;
#endif
}

extern "C" int iir_blur(halide_buffer_t *input, float alpha, halide_buffer_t *output) {
    void *uc = nullptr;
    BufArg args[2] = {{"input", input, T_F32, 3, false}, {"output", output, T_F32, 3, true}};
    int r = check_not_null(uc, args, 2);
    if (r) return r;
    if ((r = check_type_and_dims(uc, args, 2))) return r;
    auto real = [](halide_buffer_t *b) { return !(b->host == nullptr && b->device == 0); };
    if (any_bounds_query(args, 2)) {
        // a real input pins the box; otherwise the output's shape is the request (real, or shaped by the caller: RunGen passes
        // every buffer without a host), otherwise a shaped input, otherwise the generator's estimates
        halide_buffer_t *k = real(input) ? input : buffer_known(output) ? output : input;
        const bool any = real(input) || buffer_known(output) || buffer_has_shape(input);
        int z[3] = {0, 0, 0}, e[3] = {any ? k->dim[0].extent : 1536, any ? k->dim[1].extent : 2560, any ? k->dim[2].extent : 3};
        answer_query(input, z, e);
        answer_query(output, z, e);
        return 0;
    }
    if ((r = check_shape(uc, args[0])) || (r = check_shape(uc, args[1]))) return r;
    const int W = input->dim[0].extent, H = input->dim[1].extent, C = input->dim[2].extent;
    char what[48];
    for (int d = 0; d < 3; d++) {   // every scan starts at coordinate 0 and spans the input; output = the same box
        snprintf(what, sizeof what, "input.min.%d", d);
        if ((r = check_equal(uc, what, input->dim[d].min, "0", 0))) return r;
        snprintf(what, sizeof what, "output.min.%d", d);
        if ((r = check_equal(uc, what, output->dim[d].min, "0", 0))) return r;
        snprintf(what, sizeof what, "output.extent.%d", d);
        if ((r = check_equal(uc, what, output->dim[d].extent, "input.extent", input->dim[d].extent))) return r;
    }
    DeviceCtx ctx;
    if ((r = acquire_device(uc, &ctx))) return r;
    if ((r = input_to_device(uc, ctx, args[0]))) return r;
    if ((r = output_on_device(uc, ctx, args[1]))) return r;
    if (W > 0 && H > 0 && C > 0) {
        const size_t plane = ((size_t)W * H + 63) & ~(size_t)63;
        void *ws = nullptr;
        if ((r = get_workspace(uc, ctx, 2 * (size_t)C * plane * sizeof(float), &ws))) return r;
        float *scratch = (float *)ws, *t1 = scratch + (size_t)C * plane;
        hipStream_t st = ctx.stream;
        constexpr size_t lds = sizeof(float) * SLOTS * TSZ;
        HLMI_HIP(uc, hipFuncSetAttribute(reinterpret_cast<const void *>(iir_cols_T), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        timing_note_bytes(16.0 * W * H * C);
        // columns of the input -> t1 = transpose [C][W rows of H]
        HLMI_LAUNCH(uc, "iir_cols_T:1", st, iir_cols_T, dim3((W + 63) / 64, C), dim3(64 * (1 + NHELP)), lds, dev_ptr<float>(input),
                    (long)input->dim[1].stride, (long)input->dim[2].stride, W, H, alpha, scratch, t1, (long)H, (long)W * H);
        timing_note_bytes(16.0 * W * H * C);
        // columns of t1 (= rows of the input) -> output [C][H rows of W]
        HLMI_LAUNCH(uc, "iir_cols_T:2", st, iir_cols_T, dim3((H + 63) / 64, C), dim3(64 * (1 + NHELP)), lds, t1, (long)H, (long)W * H, H, W, alpha,
                    scratch, dev_ptr<float>(output), (long)output->dim[1].stride, (long)output->dim[2].stride);
    }
    mark_output_written(output);
    return 0;
}

extern "C" int iir_blur_argv(void **a) { return iir_blur((halide_buffer_t *)a[0], *(float *)a[1], (halide_buffer_t *)a[2]); }
extern "C" const halide_filter_metadata_t *iir_blur_metadata(void) { return &ib_md; }
extern "C" int iir_blur_auto_schedule(halide_buffer_t *input, float alpha, halide_buffer_t *output) { return iir_blur(input, alpha, output); }
