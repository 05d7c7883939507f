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
//   tile p = 64 rows x 64 columns, three LDS slots in rotation:  slot (p+1)%3 is being filled with tile p+1 (its rows were
//   requested three iterations earlier and waited for in registers), slot p%3 is scanned in place, slot (p-1)%3 is written out
//   forward   b = (1-a) b + a in(x, y), top to bottom; tiles go to a scratch plane as 64-float rows
//   backward  b = (1-a) b + a scratch(x, y), bottom to top; tiles are written out TRANSPOSED (the generator transposes
//             after each column blur, :31): lane = y, 64 consecutive floats per store
// Launched twice (columns of the input, then columns of the transposed intermediate = rows of the input).
#include "hlmi_internal.h"

#include <type_traits>

using namespace hlmi;

namespace {

constexpr int TR = 64, TP = 65;          // tile rows, LDS pitch (odd: the transposed reads are conflict-free)
constexpr int NHELP = 8, HR = TR / NHELP; // helper waves and the tile rows each of them moves
constexpr int NSET = 4;                   // tiles a helper has requested and not yet put into LDS (+1 being filled)

// src: [C][Hd][Wd] (row stride s_sy, plane stride s_sc); scratch: same shape, dense; dst: [C][Wd][Hd] (d_sy, d_sc)
__global__ __launch_bounds__(64 * (1 + NHELP)) void iir_cols_T(const float *__restrict__ src, long s_sy, long s_sc, int Wd, int Hd,
                                                               float alpha, float *__restrict__ scratch, float *__restrict__ dst,
                                                               long d_sy, long d_sc) {
    __shared__ float tiles[3][TR * TP];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int x0 = blockIdx.x * 64, c = blockIdx.y;
    const int x = min(x0 + lane, Wd - 1);                    // lanes past the edge shadow the last column (never stored)
    const float c1 = 1.0f - alpha;
    const float *s = src + (long)c * s_sc + x;
    float *t = scratch + ((long)c * Hd) * Wd + x;
    float *dcol = dst + (long)c * d_sc;
    const int NT = (Hd + TR - 1) / TR;
    const int h0 = (wave - 1) * HR;                          // helper: first tile row it moves
    float b = 0.0f;                                          // scanner: the recurrence state
    float set[NSET][HR];                                     // helper: rows of NSET tiles in flight

    // one pass over the column: tile p (processing order) holds steps 0..n-1; forward: step st = row 64p + st of `s`,
    // backward: row Hd-1 - 64p - st of the scratch plane.  The scanner and the helpers run DIFFERENT loops that meet at the
    // same barriers, and the helpers' steady-state iteration is straight-line code (clamped indices instead of guards):
    // any branch around the loads makes the compiler drain all outstanding loads at the join, which is the latency
    // this structure exists to hide.
    auto pass = [&](auto back_tag) {
        constexpr bool BACK = decltype(back_tag)::value;
        auto rows_of = [&](int p) { return min(TR, Hd - p * TR); };
        auto row_at = [&](int p, int st) { return BACK ? Hd - 1 - p * TR - st : p * TR + st; };
        if (wave == 0) {
            // ---- the scanner: b = c1 * b + (alpha * in), in place in the tile's LDS slot
            __syncthreads();
            for (int p = 0; p < NT; p++) {
                float *tl = tiles[p % 3] + lane;
                const int n = rows_of(p);
                if (n == TR && p != 0) {                     // a full tile: all 64 reads in flight before the chain starts
                    float u[TR];
#pragma unroll
                    for (int k = 0; k < TR; k++) u[k] = tl[k * TP];
#pragma unroll
                    for (int k = 0; k < TR; k++) {
                        b = c1 * b + u[k];
                        tl[k * TP] = b;
                    }
                } else {
                    int st = 0;
                    if (p == 0) {                            // the first row of a pass is taken as it is (:21, :26-28)
                        b = tl[0];
                        st = 1;
                    }
                    for (; st < n; st++) {
                        b = c1 * b + tl[st * TP];
                        tl[st * TP] = b;
                    }
                }
                __syncthreads();
            }
            __syncthreads();
            return;
        }
        // ---- the helpers
        auto issue = [&](int p, float (&v)[HR]) {            // p past the end re-reads the last tile, rows past a short tile its last row
            const int pc = min(p, NT - 1), last = rows_of(pc) - 1;
#pragma unroll
            for (int i = 0; i < HR; i++) {
                const int r = row_at(pc, min(h0 + i, last));
                v[i] = BACK ? t[(long)r * Wd] : s[(long)r * s_sy];
            }
        };
        // the helpers also do the recurrence's independent product: LDS holds alpha * in (the pass's very first row stays
        // raw).  Filling a slot for a tile past the end is harmless: that slot's tile has been written out already.
        auto fill = [&](int p, const float (&v)[HR]) {
            float *tl = tiles[p % 3];
#pragma unroll
            for (int i = 0; i < HR; i++) tl[(h0 + i) * TP + lane] = (p == 0 && h0 + i == 0) ? v[i] : alpha * v[i];
        };
        auto drain_full = [&](int p) {                       // tile p (64 rows) leaves LDS
            const float *tl = tiles[p % 3];
#pragma unroll
            for (int i = 0; i < HR; i++) {
                if (!BACK) {
                    t[(long)row_at(p, h0 + i) * Wd] = tl[(h0 + i) * TP + lane];
                } else {
                    // transposed: dst[c][x0 + j][y], y = the row of step `lane` (64 consecutive floats, descending with the
                    // lane); columns past the edge repeat the last one (same value to the same address)
                    const int j = min(h0 + i, Wd - 1 - x0);
                    dcol[(long)(x0 + j) * d_sy + row_at(p, lane)] = tl[lane * TP + j];
                }
            }
        };
        auto drain_last = [&](int p) {                       // the (possibly short) last tile
            const float *tl = tiles[p % 3];
            const int n = rows_of(p);
#pragma unroll
            for (int i = 0; i < HR; i++) {
                if (!BACK) {
                    if (h0 + i < n) t[(long)row_at(p, h0 + i) * Wd] = tl[(h0 + i) * TP + lane];
                } else {
                    const int j = min(h0 + i, Wd - 1 - x0);
                    if (lane < n) dcol[(long)(x0 + j) * d_sy + row_at(p, lane)] = tl[lane * TP + j];
                }
            }
        };
        auto iter = [&](int p, auto ph_tag) {                // steady state, 1 <= p: no branches
            constexpr int PH = decltype(ph_tag)::value;      // p % NSET: the register set indices must be compile-time
            fill(p + 1, set[(PH + 1) % NSET]);
            issue(p + NSET, set[PH]);
            drain_full(p - 1);
            __syncthreads();
        };
        static_assert(NSET == 4, "the unrolled tile loop below names the phases");
        issue(0, set[0]);
        issue(1, set[1]);
        issue(2, set[2]);
        issue(3, set[3]);
        fill(0, set[0]);
        __syncthreads();
        fill(1, set[1]);                                     // iteration 0: nothing to write out yet
        issue(4, set[0]);
        __syncthreads();
        int p = 1;
        for (; p + 3 < NT; p += 4) {
            iter(p, std::integral_constant<int, 1>{});
            iter(p + 1, std::integral_constant<int, 2>{});
            iter(p + 2, std::integral_constant<int, 3>{});
            iter(p + 3, std::integral_constant<int, 0>{});
        }
        if (p < NT) iter(p++, std::integral_constant<int, 1>{});
        if (p < NT) iter(p++, std::integral_constant<int, 2>{});
        if (p < NT) iter(p++, std::integral_constant<int, 3>{});
        drain_last(NT - 1);
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

extern "C" int iir_blur(halide_buffer_t *input, float alpha, halide_buffer_t *output) {
    void *uc = nullptr;
    BufArg args[2] = {{"input", input, T_F32, 3, false}, {"output", output, T_F32, 3, true}};
    int r = check_not_null(uc, args, 2);
    if (r) return r;
    if ((r = check_type_and_dims(uc, args, 2))) return r;
    auto real = [](halide_buffer_t *b) { return !(b->host == nullptr && b->device == 0); };
    if (any_bounds_query(args, 2)) {
        halide_buffer_t *k = real(input) ? input : output;
        const bool any = real(input) || real(output);
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
        timing_note_bytes(16.0 * W * H * C);
        // columns of the input -> t1 = transpose [C][W rows of H]
        HLMI_LAUNCH(uc, "iir_cols_T:1", st, iir_cols_T, dim3((W + 63) / 64, C), dim3(64 * (1 + NHELP)), 0, dev_ptr<float>(input),
                    (long)input->dim[1].stride, (long)input->dim[2].stride, W, H, alpha, scratch, t1, (long)H, (long)W * H);
        timing_note_bytes(16.0 * W * H * C);
        // columns of t1 (= rows of the input) -> output [C][H rows of W]
        HLMI_LAUNCH(uc, "iir_cols_T:2", st, iir_cols_T, dim3((H + 63) / 64, C), dim3(64 * (1 + NHELP)), 0, t1, (long)H, (long)W * H, H, W, alpha,
                    scratch, dev_ptr<float>(output), (long)output->dim[1].stride, (long)output->dim[2].stride);
    }
    mark_output_written(output);
    return 0;
}

extern "C" int iir_blur_argv(void **a) { return iir_blur((halide_buffer_t *)a[0], *(float *)a[1], (halide_buffer_t *)a[2]); }
extern "C" const halide_filter_metadata_t *iir_blur_metadata(void) { return &ib_md; }
extern "C" int iir_blur_auto_schedule(halide_buffer_t *input, float alpha, halide_buffer_t *output) { return iir_blur(input, alpha, output); }
