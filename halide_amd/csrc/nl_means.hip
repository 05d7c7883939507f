// nl_means.hip — gfx950 implementation of the reference's non-local-means AOT pipeline.
//
// Algorithm: /root/reference/apps/nl_means/nl_means_generator.cpp:24-63; boundary: `int nl_means(
// halide_buffer_t *input, int32_t patch_size, int32_t search_area, float sigma, halide_buffer_t *non_local_means)`
// (:9-14, :162).  This path is fp32-VALU/LDS bound (~2.4 kFLOP and 24 B per pixel), not HBM bound.
//
// Float sums are order-sensitive and the oracle fixes the order (d over channels, blur_d_y over y, blur_d
// over x, 49-term weighted sum over the search window, x fastest) — so no running-sum box filters: every sum
// is re-added in order from shared, deterministic partial results.
//
// Fast path (patch_size == search_area == 7, the reference's benchmark setting, :79-81): one workgroup owns a
// 58 x 64 output tile, stages the (58+12) x (64+12) x 3 clamped input window in LDS once, then for each of the
// 49 offsets (dy outer, dx inner):
//   phase 1  thread <-> (column of the 64-wide blur_d_y tile, 8-row segment): walks down the column with the
//            last 7 values of d in registers, writes blur_d_y to LDS
//   phase 2  thread <-> (row, 8-px segment): walks along the row with the last 7 blur_d_y values in registers,
//            w = fast_exp(blur_d * inv_sigma_sq), accumulates 4 sums per pixel in registers
// 512 threads per workgroup (the 81 KB window allows one workgroup per CU: 2 waves per SIMD).
// Odd LDS pitches (71, 65) keep the row-per-lane accesses of phase 2 conflict-free.
// Any other (patch_size, search_area): a straightforward one-thread-per-pixel kernel from global memory.
#include "hlmi_device_math.h"
#include "hlmi_internal.h"

#include <stdlib.h>

using namespace hlmi;

namespace {

constexpr int P = 7, SA = 7, HALF = 3;
constexpr int TW = 58;                           // output tile width; the height TH is a template parameter (64 or 80)
constexpr int IW = TW + 4 * HALF, IWP = 71;      // input window (halo 3 patch + 3 search), padded pitch
// Threads per workgroup NT = 16 TH: a thread owns 4 rows of a blur_d_y column in phase 1 (ROWS1) and 4 pixels of a row in
// phase 2 (SEG).  TH = 64 (1024 threads, 81 KB: one workgroup per CU) leaves a third of the chip idle at 1920 x 1080 —
// 578 tiles on 256 CUs run as three rounds of ~95 us (scripts/nlm_scale.py) — TH = 16 (256 threads, 28 KB, five
// workgroups per CU that interleave their phases) has 2312 tiles that back-fill the CUs as they finish.

struct NGeom {
    int ix0, ix1, iy0, iy1, ic0, ic1;  // clamp box of the input (absolute)
    int ox0, oy0, ow, oh;              // output region (absolute origin, extents)
    float inv;                         // -1 / (sigma*sigma*patch*patch)
};

// The clamped input window (repeat_edge on x, y and c, generator :27) of a tile: WH rows x WW columns (64 < WW <= 128, LDS
// pitch IWP) x 3 channels from (wx0, wy0).  A wave takes every NW-th row: a lane loads its column and lanes below WW - 64 the
// row's tail, and a wave requests half of its rows before it writes the first to LDS — two memory round trips per tile
// instead of one per pair of elements (the flat-index loop this replaces waited for each pair of loads).
template<int WH, int WW, int NW>
__device__ __forceinline__ void stage_window(float *sin, const float *__restrict__ in, long in_sy, long in_sc, const NGeom &g, int wx0,
                                             int wy0, int lane, int wave) {
    constexpr int ROWS_ALL = 3 * WH, PER = (ROWS_ALL + NW - 1) / NW, BATCH = (PER + 1) / 2;
    const int xa = dev::clampi(wx0 + lane, g.ix0, g.ix1) - g.ix0;
    const int xb = dev::clampi(wx0 + 64 + lane, g.ix0, g.ix1) - g.ix0;
    const bool tail = lane < WW - 64;
#pragma unroll 1
    for (int b0 = 0; b0 < PER; b0 += BATCH) {
        float va[BATCH], vb[BATCH];
#pragma unroll
        for (int b = 0; b < BATCH; b++) {
            const int rr = wave + NW * (b0 + b);
            va[b] = vb[b] = 0.0f;
            if (b0 + b < PER && rr < ROWS_ALL) {
                const int c = rr / WH, r = rr - c * WH;
                const int y = dev::clampi(wy0 + r, g.iy0, g.iy1) - g.iy0, cc = dev::clampi(c, g.ic0, g.ic1) - g.ic0;
                const float *row = in + (long)y * in_sy + (long)cc * in_sc;
                va[b] = row[xa];
                if (tail) vb[b] = row[xb];
            }
        }
#pragma unroll
        for (int b = 0; b < BATCH; b++) {
            const int rr = wave + NW * (b0 + b);
            if (b0 + b < PER && rr < ROWS_ALL) {
                sin[rr * IWP + lane] = va[b];
                if (tail) sin[rr * IWP + 64 + lane] = vb[b];
            }
        }
    }
}

// ---- nlm_7x7w: the same computation with NO barrier inside the offset loop.  nlm_7x7 hands blur_d_y from its phase-1 threads
// (a column, four rows) to its phase-2 threads (a row, four pixels) through LDS, one workgroup barrier per offset: all waves
// of a workgroup read at the same time, compute at the same time and wait for each other 49 times (a third of all
// wave-cycles, SQ_WAIT_INST_ANY).  Here a WAVE owns 64 columns x 4 rows of the tile for BOTH stages — lane = column:
//   * stage 1 as before: d for the lane's column over the wave's 4 + 6 rows, ordered 7-term column sums -> blur_d_y of 4 rows;
//   * stage 2 in the same lane: blur_d(x) = sum over blur_d_y(x - 3 .. x + 3) comes from the six neighbouring LANES by DPP wave
//     shifts (three to the right, three to the left, added in the oracle's order: leftmost first), so lane L finishes output
//     column L - 3 (lanes 3 .. 60: the tile's 58 columns; the shifted-in zeros only reach lanes whose results are not stored);
//   * the pixel values in(x + dx, y + dy) the weighted sum needs ARE the shifted operand of stage 1 (rows 3 .. 6 of the
//     lane's ten): no second read.
// Per offset and lane: 30 LDS reads (the shifted operand), 6 DPP moves per output row, no LDS writes, no barrier; waves never
// wait for each other after the window has been staged.
// Round 3 — the kernel is VALU-issue bound (every instruction left is arithmetic of the algorithm: per pixel and offset 24.5
// adds, 18.7 multiplies, 6.3 subtractions, 4 conversions at eight rows per wave; SQ_ACTIVE_INST_VALU ~ all cycles), so what
// helps is fewer VALU instructions: XL takes the six DPP moves per value off the VALU (0.198 -> 0.177 ms), eight rows per wave
// instead of four recompute fewer rows of d (14 / 8 instead of 10 / 4 per output row: -> 0.169 ms).  What did NOT help:
// packed f32 arithmetic on pairs of offsets (v_pk_add_f32 / v_pk_mul_f32: 30 % fewer instructions, bit-exact, 0.212 ms — the
// packed f32 operations do not issue at the scalar rate on this part), sixteen rows per wave (0.37 ms: registers).
__device__ __forceinline__ float nl_lane_prev(float v) {  // value held by lane-1 (0 for lane 0): DPP wave_shr:1
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float nl_lane_next(float v) {  // value held by lane+1 (0 for lane 63): DPP wave_shl:1
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}
// XL: the sum across columns reads the six neighbours' blur_d_y from a wave-private LDS row (the lanes of a wave run in lock
// step and a wave's LDS operations complete in order) instead of six chained DPP moves per value: the VALU is the unit this
// kernel saturates (79 instructions per pixel and offset, every issue cycle taken: profiles/r03_nlm_pmc.txt), the LDS pipe is not.
constexpr int XSP = 64 + 2 * HALF;   // a wave's row: three columns of padding either side (their results are never used)
template<int TH, int ROWS = 4, bool XL = false>
__global__ __launch_bounds__(64 * TH / ROWS) void nlm_7x7w(const float *__restrict__ in, long in_sy, long in_sc, NGeom g,
                                                    float *__restrict__ out, long out_sy, long out_sc) {
    constexpr int NT = 64 * TH / ROWS, IH = TH + 4 * HALF;
    extern __shared__ float lds[];
    float *sin = lds;                       // [3][IH][IWP]
    const int tid = threadIdx.x;
    const int tx0 = g.ox0 + blockIdx.x * TW, ty0 = g.oy0 + blockIdx.y * TH;  // absolute coords of the tile
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    stage_window<IH, IW, NT / 64>(sin, in, in_sy, in_sc, g, tx0 - 2 * HALF, ty0 - 2 * HALF, lane, wave);
    __syncthreads();
    const int col = lane + HALF;             // window column of abs x = tx0 - 3 + lane
    const int row0 = ROWS * wave + HALF;     // window row of abs y = ty0 + ROWS * wave - 3
    float *xs = lds + 3 * IH * IWP + wave * ROWS * XSP + HALF + lane;   // XL: this lane's column of the wave's rows
    float u[ROWS + 6][3];
#pragma unroll
    for (int i = 0; i < ROWS + 6; i++) {
#pragma unroll
        for (int c = 0; c < 3; c++) u[i][c] = sin[(c * IH + row0 + i) * IWP + col];
    }
    float acc[ROWS][4];
#pragma unroll
    for (int o = 0; o < ROWS; o++) acc[o][0] = acc[o][1] = acc[o][2] = acc[o][3] = 0.0f;
#pragma unroll 1
    for (int dy = -HALF; dy <= HALF; dy++) {
        const float *shifted = sin + (row0 + dy) * IWP + col;
#pragma unroll
        for (int dxi = 0; dxi < SA; dxi++) {
            const int dx = dxi - HALF;
            float sh[ROWS + 6][3];
#pragma unroll
            for (int i = 0; i < ROWS + 6; i++) {
#pragma unroll
                for (int c = 0; c < 3; c++) sh[i][c] = shifted[(c * IH + i) * IWP + dx];
            }
            __builtin_amdgcn_sched_barrier(0);
            float d[ROWS + 6];
#pragma unroll
            for (int i = 0; i < ROWS + 6; i++) {
                float dd;   // the oracle's sums start from 0; 0 + x is x for every x but -0, and a square is never -0
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    const float t = u[i][c] - sh[i][c];
                    dd = c == 0 ? t * t : dev::mad(t, t, dd);   // 0 + t t == t t in either form
                }
                d[i] = dd;
            }
            float bdy[ROWS];
#pragma unroll
            for (int o = 0; o < ROWS; o++) {
                float sum = d[o];
#pragma unroll
                for (int q = 1; q < 7; q++) sum = sum + d[o + q];
                bdy[o] = sum;
            }
            if (XL) {
#pragma unroll
                for (int o = 0; o < ROWS; o++) xs[o * XSP] = bdy[o];
                // what the neighbours stored above is what the loads below return; the fences keep the compiler from moving
                // one across the other
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
            // blur_d for output column lane - 3: blur_d_y of lanes lane - 3 .. lane + 3, leftmost first
#pragma unroll
            for (int o = 0; o < ROWS; o++) {
                float r1, r2, r3, l1, l2, l3;
                if (XL) {
                    const float *n = xs + o * XSP;
                    r3 = n[-3], r2 = n[-2], r1 = n[-1], l1 = n[1], l2 = n[2], l3 = n[3];
                } else {
                    r1 = nl_lane_prev(bdy[o]), r2 = nl_lane_prev(r1), r3 = nl_lane_prev(r2);
                    l1 = nl_lane_next(bdy[o]), l2 = nl_lane_next(l1), l3 = nl_lane_next(l2);
                }
                const float sum = (((((r3 + r2) + r1) + bdy[o]) + l1) + l2) + l3;
                const float w = dev::fast_exp(sum * g.inv);
#pragma unroll
                for (int c = 0; c < 3; c++) acc[o][c] = dev::mad(w, sh[o + HALF][c], acc[o][c]);
                acc[o][3] = acc[o][3] + w * 1.0f;
            }
            if (XL) {   // the next offset's stores stay behind these loads
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
    // ---- normalise + store: lane L holds output column L - 3
    const int X = tx0 + lane - HALF - g.ox0;
    if (lane >= HALF && lane < HALF + TW && X < g.ow) {
#pragma unroll
        for (int o = 0; o < ROWS; o++) {
            const int Y = ty0 + ROWS * wave + o - g.oy0;
            if (Y < g.oh) {
#pragma unroll
                for (int c = 0; c < 3; c++) out[(long)Y * out_sy + X + (long)c * out_sc] = dev::clampf(acc[o][c] / acc[o][3], 0.0f, 1.0f);
            }
        }
    }
}

// Measured and not kept (round 5, profiles/r05_nlm_shared_d.txt): `nlm_7x7s`, this kernel with every row of d computed ONCE per
// workgroup and offset — a wave owns the eight rows whose shifted operand it needs anyway plus at most two of the tile's six
// border rows (10 rows of d per wave instead of 14) and takes the three rows above and below from its neighbours through a
// double-buffered LDS exchange, one workgroup barrier per offset.  Bit-exact; 0.215 ms per call against 0.154-0.165 for this kernel
// on the same box, SQ_INSTS_VALU 99.36 M for both: the compiler only holds the exchange version in registers with the seven
// column offsets NOT unrolled (unrolled: 560 scratch accesses), and the loop and address arithmetic that costs eats the saved
// d rows, while the 49 barriers put the waves of a workgroup back in lock step (SQ_BUSY_CYCLES 10.5 M -> 15.7 M).

// generic (patch, search): one thread per pixel, everything from (L2-resident) global memory, same sum orders
__global__ __launch_bounds__(256) void nlm_generic(const float *__restrict__ in, long in_sy, long in_sc, NGeom g, int patch,
                                                  int search, float *__restrict__ out, long out_sy, long out_sc) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= g.ow) return;
    const int X = g.ox0 + x, Y = g.oy0 + y;
    const int p0 = -(patch / 2), s0 = -(search / 2);
    auto IN = [&](int ax, int ay, int c) -> float {
        int xx = dev::clampi(ax, g.ix0, g.ix1) - g.ix0, yy = dev::clampi(ay, g.iy0, g.iy1) - g.iy0;
        int cc = dev::clampi(c, g.ic0, g.ic1) - g.ic0;
        return in[(long)yy * in_sy + xx + (long)cc * in_sc];
    };
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int sy = s0; sy < s0 + search; sy++) {
        for (int sx = s0; sx < s0 + search; sx++) {
            float bd = 0.0f;
            for (int px = p0; px < p0 + patch; px++) {
                float bdy = 0.0f;
                for (int py = p0; py < p0 + patch; py++) {
                    float d = 0.0f;
                    for (int c = 0; c < 3; c++) {
                        float t = IN(X + px, Y + py, c) - IN(X + px + sx, Y + py + sy, c);
                        d = dev::mad(t, t, d);
                    }
                    bdy = bdy + d;
                }
                bd = bd + bdy;
            }
            float w = dev::fast_exp(bd * g.inv);
            for (int c = 0; c < 3; c++) acc[c] = dev::mad(w, IN(X + sx, Y + sy, c), acc[c]);
            acc[3] = acc[3] + w * 1.0f;
        }
    }
    for (int c = 0; c < 3; c++) out[(long)y * out_sy + x + (long)c * out_sc] = dev::clampf(acc[c] / acc[3], 0.0f, 1.0f);
}

const int64_t e0 = 0, ew = 1536, eh = 2560, ec = 3;
const int64_t *const est[6] = {&e0, &ew, &e0, &eh, &e0, &ec};
const halide_scalar_value_t est7 = [] { halide_scalar_value_t v{}; v.u.i32 = 7; return v; }();
const halide_scalar_value_t est_sigma = [] { halide_scalar_value_t v{}; v.u.f32 = 0.12f; return v; }();
const halide_type_t ty_f32 = {(decltype(halide_type_t::code))2, 32, 0};
const halide_type_t ty_i32 = {(decltype(halide_type_t::code))0, 32, 0};
// estimates: generator :76-82
const halide_filter_argument_t nlm_args[5] = {
    {"input", halide_argument_kind_input_buffer, 3, ty_f32, nullptr, nullptr, nullptr, nullptr, est},
    {"patch_size", halide_argument_kind_input_scalar, 0, ty_i32, nullptr, nullptr, nullptr, &est7, nullptr},
    {"search_area", halide_argument_kind_input_scalar, 0, ty_i32, nullptr, nullptr, nullptr, &est7, nullptr},
    {"sigma", halide_argument_kind_input_scalar, 0, ty_f32, nullptr, nullptr, nullptr, &est_sigma, nullptr},
    {"non_local_means", halide_argument_kind_output_buffer, 3, ty_f32, nullptr, nullptr, nullptr, nullptr, est},
};
const halide_filter_metadata_t nlm_md = {1, 5, nlm_args, kTargetString, "nl_means"};

}  // namespace

extern "C" int nl_means(halide_buffer_t *input, int32_t patch_size, int32_t search_area, float sigma,
                        halide_buffer_t *output) {
    void *uc = nullptr;
    BufArg args[2] = {{"input", input, T_F32, 3, false}, {"non_local_means", output, T_F32, 3, true}};
    int r = check_not_null(uc, args, 2);
    if (r) return r;
    if ((r = check_type_and_dims(uc, args, 2))) return r;
    if (any_bounds_query(args, 2)) {
        // all input taps are clamped (:27); propose the output's x/y region and the 3 channels the algorithm names.
        // The output's channel dimension is pinned to [0, 3) (:68).
        int mins[3] = {output->dim[0].min, output->dim[1].min, 0}, ext[3] = {output->dim[0].extent, output->dim[1].extent, 3};
        answer_query(input, mins, ext);
        answer_query(output, mins, ext);
        return 0;
    }
    if ((r = check_shape(uc, args[0])) || (r = check_shape(uc, args[1]))) return r;
    if ((r = check_equal(uc, "non_local_means.min.2", output->dim[2].min, "0", 0))) return r;
    if ((r = check_equal(uc, "non_local_means.extent.2", output->dim[2].extent, "3", 3))) return r;
    if (patch_size < 1 || search_area < 1) {
        return report(uc, halide_error_code_param_too_small, "Parameters patch_size (%d) and search_area (%d) must be >= 1",
                      patch_size, search_area);
    }
    const int ow = output->dim[0].extent, oh = output->dim[1].extent;
    if (ow > 0 && oh > 0 && (input->dim[0].extent < 1 || input->dim[1].extent < 1 || input->dim[2].extent < 1)) {
        return report(uc, halide_error_code_access_out_of_bounds, "Input buffer input is empty but is accessed (clamped)");
    }
    DeviceCtx ctx;
    if ((r = acquire_device(uc, &ctx))) return r;
    if ((r = input_to_device(uc, ctx, args[0]))) return r;
    if ((r = output_on_device(uc, ctx, args[1]))) return r;
    if (ow == 0 || oh == 0) {
        mark_output_written(output);
        return 0;
    }
    NGeom g;
    g.ix0 = input->dim[0].min, g.ix1 = g.ix0 + input->dim[0].extent - 1;
    g.iy0 = input->dim[1].min, g.iy1 = g.iy0 + input->dim[1].extent - 1;
    g.ic0 = input->dim[2].min, g.ic1 = g.ic0 + input->dim[2].extent - 1;
    g.ox0 = output->dim[0].min, g.oy0 = output->dim[1].min, g.ow = ow, g.oh = oh;
    g.inv = -1.0f / (((sigma * sigma) * (float)patch_size) * (float)patch_size);
    const float *din = dev_ptr<float>(input);
    float *dout = dev_ptr<float>(output);
    const long in_sy = input->dim[1].stride, in_sc = input->dim[2].stride;
    const long out_sy = output->dim[1].stride, out_sc = output->dim[2].stride;
    if (patch_size == P && search_area == SA) {
        // nlm_7x7w<32, 8, true>: 58 x 32 output tiles, a wave owns 64 columns x 8 rows for both stages, the row sum through a
        // wave-private LDS row.  (Round 2's kernel with a barrier per offset, four rows per wave, the DPP row sum and the other tile
        // heights were A/B switches of rounds 2-3 — HLMI_NLM_LDS / ROWS4 / XLDS / TH — and went with their measurements,
        // profiles/r03_nlm_pmc.txt.)
        const size_t sh_w = sizeof(float) * ((size_t)3 * (32 + 4 * HALF) * IWP + (size_t)32 * XSP);
        dim3 grid((ow + TW - 1) / TW, (oh + 31) / 32);
        HLMI_LAUNCH(uc, "nlm_7x7", ctx.stream, (nlm_7x7w<32, 8, true>), grid, dim3(256), sh_w, din, in_sy, in_sc, g, dout, out_sy, out_sc);
    } else {
        HLMI_LAUNCH(uc, "nlm_generic", ctx.stream, nlm_generic, dim3((ow + 255) / 256, oh), dim3(256), 0, din, in_sy, in_sc, g,
                    patch_size, search_area, dout, out_sy, out_sc);
    }
    mark_output_written(output);
    return 0;
}

extern "C" int nl_means_argv(void **a) {
    return nl_means((halide_buffer_t *)a[0], *(int32_t *)a[1], *(int32_t *)a[2], *(float *)a[3], (halide_buffer_t *)a[4]);
}
extern "C" const halide_filter_metadata_t *nl_means_metadata(void) { return &nlm_md; }
extern "C" int nl_means_auto_schedule(halide_buffer_t *input, int32_t patch_size, int32_t search_area, float sigma,
                                      halide_buffer_t *output) {
    return nl_means(input, patch_size, search_area, sigma, output);
}
