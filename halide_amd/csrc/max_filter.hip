// max_filter.hip — gfx950 implementation of the reference's max_filter AOT pipeline (an adjacent app with
// the same boundary).  Algorithm: /root/reference/apps/max_filter/max_filter_generator.cpp:14-53 (radius = 26, :10);
// boundary: `int max_filter(halide_buffer_t *input, halide_buffer_t *output)`, f32 [W,H,3] planar in and out (:11-12).
//
// Closed form of the reference's log-slice construction (oracle/max_filter_oracle.c evaluates it literally; the parity
// tests are the proof): with h(dx) = clamp(filter_height(dx), 0, 27) and the input edge-clamped in x and y,
//     output(x, y, c) = max { input(x + dx, y + dy, c) : |dx| <= 26, |dy| <= h(dx) }.
// vert(., y, ., t) is the max over rows [y - t, y + t] — the two 2^s-tall samples overlap or abut for s =
// floor(log2(2t + 1)), and the rows the vert_log update never touches (below -26, above height - 1) hold the clamped edge
// row, which the other sample covers.  That argument needs the input's rows to start at 0 (the update's row range is
// absolute, :28); other mins are refused.  It also fails for output rows ABOVE the image (-26 <= y <= -12): there the second
// sample can start below row -26, where vert_log was never updated and holds row 0 alone, so the reference's result misses
// rows the footprint has.  Output rows y < 0 therefore go through max_filter_literal, which evaluates vert as written.
// Only comparisons: the result is exact whatever the evaluation order.
//
// The footprint is transposed here — rows of a 55-row window, each a horizontal max of half-width w(|dy|) = max{dx :
// h(dx) >= |dy|} — so that lanes read consecutive LDS words: a workgroup owns a 64 x 64 output tile, walks the 118 input
// rows it depends on in chunks of 8, builds the horizontal doubling slices R_s[i] = max of 2^s consecutive pixels in LDS
// (5 steps; since round 5 in registers, see the kernel), and every thread folds max(R_s[x - w], R_s[x + w + 1 - 2^s]) into its 8
// outputs.  110 sample reads per output (about 70 LDS instructions after the outputs of a lane share them);
// HBM: 4 B/px/channel read (+ the tile halo from L2) and 4 written.
#include "hlmi_internal.h"

#include <math.h>

using namespace hlmi;

namespace {

constexpr int RAD = 26, VR = RAD + 1;          // horizontal radius; largest vertical half-height (filter_height(0) = 27)
constexpr int TW = 64, TY = 64, NT = 256;      // output tile, threads (four waves)
constexpr int CH = 8;                          // input rows per chunk
constexpr int ORW = TY / (NT / 64);            // output rows per wave (16: a lane's outputs share more of their sample reads than with 8)
constexpr int SW = TW + 2 * RAD;               // 116 staged pixels per row
constexpr int SP = 128;                        // LDS row pitch
static_assert(SW <= SP, "a staged row fits its pitch");
constexpr int NSLOT = 4;                       // LDS slices: R_2, R_3, R_4, R_5 (R_s = max of 2^s consecutive pixels; R_0 and R_1 stay in registers)
constexpr int NCHUNK = (TY + 2 * VR + CH - 1) / CH;
constexpr int NFOLD = (ORW - 1 + 2 * VR) / CH + 1;  // chunks that hold rows of one wave's 16 + 54 row window

// filter_height(dx) clamped to [0, 27] (generator :47-53)
constexpr int mf_height(int dx) {
    int n = 0;
    for (int dy = 0; dy <= RAD; dy++) n += ((float)(dx * dx + dy * dy) < (RAD + 0.25f) * (RAD + 0.25f)) ? 1 : 0;
    return n < VR ? n : VR;
}
// half-width of the footprint's row |dy|: the largest dx whose column reaches it (the heights fall with |dx|)
constexpr int mf_halfwidth(int ady) {
    int w = 0;
    for (int dx = 0; dx <= RAD; dx++) {
        if (mf_height(dx) >= ady) w = dx;
    }
    return w;
}
constexpr int mf_log2(int v) {   // floor(log2(v)), v >= 1: slice_for_radius(t) = mf_log2(2t + 1) (:37)
    int s = 0;
    while ((2 << s) <= v) s++;
    return s;
}
static_assert(mf_halfwidth(0) == RAD && mf_halfwidth(VR) == 3 && mf_log2(2 * 3 + 1) == 2 && mf_log2(2 * RAD + 1) == 5,
              "every row of the footprint is two samples of a slice R_2 .. R_5");
// LDS word offsets (relative to row start + own column) of the two samples covering [x - w, x + w]
constexpr int mf_slot(int ady) { return mf_log2(2 * mf_halfwidth(ady) + 1) - 2; }   // slice R_s lives in slot s - 2
constexpr int mf_offA(int ady) { return mf_slot(ady) * CH * SP + RAD - mf_halfwidth(ady); }
constexpr int mf_offB(int ady) { return mf_slot(ady) * CH * SP + RAD + mf_halfwidth(ady) + 1 - (1 << (mf_slot(ady) + 2)); }

struct MFGeom {
    int ix0, iy0, W, H;
    int ox0, oy0, ow, oh;
    long in_sy, in_sc, out_sy, out_sc;
    int t_of[RAD + 1], s_of[RAD + 1]; // per |dx|: clamp(filter_height, 0, 27) and its slice (max_filter_literal)
    int lit_rows;                     // output rows y < 0, handled by max_filter_literal
};

// vert_log(x, row, c, s) as the reference leaves it (:26-30): rows -26 .. H-1 were updated and hold the max of 2^s rows
// starting at `row` (edge-clamped); every other row still holds the clamped input row.
__device__ __forceinline__ float mf_vert_log(const float *col, long sy, int H, int row, int s) {
    const int n = (row >= -RAD && row <= H - 1) ? (1 << s) : 1;
    float m = -INFINITY;
    for (int i = 0; i < n; i++) m = fmaxf(m, col[(long)min(max(row + i, 0), H - 1) * sy]);
    return m;
}

// Output rows above the image (y < 0; a few rows of a region nobody normally asks for): the definition as written, one
// thread per output pixel.
__global__ __launch_bounds__(256) void max_filter_literal(const float *__restrict__ in, float *__restrict__ out, MFGeom g) {
    const int x = blockIdx.x * 256 + threadIdx.x, yr = blockIdx.y;
    if (x >= g.ow) return;
    const int X = g.ox0 + x, Y = g.oy0 + yr;
    const float *inc = in + (long)blockIdx.z * g.in_sc;
    float m = -INFINITY;
    for (int dx = -RAD; dx <= RAD; dx++) {
        const int t = g.t_of[dx < 0 ? -dx : dx], s = g.s_of[dx < 0 ? -dx : dx];
        const float *col = inc + (min(max(X + dx, g.ix0), g.ix0 + g.W - 1) - g.ix0);
        m = fmaxf(m, fmaxf(mf_vert_log(col, g.in_sy, g.H, Y - t, s), mf_vert_log(col, g.in_sy, g.H, Y + t + 1 - (1 << s), s)));
    }
    out[(long)blockIdx.z * g.out_sc + (long)yr * g.out_sy + x] = m;
}

// acc = max(acc, a, b) as one instruction; the compiler's fmaxf would first canonicalise both loaded values
__device__ __forceinline__ void mf_max3(float &acc, float a, float b) { asm("v_max3_f32 %0, %0, %1, %2" : "+v"(acc) : "v"(a), "v"(b)); }

// Fold the chunk that sits M chunks below the wave's first output row into its 8 accumulators: input row r of the chunk
// meets output row k at dy = 8M + r - k - 27.  Everything but the lane's column is a compile-time constant, so each pair
// is two ds_read_b32 with immediate offsets and one v_max3; pairs outside the footprint do not exist.
template <int M, int I = 0>
__device__ __forceinline__ void mf_fold(float (&acc)[ORW], const float *base) {
    if constexpr (I < CH * ORW) {
        constexpr int r = I / ORW, k = I % ORW, dy = M * CH + r - k - VR, ady = dy < 0 ? -dy : dy;
        if constexpr (ady <= VR) mf_max3(acc[k], base[r * SP + mf_offA(ady)], base[r * SP + mf_offB(ady)]);
        mf_fold<M, I + 1>(acc, base);
    }
}

__global__ __launch_bounds__(NT) void max_filter_tile(const float *__restrict__ in, float *__restrict__ out, MFGeom g) {
    __shared__ __attribute__((aligned(16))) float s_r[NSLOT * CH * SP];   // slices R_2 .. R_5 in slots 0 .. 3
    const int tid = threadIdx.x, cx = tid & 63;
    const float *base = s_r + cx;
    const int rg = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int X0 = g.ox0 + blockIdx.x * TW, Y0 = g.oy0 + g.lit_rows + blockIdx.y * TY;
    const float *inc = in + (long)blockIdx.z * g.in_sc;
    float acc[ORW];
#pragma unroll
    for (int k = 0; k < ORW; k++) acc[k] = -INFINITY;

    // The slices of a chunk are built in REGISTERS (round 5): threads 0..255 each own four adjacent staged pixels of one of the
    // chunk's CH rows (fetched one chunk ahead, straight from memory: R_0 never goes to LDS), the neighbours' pixels a window needs
    // come by DPP wave shifts — lane l + 1 holds the next four pixels, and lane 32 the first four of the next row, exactly the
    // words that follow in the LDS rows the per-entry builds read — and every slice is written once, 16 bytes per lane.  Entries
    // whose window runs past column 115 (or off the end of a wave) are garbage as before; no fold sample reads them (the samples end
    // at x + w <= column 115).  One LDS write per entry and slice instead of 4 + 3 x 2 reads and 5 writes per entry: the launch
    // is bound by LDS instructions, and the builds were 43 % of them; two workgroup barriers per chunk instead of six.
    static_assert(CH * SP / 4 == NT && SP == 128, "one quad per thread");
    const int qr = tid >> 5, qc = tid & 31;            // the thread's row of the chunk and quad of the row (threads 0..255)
    int qx[4];
#pragma unroll
    for (int j = 0; j < 4; j++) qx[j] = min(max(X0 - RAD + 4 * qc + j, g.ix0), g.ix0 + g.W - 1) - g.ix0;
    float nxt[4];
    auto fetch = [&](int chunk) {
        const int y = min(max(Y0 - VR + chunk * CH + qr, g.iy0), g.iy0 + g.H - 1) - g.iy0;
        const float *rowp = inc + (long)y * g.in_sy;
#pragma unroll
        for (int j = 0; j < 4; j++) nxt[j] = rowp[qx[j]];
    };
    auto nextlane = [](float v) {   // the value lane + 1 holds (0 in lane 63)
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
    };
    typedef float mf_f4 __attribute__((ext_vector_type(4)));
    fetch(0);
    for (int chunk = 0; chunk < NCHUNK; chunk++) {
        if (chunk > 0) __syncthreads();   // every wave has folded the previous chunk's slices
        {
            const float r0[4] = {nxt[0], nxt[1], nxt[2], nxt[3]};
            if (chunk + 1 < NCHUNK) fetch(chunk + 1);
            float n[4], r2[4], r3[4], r4[4], r5[4];
#pragma unroll
            for (int j = 0; j < 3; j++) n[j] = nextlane(r0[j]);
            {
                const float m01 = fmaxf(r0[0], r0[1]), m23 = fmaxf(r0[2], r0[3]), n01 = fmaxf(n[0], n[1]);
                r2[0] = fmaxf(m01, m23), r2[1] = fmaxf(fmaxf(r0[1], r0[2]), fmaxf(r0[3], n[0]));
                r2[2] = fmaxf(m23, n01), r2[3] = fmaxf(fmaxf(r0[3], n[0]), fmaxf(n[1], n[2]));
            }
#pragma unroll
            for (int j = 0; j < 4; j++) r3[j] = fmaxf(r2[j], nextlane(r2[j]));                                   // + 4 words
#pragma unroll
            for (int j = 0; j < 4; j++) r4[j] = fmaxf(r3[j], nextlane(nextlane(r3[j])));                         // + 8
#pragma unroll
            for (int j = 0; j < 4; j++) r5[j] = fmaxf(r4[j], nextlane(nextlane(nextlane(nextlane(r4[j])))));    // + 16
            float *w = s_r + qr * SP + 4 * qc;
            *reinterpret_cast<mf_f4 *>(w + 0 * CH * SP) = mf_f4{r2[0], r2[1], r2[2], r2[3]};
            *reinterpret_cast<mf_f4 *>(w + 1 * CH * SP) = mf_f4{r3[0], r3[1], r3[2], r3[3]};
            *reinterpret_cast<mf_f4 *>(w + 2 * CH * SP) = mf_f4{r4[0], r4[1], r4[2], r4[3]};
            *reinterpret_cast<mf_f4 *>(w + 3 * CH * SP) = mf_f4{r5[0], r5[1], r5[2], r5[3]};
        }
        __syncthreads();   // the slices of the chunk are complete
        switch (chunk - rg * (ORW / CH)) {   // wave-uniform
        case 0: mf_fold<0>(acc, base); break;
        case 1: mf_fold<1>(acc, base); break;
        case 2: mf_fold<2>(acc, base); break;
        case 3: mf_fold<3>(acc, base); break;
        case 4: mf_fold<4>(acc, base); break;
        case 5: mf_fold<5>(acc, base); break;
        case 6: mf_fold<6>(acc, base); break;
        case 7: mf_fold<7>(acc, base); break;
        case 8: mf_fold<8>(acc, base); break;
        default: break;
        }
        static_assert(NFOLD == 9, "one case per chunk of a wave's window");
    }
    const int x = blockIdx.x * TW + cx;
    if (x < g.ow) {
        float *o = out + (long)blockIdx.z * g.out_sc + x;
#pragma unroll
        for (int k = 0; k < ORW; k++) {
            const int y = g.lit_rows + blockIdx.y * TY + rg * ORW + k;
            if (y < g.oh) o[(long)y * g.out_sy] = acc[k];
        }
    }
}

const int64_t e0 = 0, ew = 1536, eh = 2560, ec = 3;
const int64_t *const est[6] = {&e0, &ew, &e0, &eh, &e0, &ec};
const halide_type_t ty_f32 = {(decltype(halide_type_t::code))2, 32, 0};
// estimates: generator :55-62
const halide_filter_argument_t mf_args[2] = {
    {"input", halide_argument_kind_input_buffer, 3, ty_f32, nullptr, nullptr, nullptr, nullptr, est},
    {"output", halide_argument_kind_output_buffer, 3, ty_f32, nullptr, nullptr, nullptr, nullptr, est},
};
const halide_filter_metadata_t mf_md = {1, 2, mf_args, kTargetString, "max_filter"};

}  // namespace

extern "C" int max_filter(halide_buffer_t *input, halide_buffer_t *output) {
    void *uc = nullptr;
    BufArg args[2] = {{"input", input, T_F32, 3, false}, {"output", output, T_F32, 3, true}};
    int r = check_not_null(uc, args, 2);
    if (r) return r;
    if ((r = check_type_and_dims(uc, args, 2))) return r;
    if (any_bounds_query(args, 2)) {
        // every x, y tap goes through repeat_edge (:17-19); the channel is read as asked
        int mins[3], ext[3];
        for (int d = 0; d < 3; d++) mins[d] = output->dim[d].min, ext[d] = output->dim[d].extent;
        answer_query(input, mins, ext);
        answer_query(output, mins, ext);
        return 0;
    }
    if ((r = check_shape(uc, args[0])) || (r = check_shape(uc, args[1]))) return r;
    const int ow = output->dim[0].extent, oh = output->dim[1].extent, oc = output->dim[2].extent;
    if ((r = check_covers(uc, args[0], 2, output->dim[2].min, oc))) return r;
    if (ow > 0 && oh > 0 && oc > 0) {
        if (input->dim[0].extent < 1 || input->dim[1].extent < 1) {
            return report(uc, halide_error_code_access_out_of_bounds, "Input buffer input is empty in x or y: nothing to clamp to");
        }
        if (input->dim[1].min != 0) {
            // the vert_log update runs over absolute rows -26 .. height-1 (:28); only with min 0 is the pipeline the
            // footprint max this kernel computes
            return report(uc, halide_error_code_constraint_violated, "Input buffer input: dimension 1 must start at 0");
        }
    }
    DeviceCtx ctx;
    if ((r = acquire_device(uc, &ctx))) return r;
    if ((r = input_to_device(uc, ctx, args[0]))) return r;
    if ((r = output_on_device(uc, ctx, args[1]))) return r;
    if (ow > 0 && oh > 0 && oc > 0) {
        MFGeom g;
        g.ix0 = input->dim[0].min, g.iy0 = input->dim[1].min, g.W = input->dim[0].extent, g.H = input->dim[1].extent;
        g.ox0 = output->dim[0].min, g.oy0 = output->dim[1].min, g.ow = ow, g.oh = oh;
        g.in_sy = input->dim[1].stride, g.in_sc = input->dim[2].stride;
        g.out_sy = output->dim[1].stride, g.out_sc = output->dim[2].stride;
        // filter_height (:47-49) and, from it, the half-width of every row of the footprint
        int h[RAD + 1];
        const float lim = (RAD + 0.25f) * (RAD + 0.25f);
        for (int dx = 0; dx <= RAD; dx++) {
            int n = 0;
            for (int dy = 0; dy <= RAD; dy++) n += ((float)(dx * dx + dy * dy) < lim) ? 1 : 0;
            h[dx] = n < VR ? n : VR;
        }
        for (int dx = 0; dx <= RAD; dx++) {
            g.t_of[dx] = h[dx];
            int sl = 0;
            while ((2 << sl) <= 2 * h[dx] + 1) sl++;     // slice_for_radius (:37), never above slices - 1 = 5
            g.s_of[dx] = sl;
        }
        g.lit_rows = output->dim[1].min < 0 ? (-output->dim[1].min < oh ? -output->dim[1].min : oh) : 0;
        const float *din = dev_ptr<float>(input) + (long)(output->dim[2].min - input->dim[2].min) * g.in_sc;
        timing_note_bytes(8.0 * ow * oh * oc);
        if (g.lit_rows > 0) {
            HLMI_LAUNCH(uc, "max_filter_literal", ctx.stream, max_filter_literal, dim3((ow + 255) / 256, g.lit_rows, oc), dim3(256), 0, din,
                        dev_ptr<float>(output), g);
        }
        if (oh > g.lit_rows) {
            HLMI_LAUNCH(uc, "max_filter_tile", ctx.stream, max_filter_tile, dim3((ow + TW - 1) / TW, (oh - g.lit_rows + TY - 1) / TY, oc),
                        dim3(NT), 0, din, dev_ptr<float>(output), g);
        }
    }
    mark_output_written(output);
    return 0;
}

extern "C" int max_filter_argv(void **a) { return max_filter((halide_buffer_t *)a[0], (halide_buffer_t *)a[1]); }
extern "C" const halide_filter_metadata_t *max_filter_metadata(void) { return &mf_md; }
extern "C" int max_filter_auto_schedule(halide_buffer_t *input, halide_buffer_t *output) { return max_filter(input, output); }
