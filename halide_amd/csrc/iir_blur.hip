// iir_blur.hip — gfx950 implementation of the reference's iir_blur AOT pipeline (first-order IIR low pass down/up the
// columns, then along the rows).  Algorithm: /root/reference/apps/iir_blur/iir_blur_generator.cpp:13-31,
// 146-156; boundary: `int iir_blur(halide_buffer_t *input, float alpha, halide_buffer_t *output)`, f32 [W,H,C] planar.
// The reference pins 1536 x 2560 x 3 (:158-163); this entry point takes any extents (superset) with mins 0.
//
// The scans are sequential BY DEFINITION (float recurrences do not re-associate), so the only parallelism is across
// columns and channels: one WAVE owns 64 adjacent columns of one channel.
//   forward   b = (1-a) b + a in(x, y), top to bottom, b written to a scratch plane (64 consecutive floats per row)
//   backward  b = (1-a) b + a scratch(x, y), bottom to top, 64 rows at a time into an LDS tile that is written out
//             TRANSPOSED (the generator transposes after each column blur, :31): lane = y, 64 consecutive floats per store
// Launched twice (columns of the input, then columns of the transposed intermediate = rows of the input).  The
// kernel is bound by the dependent multiply-add chain (2 (H + W) steps), not by HBM: the rows are requested 8 steps
// ahead of the chain.
#include "hlmi_internal.h"

using namespace hlmi;

namespace {

// src: [C][Hd][Wd] (row stride s_sy, plane stride s_sc); scratch: same shape, dense; dst: [C][Wd][Hd] (d_sy, d_sc)
__global__ __launch_bounds__(64) void iir_cols_T(const float *__restrict__ src, long s_sy, long s_sc, int Wd, int Hd, float alpha,
                                                float *__restrict__ scratch, float *__restrict__ dst, long d_sy, long d_sc) {
    __shared__ float tile[64 * 65];
    const int lane = threadIdx.x, x0 = blockIdx.x * 64, c = blockIdx.y;
    const int x = min(x0 + lane, Wd - 1);                    // lanes past the edge shadow the last column (never stored)
    const float c1 = 1.0f - alpha;
    const float *s = src + (long)c * s_sc + x;
    float *t = scratch + ((long)c * Hd) * Wd + x;
    float b = s[0];
    t[0] = b;
    int y = 1;
    for (; y + 8 <= Hd; y += 8) {                            // 8 rows requested before the chain consumes them
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = s[(long)(y + k) * s_sy];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            b = c1 * b + alpha * v[k];
            t[(long)(y + k) * Wd] = b;
        }
    }
    for (; y < Hd; y++) {
        b = c1 * b + alpha * s[(long)y * s_sy];
        t[(long)y * Wd] = b;
    }
    // backward, 64 rows per LDS tile, from the bottom; row Hd-1 keeps its forward value (:26-28 starts at Hd-2)
    for (int y1 = Hd - 1; y1 >= 0; y1 -= 64) {
        const int y0 = max(y1 - 63, 0), n = y1 - y0 + 1;
        for (int yy = y1; yy >= y0; yy -= 8) {
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; k++) v[k] = (yy - k >= y0) ? t[(long)(yy - k) * Wd] : 0.0f;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                if (yy - k >= y0) {
                    if (yy - k != Hd - 1) b = c1 * b + alpha * v[k];
                    tile[(yy - k - y0) * 65 + lane] = b;
                }
            }
        }
        __syncthreads();
        // transposed store: dst[c][x0 + j][y0 + lane]
        for (int j = 0; j < 64 && x0 + j < Wd; j++) {
            if (lane < n) dst[(long)c * d_sc + (long)(x0 + j) * d_sy + y0 + lane] = tile[lane * 65 + j];
        }
        __syncthreads();
    }
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
        HLMI_LAUNCH(uc, "iir_cols_T:1", st, iir_cols_T, dim3((W + 63) / 64, C), dim3(64), 0, dev_ptr<float>(input),
                    (long)input->dim[1].stride, (long)input->dim[2].stride, W, H, alpha, scratch, t1, (long)H, (long)W * H);
        timing_note_bytes(16.0 * W * H * C);
        // columns of t1 (= rows of the input) -> output [C][H rows of W]
        HLMI_LAUNCH(uc, "iir_cols_T:2", st, iir_cols_T, dim3((H + 63) / 64, C), dim3(64), 0, t1, (long)H, (long)W * H, H, W, alpha,
                    scratch, dev_ptr<float>(output), (long)output->dim[1].stride, (long)output->dim[2].stride);
    }
    mark_output_written(output);
    return 0;
}

extern "C" int iir_blur_argv(void **a) { return iir_blur((halide_buffer_t *)a[0], *(float *)a[1], (halide_buffer_t *)a[2]); }
extern "C" const halide_filter_metadata_t *iir_blur_metadata(void) { return &ib_md; }
extern "C" int iir_blur_auto_schedule(halide_buffer_t *input, float alpha, halide_buffer_t *output) { return iir_blur(input, alpha, output); }
