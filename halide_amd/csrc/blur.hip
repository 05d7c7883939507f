// blur.hip — gfx950 implementation of the reference's 3x3 box blur (apps/blur).
//
// Algorithm: /root/reference/apps/blur/halide_blur_generator.cpp:39-40, uint16 wrap-around arithmetic
// (src/IR.h:29-47), no boundary condition; boundary: `int halide_blur(halide_buffer_t *input,
// halide_buffer_t *blur_y)` (:31-32, :117).  HBM-bound: 2 B read + 2 B written per pixel.
// One thread produces a 4-wide x ROWS-tall strip with a sliding window of blur_x rows held in
// registers (the "slide" idea of the reference's GPU schedule, :77-94), so every input element is
// loaded once per strip plus a 2-row / 2-column halo.
#include "hlmi_internal.h"

using namespace hlmi;

namespace {

constexpr int VX = 4, ROWS = 8;

__device__ __forceinline__ void blur_x_row(const uint16_t *__restrict__ p, int navail, uint16_t bx[VX]) {
    // p[0 .. VX+1] needed; navail = how many of those elements exist (>= 1): beyond them the outputs are unused
    uint16_t v[VX + 2];
#pragma unroll
    for (int i = 0; i < VX + 2; i++) v[i] = i < navail ? p[i] : (uint16_t)0;
#pragma unroll
    for (int i = 0; i < VX; i++) bx[i] = (uint16_t)((uint16_t)((uint16_t)(v[i] + v[i + 1]) + v[i + 2]) / (uint16_t)3);
}

__global__ __launch_bounds__(256) void blur3x3_u16(const uint16_t *__restrict__ in, long in_sy, uint16_t *__restrict__ out,
                                                  long out_sy, int W, int H) {
    const int x0 = (blockIdx.x * 64 + (threadIdx.x & 63)) * VX;
    const int y0 = (blockIdx.y * 4 + (threadIdx.x >> 6)) * ROWS;
    if (x0 >= W || y0 >= H) return;
    const int navail = min(VX + 2, W + 2 - x0);
    const int nout = min(VX, W - x0);
    uint16_t r0[VX], r1[VX], r2[VX];
    blur_x_row(in + (long)y0 * in_sy + x0, navail, r0);
    blur_x_row(in + (long)(y0 + 1) * in_sy + x0, navail, r1);
    const int ny = min(ROWS, H - y0);
    for (int t = 0; t < ny; t++) {
        blur_x_row(in + (long)(y0 + t + 2) * in_sy + x0, navail, r2);
        uint16_t o[VX];
#pragma unroll
        for (int i = 0; i < VX; i++) {
            o[i] = (uint16_t)((uint16_t)((uint16_t)(r0[i] + r1[i]) + r2[i]) / (uint16_t)3);
            r0[i] = r1[i];
            r1[i] = r2[i];
        }
        uint16_t *op = out + (long)(y0 + t) * out_sy + x0;
        if (nout == VX && ((((uintptr_t)op) & 7) == 0)) {
            uint2 pk;
            pk.x = (uint32_t)o[0] | ((uint32_t)o[1] << 16);
            pk.y = (uint32_t)o[2] | ((uint32_t)o[3] << 16);
            *reinterpret_cast<uint2 *>(op) = pk;
        } else {
            for (int i = 0; i < nout; i++) op[i] = o[i];
        }
    }
}

const int64_t e0 = 0, ew = 1536, eh = 2560, ewi = 1538, ehi = 2562;
const int64_t *const est_in[4] = {&e0, &ewi, &e0, &ehi};
const int64_t *const est_out[4] = {&e0, &ew, &e0, &eh};
const halide_type_t ty_u16 = {(decltype(halide_type_t::code))1, 16, 0};
const halide_filter_argument_t blur_args[2] = {
    {"input", halide_argument_kind_input_buffer, 2, ty_u16, nullptr, nullptr, nullptr, nullptr, est_in},
    {"blur_y", halide_argument_kind_output_buffer, 2, ty_u16, nullptr, nullptr, nullptr, nullptr, est_out},
};
const halide_filter_metadata_t blur_md = {1, 2, blur_args, kTargetString, "halide_blur"};

}  // namespace

extern "C" int halide_blur(halide_buffer_t *input, halide_buffer_t *blur_y) {
    void *uc = nullptr;
    BufArg args[2] = {{"input", input, T_U16, 2, false}, {"blur_y", blur_y, T_U16, 2, true}};
    int r = check_not_null(uc, args, 2);
    if (r) return r;
    if ((r = check_type_and_dims(uc, args, 2))) return r;
    if (any_bounds_query(args, 2)) {
        // input is read on the output region grown by 2 in x and y (taps x..x+2, y..y+2; generator :39-40)
        int omin[2] = {blur_y->dim[0].min, blur_y->dim[1].min}, oext[2] = {blur_y->dim[0].extent, blur_y->dim[1].extent};
        int iext[2] = {oext[0] + 2, oext[1] + 2};
        answer_query(input, omin, iext);
        answer_query(blur_y, omin, oext);
        return 0;
    }
    if ((r = check_shape(uc, args[0])) || (r = check_shape(uc, args[1]))) return r;
    const int W = blur_y->dim[0].extent, H = blur_y->dim[1].extent;
    if ((r = check_covers(uc, args[0], 0, blur_y->dim[0].min, W + 2))) return r;
    if ((r = check_covers(uc, args[0], 1, blur_y->dim[1].min, H + 2))) return r;

    DeviceCtx ctx;
    if ((r = acquire_device(uc, &ctx))) return r;
    if ((r = input_to_device(uc, ctx, args[0]))) return r;
    if ((r = output_on_device(uc, ctx, args[1]))) return r;
    if (W > 0 && H > 0) {
        const long in_sy = input->dim[1].stride, out_sy = blur_y->dim[1].stride;
        const uint16_t *din = dev_ptr<uint16_t>(input) + (long)(blur_y->dim[1].min - input->dim[1].min) * in_sy +
                              (blur_y->dim[0].min - input->dim[0].min);
        dim3 grid((W + 64 * VX - 1) / (64 * VX), (H + 4 * ROWS - 1) / (4 * ROWS));
        HLMI_LAUNCH(uc, "blur3x3_u16", ctx.stream, blur3x3_u16, grid, dim3(256), 0, din, in_sy, dev_ptr<uint16_t>(blur_y),
                    out_sy, W, H);
    }
    mark_output_written(blur_y);
    return 0;
}

extern "C" int halide_blur_argv(void **a) { return halide_blur((halide_buffer_t *)a[0], (halide_buffer_t *)a[1]); }
extern "C" const halide_filter_metadata_t *halide_blur_metadata(void) { return &blur_md; }
extern "C" int halide_blur_auto_schedule(halide_buffer_t *input, halide_buffer_t *blur_y) { return halide_blur(input, blur_y); }
