// harris.hip — gfx950 implementation of the reference's harris AOT pipeline (Harris corner response).
// Algorithm: /root/reference/apps/harris/harris_generator.cpp:7-62; boundary: `int harris(halide_buffer_t *input,
// halide_buffer_t *output)`, f32 [W,H,3] planar in, f32 [W-6,H-6] out with min (3,3) in the driver (filter.cpp:24-26).
// No boundary condition: the input must cover the output grown by 2 (error -4 otherwise).
//
// One workgroup = a 64 x 16 output tile: gray of the tile + 2 (LDS), the three gradient products of the tile + 1 (LDS),
// then the 3x3 sums and the response per pixel.  Sums are left to right as written in the generator, one rounding per
// operator (oracle/harris_oracle.c).  HBM: 12 B/px read, 4 B/px written.
#include "hlmi_device_math.h"
#include "hlmi_internal.h"

using namespace hlmi;

namespace {

constexpr int TW = 64, TH = 16;
constexpr int GW = TW + 4, GH = TH + 4, GP = GW + 1;   // gray window
constexpr int DW = TW + 2, DH = TH + 2, DP = DW + 1;   // gradient-product window

struct RGeom {
    int ix0, iy0, ox0, oy0, ow, oh;
    long in_sy, in_sc, out_sy;
};

__global__ __launch_bounds__(256) void harris_tile(const float *__restrict__ in, float *__restrict__ out, RGeom g) {
    __shared__ float s_g[GH * GP];
    __shared__ float s_xx[DH * DP], s_yy[DH * DP], s_xy[DH * DP];
    const int tid = threadIdx.x;
    const int X0 = g.ox0 + blockIdx.x * TW, Y0 = g.oy0 + blockIdx.y * TH;
    // the last tiles poke past the output region: clamp their reads to the rows / columns the region itself needs
    const int xmax = g.ox0 + g.ow + 1, ymax = g.oy0 + g.oh + 1;
    {
        // all of the thread's elements are requested before the first is used (as a loop, every iteration waited for its own
        // three loads: six memory round trips in a row at the head of every workgroup)
        constexpr int N1 = (GW * GH + 255) / 256;
        float v[N1][3];
#pragma unroll
        for (int k = 0; k < N1; k++) {
            const int i = min(tid + 256 * k, GW * GH - 1);
            const int r = i / GW, c = i - r * GW;
            const int x = min(X0 - 2 + c, xmax) - g.ix0, y = min(Y0 - 2 + r, ymax) - g.iy0;
            const float *p = in + (long)y * g.in_sy + x;
            v[k][0] = p[0], v[k][1] = p[g.in_sc], v[k][2] = p[2 * g.in_sc];
        }
#pragma unroll
        for (int k = 0; k < N1; k++) {
            const int i = tid + 256 * k;
            if (i < GW * GH) {
                const int r = i / GW, c = i - r * GW;
                s_g[r * GP + c] = dev::mad(0.114f, v[k][2], dev::mad2(0.299f, v[k][0], 0.587f, v[k][1]));
            }
        }
    }
    __syncthreads();
    const float a = -1.0f / 12, b = 1.0f / 12, c2 = -2.0f / 12, d = 2.0f / 12;
    for (int i = tid; i < DW * DH; i += 256) {
        const int r = i / DW, c = i - r * DW;
        const float *q = s_g + (r + 1) * GP + (c + 1);   // gray at (x, y) of this gradient sample
        // ((((g0 k0 + g1 k1) + g2 k2) + g3 k3) + g4 k4) + g5 k5
        auto d6 = [](float g0, float k0, float g1, float k1, float g2, float k2, float g3, float k3, float g4, float k4, float g5, float k5) {
            return dev::mad(g5, k5, dev::mad(g4, k4, dev::mad(g3, k3, dev::mad(g2, k2, dev::mad2(g0, k0, g1, k1)))));
        };
        const float iy = d6(q[-GP - 1], a, q[GP - 1], b, q[-GP], c2, q[GP], d, q[-GP + 1], a, q[GP + 1], b);
        const float ix = d6(q[-GP - 1], a, q[-GP + 1], b, q[-1], c2, q[1], d, q[GP - 1], a, q[GP + 1], b);
        if (dev::CANON_FMA) {
            // Ixx / Iyy / Ixy are inline in the reference's schedule (:111-123): under the fma canon their products are contracted into
            // the 3 x 3 sums, so the tile keeps the gradients and the sums multiply (s_xy is unused in this form)
            s_xx[r * DP + c] = ix, s_yy[r * DP + c] = iy;
        } else {
            s_xx[r * DP + c] = ix * ix, s_yy[r * DP + c] = iy * iy, s_xy[r * DP + c] = ix * iy;
        }
    }
    __syncthreads();
    for (int i = tid; i < TW * TH; i += 256) {
        const int r = i / TW, c = i - r * TW;
        const int x = blockIdx.x * TW + c, y = blockIdx.y * TH + r;
        if (x >= g.ow || y >= g.oh) continue;
        auto s3 = [&](const float *f) {   // sum3x3 (:7-11): x-1 column first, y ascending within a column
            const float *q = f + (r + 1) * DP + (c + 1);
            return (((((((q[-DP - 1] + q[-1]) + q[DP - 1]) + q[-DP]) + q[0]) + q[DP]) + q[-DP + 1]) + q[1]) + q[DP + 1];
        };
        auto s3p = [&](const float *f, const float *h) {   // sum3x3 of the inline product f h, fma canon: the first product is fused
            const float *q = f + (r + 1) * DP + (c + 1), *w = h + (r + 1) * DP + (c + 1);   // with the second, the others with the sum
            float acc = dev::mad2(q[-DP - 1], w[-DP - 1], q[-1], w[-1]);
            acc = dev::mad(q[DP - 1], w[DP - 1], acc), acc = dev::mad(q[-DP], w[-DP], acc), acc = dev::mad(q[0], w[0], acc);
            acc = dev::mad(q[DP], w[DP], acc), acc = dev::mad(q[-DP + 1], w[-DP + 1], acc), acc = dev::mad(q[1], w[1], acc);
            return dev::mad(q[DP + 1], w[DP + 1], acc);
        };
        const float sxx = dev::CANON_FMA ? s3p(s_xx, s_xx) : s3(s_xx), syy = dev::CANON_FMA ? s3p(s_yy, s_yy) : s3(s_yy);
        const float sxy = dev::CANON_FMA ? s3p(s_xx, s_yy) : s3(s_xy);
        const float det = dev::mulsub(sxx, syy, sxy * sxy), trace = sxx + syy;
        out[(long)y * g.out_sy + x] = dev::msub(det, 0.04f * trace, trace);
    }
}

const int64_t e0 = 0, e3 = 3, ew = 1536, eh = 2560, ew6 = 1530, eh6 = 2554, ec = 3;
const int64_t *const est_in[6] = {&e0, &ew, &e0, &eh, &e0, &ec};
const int64_t *const est_out[4] = {&e3, &ew6, &e3, &eh6};
const halide_type_t ty_f32 = {(decltype(halide_type_t::code))2, 32, 0};
// estimates: generator :64-72
const halide_filter_argument_t r_args[2] = {
    {"input", halide_argument_kind_input_buffer, 3, ty_f32, nullptr, nullptr, nullptr, nullptr, est_in},
    {"output", halide_argument_kind_output_buffer, 2, ty_f32, nullptr, nullptr, nullptr, nullptr, est_out},
};
const halide_filter_metadata_t r_md = {1, 2, r_args, kTargetString, "harris"};

}  // namespace

extern "C" int harris(halide_buffer_t *input, halide_buffer_t *output) {
    void *uc = nullptr;
    BufArg args[2] = {{"input", input, T_F32, 3, false}, {"output", output, T_F32, 2, true}};
    int r = check_not_null(uc, args, 2);
    if (r) return r;
    if ((r = check_type_and_dims(uc, args, 2))) return r;
    const int ow = output->dim[0].extent, oh = output->dim[1].extent;
    if (any_bounds_query(args, 2)) {
        int mins[3] = {output->dim[0].min - 2, output->dim[1].min - 2, 0}, ext[3] = {ow + 4, oh + 4, 3};
        answer_query(input, mins, ext);
        int omins[2] = {output->dim[0].min, output->dim[1].min}, oext[2] = {ow, oh};
        answer_query(output, omins, oext);
        return 0;
    }
    if ((r = check_shape(uc, args[0])) || (r = check_shape(uc, args[1]))) return r;
    if (ow > 0 && oh > 0) {
        if ((r = check_covers(uc, args[0], 0, output->dim[0].min - 2, ow + 4)) || (r = check_covers(uc, args[0], 1, output->dim[1].min - 2, oh + 4)) ||
            (r = check_covers(uc, args[0], 2, 0, 3))) return r;
    }
    DeviceCtx ctx;
    if ((r = acquire_device(uc, &ctx))) return r;
    if ((r = input_to_device(uc, ctx, args[0]))) return r;
    if ((r = output_on_device(uc, ctx, args[1]))) return r;
    if (ow > 0 && oh > 0) {
        RGeom g;
        g.ix0 = input->dim[0].min, g.iy0 = input->dim[1].min, g.ox0 = output->dim[0].min, g.oy0 = output->dim[1].min;
        g.ow = ow, g.oh = oh, g.in_sy = input->dim[1].stride, g.in_sc = input->dim[2].stride, g.out_sy = output->dim[1].stride;
        const float *din = dev_ptr<float>(input) + (long)(0 - input->dim[2].min) * g.in_sc;
        timing_note_bytes(16.0 * ow * oh);
        HLMI_LAUNCH(uc, "harris_tile", ctx.stream, harris_tile, dim3((ow + TW - 1) / TW, (oh + TH - 1) / TH), dim3(256), 0, din,
                    dev_ptr<float>(output), g);
    }
    mark_output_written(output);
    return 0;
}

extern "C" int harris_argv(void **a) { return harris((halide_buffer_t *)a[0], (halide_buffer_t *)a[1]); }
extern "C" const halide_filter_metadata_t *harris_metadata(void) { return &r_md; }
extern "C" int harris_auto_schedule(halide_buffer_t *input, halide_buffer_t *output) { return harris(input, output); }
