// conv_layer.hip — gfx950 implementation of the reference's conv_layer AOT pipeline (3x3 conv + bias + ReLU).
//
// Algorithm: /root/reference/apps/conv_layer/conv_layer_generator.cpp:21-27; boundary: `int conv_layer(
// halide_buffer_t *input, halide_buffer_t *filter, halide_buffer_t *bias, halide_buffer_t *relu)` (:9-12, :207).
// Layouts (dimension 0 innermost, :35-50): input [CI, W+2, H+2, N], filter [CO, 3(kx), 3(ky), CI], bias [CO],
// relu [CO, W, H, N].  The reference pins N=5, CI=CO=128, W=100, H=80; this entry point takes any N, W, H and any
// CI, CO that are multiples of 32 (superset), still with dense strides.
//
// Exact f32 path (this file's `conv_layer`): implicit GEMM on the f32-input matrix cores.
//   D[pixel][co] = bias[co] + sum_k A[pixel][k] * B[k][co],  k = (ky*3 + kx)*CI + ci   (RDom order, ci fastest)
// `v_mfma_f32_32x32x2_f32` evaluates D = fma(a_k1, b_k1, fma(a_k0, b_k0, C)) — a k-ordered fmaf chain with one
// rounding per product (MI355X guide, §FP32-input MFMA) — so feeding k in RDom order reproduces the oracle's
// `acc = fmaf(filter, input, acc)` chain bit for bit, at the 157 TFLOP/s f32 matrix rate.
//   workgroup = 4 waves = 128 pixels x 128 output channels; wave = 2x2 accumulators of 32x32
//   per (ky, kx, 32-ci chunk): A (128 px x 32 ci, 128 B contiguous per pixel) and B (32 ci x 128 co, 512 B rows)
//   are staged in LDS k-major; operands are one f32 VGPR per lane: A[i = lane&31][k = lane>>5], B[k][j = lane&31].
//   Epilogue: ReLU, C/D map row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) -> pixel, col = lane&31 -> co: each
//   half-wave stores 128 contiguous bytes.
// bf16 MFMA path (`conv_layer_bf16`, BASELINE configs[4]): see conv_layer_bf16.hip.
#include "hlmi_internal.h"

using namespace hlmi;

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int TP = 128;        // pixels per workgroup
constexpr int TC = 128;        // output channels per workgroup
constexpr int KC = 32;         // ci per staged chunk
constexpr int AP = TP + 1;     // LDS pitch of A rows (k-major), odd -> conflict-light scalar stores
constexpr int BP = TC;         // LDS pitch of B rows (k-major), float4 stores

struct CGeom {
    int CI, CO, W, H, N;
    long npix;                 // W*H*N
};

__global__ __launch_bounds__(256) void conv3x3_f32_mfma(const float *__restrict__ in, const float *__restrict__ filt,
                                                       const float *__restrict__ bias, float *__restrict__ out, CGeom g) {
    __shared__ float sA[KC * AP];
    __shared__ float sB[KC * BP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;                  // wave tile: pixels [64*wm, +64), co [64*wn, +64)
    const long p0 = (long)blockIdx.x * TP;                    // first pixel of this workgroup
    const int co0 = blockIdx.y * TC;

    // loader role for A: pixel la_p (+32*i), float4 index la_q of the 32-ci chunk
    const int la_q = tid & 7, la_p = tid >> 3;
    long a_base[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        long p = p0 + la_p + 32 * i;
        if (p >= g.npix) p = g.npix - 1;                      // clamp: padded rows are computed but never stored
        int x = (int)(p % g.W);
        long t = p / g.W;
        int y = (int)(t % g.H), n = (int)(t / g.H);
        a_base[i] = (((long)n * (g.H + 2) + y) * (g.W + 2) + x) * g.CI;
    }
    // loader role for B: row lb_k (+8*i) of the chunk, float4 index lb_q of the 128 co
    const int lb_q = tid & 31, lb_k = tid >> 5;

    floatx16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++) {
            const float bv = bias[co0 + 64 * wn + 32 * b + (lane & 31)];  // C init = bias broadcast down the rows
#pragma unroll
            for (int r = 0; r < 16; r++) acc[a][b][r] = bv;
        }

    const long in_row = (long)(g.W + 2) * g.CI;
    const int nchunk = g.CI / KC;
#pragma unroll 1
    for (int kk = 0; kk < 9; kk++) {
        const int ky = kk / 3, kx = kk - 3 * ky;
        const long a_off = (long)ky * in_row + (long)kx * g.CI;
        const long b_off = (long)(kx + 3 * ky) * g.CO + co0;
#pragma unroll 1
        for (int ch = 0; ch < nchunk; ch++) {
            const int ci0 = ch * KC;
            // ---- stage A: sA[k][pixel]
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const float4 v = *reinterpret_cast<const float4 *>(in + a_base[i] + a_off + ci0 + 4 * la_q);
                const int px = la_p + 32 * i;
                sA[(4 * la_q + 0) * AP + px] = v.x;
                sA[(4 * la_q + 1) * AP + px] = v.y;
                sA[(4 * la_q + 2) * AP + px] = v.z;
                sA[(4 * la_q + 3) * AP + px] = v.w;
            }
            // ---- stage B: sB[k][co]
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int k = lb_k + 8 * i;
                const float4 v = *reinterpret_cast<const float4 *>(filt + (long)(ci0 + k) * 9 * g.CO + b_off + 4 * lb_q);
                *reinterpret_cast<float4 *>(&sB[k * BP + 4 * lb_q]) = v;
            }
            __syncthreads();
            // ---- 16 MFMA steps of K=2, k ascending
            const float *pa = sA + (lane >> 5) * AP + 64 * wm + (lane & 31);
            const float *pb = sB + (lane >> 5) * BP + 64 * wn + (lane & 31);
#pragma unroll
            for (int s = 0; s < KC / 2; s++) {
                const float a0 = pa[2 * s * AP], a1 = pa[2 * s * AP + 32];
                const float b0 = pb[2 * s * BP], b1 = pb[2 * s * BP + 32];
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            }
            __syncthreads();
        }
    }
    // ---- epilogue: relu = max(0, conv), store (co fastest)
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            const long p = p0 + 64 * wm + 32 * a + row;
            if (p < g.npix) {
#pragma unroll
                for (int b = 0; b < 2; b++) {
                    const float v = acc[a][b][r];
                    out[p * g.CO + co0 + 64 * wn + 32 * b + (lane & 31)] = v > 0.0f ? v : 0.0f;
                }
            }
        }
}

const halide_type_t ty_f32 = {(decltype(halide_type_t::code))2, 32, 0};
const int64_t e0 = 0, e128 = 128, e3 = 3, e102 = 102, e82 = 82, e100 = 100, e80 = 80, e5 = 5;
const int64_t *const est_in[8] = {&e0, &e128, &e0, &e102, &e0, &e82, &e0, &e5};
const int64_t *const est_f[8] = {&e0, &e128, &e0, &e3, &e0, &e3, &e0, &e128};
const int64_t *const est_b[2] = {&e0, &e128};
const int64_t *const est_o[8] = {&e0, &e128, &e0, &e100, &e0, &e80, &e0, &e5};
const halide_filter_argument_t conv_args[4] = {
    {"input", halide_argument_kind_input_buffer, 4, ty_f32, nullptr, nullptr, nullptr, nullptr, est_in},
    {"filter", halide_argument_kind_input_buffer, 4, ty_f32, nullptr, nullptr, nullptr, nullptr, est_f},
    {"bias", halide_argument_kind_input_buffer, 1, ty_f32, nullptr, nullptr, nullptr, nullptr, est_b},
    {"relu", halide_argument_kind_output_buffer, 4, ty_f32, nullptr, nullptr, nullptr, nullptr, est_o},
};
const halide_filter_metadata_t conv_md = {1, 4, conv_args, kTargetString, "conv_layer"};

}  // namespace

namespace hlmi {

// Shared by conv_layer and conv_layer_bf16: argument protocol + the dense-layout constraints of the generator
// (:35-50) generalised to runtime N, W, H, CI, CO.  Returns 0 and fills the extents, or an error code; *query is
// set when the call was a bounds query (answered here).
int conv_check_args(void *uc, BufArg *args, int *CI, int *CO, int *W, int *H, int *N, bool *query) {
    halide_buffer_t *input = args[0].buf, *filter = args[1].buf, *bias = args[2].buf, *relu = args[3].buf;
    int r = check_not_null(uc, args, 4);
    if (r) return r;
    if ((r = check_type_and_dims(uc, args, 4))) return r;
    *query = false;
    if (any_bounds_query(args, 4)) {
        // Shapes follow from whichever buffers are real or came shaped (the output first); the reference's estimates (:35-50) fill the rest.
        int co = 128, ci = 128, w = 100, h = 80, n = 5;
        auto real = [](halide_buffer_t *b) { return buffer_known(b); };   // real, or a query buffer the caller shaped
        if (real(relu)) co = relu->dim[0].extent, w = relu->dim[1].extent, h = relu->dim[2].extent, n = relu->dim[3].extent;
        else if (real(input)) w = input->dim[1].extent - 2, h = input->dim[2].extent - 2, n = input->dim[3].extent;
        if (real(input)) ci = input->dim[0].extent;
        else if (real(filter)) ci = filter->dim[3].extent;
        if (!real(relu) && real(filter)) co = filter->dim[0].extent;
        else if (!real(relu) && real(bias)) co = bias->dim[0].extent;
        int z4[4] = {0, 0, 0, 0}, z1[1] = {0};
        int ei[4] = {ci, w + 2, h + 2, n}, ef[4] = {co, 3, 3, ci}, eb[1] = {co}, eo[4] = {co, w, h, n};
        answer_query(input, z4, ei);
        answer_query(filter, z4, ef);
        answer_query(bias, z1, eb);
        answer_query(relu, z4, eo);
        *query = true;
        return 0;
    }
    for (int i = 0; i < 4; i++)
        if ((r = check_shape(uc, args[i]))) return r;
    const int co = relu->dim[0].extent, w = relu->dim[1].extent, h = relu->dim[2].extent, n = relu->dim[3].extent;
    const int ci = input->dim[0].extent;
    char what[64];
    // all mins are 0 and all strides dense, as the generator pins them (:35-50)
    for (int i = 0; i < 4; i++) {
        long dense = 1;
        for (int d = 0; d < args[i].buf->dimensions; d++) {
            snprintf(what, sizeof what, "%s.min.%d", args[i].name, d);
            if ((r = check_equal(uc, what, args[i].buf->dim[d].min, "0", 0))) return r;
            snprintf(what, sizeof what, "%s.stride.%d", args[i].name, d);
            if ((r = check_equal(uc, what, args[i].buf->dim[d].stride, "dense", (int)dense))) return r;
            dense *= args[i].buf->dim[d].extent;
        }
    }
    if ((r = check_equal(uc, "input.extent.1", input->dim[1].extent, "relu.extent.1 + 2", w + 2))) return r;
    if ((r = check_equal(uc, "input.extent.2", input->dim[2].extent, "relu.extent.2 + 2", h + 2))) return r;
    if ((r = check_equal(uc, "input.extent.3", input->dim[3].extent, "relu.extent.3", n))) return r;
    if ((r = check_equal(uc, "filter.extent.0", filter->dim[0].extent, "relu.extent.0", co))) return r;
    if ((r = check_equal(uc, "filter.extent.1", filter->dim[1].extent, "3", 3))) return r;
    if ((r = check_equal(uc, "filter.extent.2", filter->dim[2].extent, "3", 3))) return r;
    if ((r = check_equal(uc, "filter.extent.3", filter->dim[3].extent, "input.extent.0", ci))) return r;
    if ((r = check_equal(uc, "bias.extent.0", bias->dim[0].extent, "relu.extent.0", co))) return r;
    if (ci % 32 != 0 || co % 128 != 0) {
        return report(uc, halide_error_code_constraint_violated,
                      "Constraint violated: input channels (%d) must be a multiple of 32 and output channels (%d) of 128", ci, co);
    }
    *CI = ci, *CO = co, *W = w, *H = h, *N = n;
    return 0;
}

}  // namespace hlmi

extern "C" int conv_layer(halide_buffer_t *input, halide_buffer_t *filter, halide_buffer_t *bias, halide_buffer_t *relu) {
    void *uc = nullptr;
    BufArg args[4] = {{"input", input, T_F32, 4, false}, {"filter", filter, T_F32, 4, false}, {"bias", bias, T_F32, 1, false},
                      {"relu", relu, T_F32, 4, true}};
    CGeom g;
    bool query;
    int r = conv_check_args(uc, args, &g.CI, &g.CO, &g.W, &g.H, &g.N, &query);
    if (r || query) return r;
    DeviceCtx ctx;
    if ((r = acquire_device(uc, &ctx))) return r;
    for (int i = 0; i < 3; i++)
        if ((r = input_to_device(uc, ctx, args[i]))) return r;
    if ((r = output_on_device(uc, ctx, args[3]))) return r;
    g.npix = (long)g.W * g.H * g.N;
    if (g.npix > 0) {
        dim3 grid((unsigned)((g.npix + TP - 1) / TP), g.CO / TC);
        HLMI_LAUNCH(uc, "conv3x3_f32_mfma", ctx.stream, conv3x3_f32_mfma, grid, dim3(256), 0, dev_ptr<float>(input),
                    dev_ptr<float>(filter), dev_ptr<float>(bias), dev_ptr<float>(relu), g);
    }
    mark_output_written(relu);
    return 0;
}

extern "C" int conv_layer_argv(void **a) {
    return conv_layer((halide_buffer_t *)a[0], (halide_buffer_t *)a[1], (halide_buffer_t *)a[2], (halide_buffer_t *)a[3]);
}
extern "C" const halide_filter_metadata_t *conv_layer_metadata(void) { return &conv_md; }
extern "C" int conv_layer_auto_schedule(halide_buffer_t *input, halide_buffer_t *filter, halide_buffer_t *bias,
                                        halide_buffer_t *relu) {
    return conv_layer(input, filter, bias, relu);
}
