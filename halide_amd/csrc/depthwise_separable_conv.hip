// depthwise_separable_conv.hip — gfx950 implementation of the reference's depthwise_separable_conv AOT pipeline
// (depthwise FWxFH convolution with zero padding, pointwise 1x1 convolution, bias, ReLU).
//
// Algorithm: /root/reference/apps/depthwise_separable_conv/depthwise_separable_conv_generator.cpp:24-75; boundary:
// `int depthwise_separable_conv(halide_buffer_t *input, halide_buffer_t *depthwise_filter, halide_buffer_t
// *pointwise_filter, halide_buffer_t *bias, halide_buffer_t *output)` (:11-23).  Layouts (dimension 0 innermost):
// input [CI, W, H, N], depthwise_filter [CM, IC', FW, FH] with stride(1) == CM (:283), pointwise_filter [CO, IC],
// bias [CO], output [CO, W, H, N].  The reference's driver runs MobileNet-v2's second layer: N=4, CI=32, CO=16, CM=1,
// 112x112, 3x3 (process.cpp:13).
//
// Arithmetic: every update of the two reductions is one fma in RDom order (rd fastest, then rx, ry; rc ascending
// from the bias) — the same canonical order as oracle/depthwise_separable_conv_oracle.c, so the result is bit-exact.
//
// The layer is HBM/launch bound (80 MFLOP against 6.4 MB in + 3.2 MB out at the driver's shape), so no matrix
// cores: a workgroup owns TP consecutive pixels of one image row; phase 1 computes the IC depthwise channels of each
// pixel into LDS (thread <-> (pixel, channel): channel-contiguous = coalesced input rows), phase 2 the CO outputs of
// each pixel from LDS (thread <-> (pixel, output channel)), with both filters staged in LDS once per workgroup.
#include "hlmi_internal.h"

#include <stdlib.h>

using namespace hlmi;

namespace {

constexpr int TP = 32;  // pixels of one output row per workgroup

struct DGeom {
    int CI, W, H, N, CM, FW, FH, IC, CO;
    int ix0, iy0;            // input mins in x, y (<= 0)
    int ox0, oy0, ow, oh;    // output region
    long in_sx, in_sy, in_sn, out_sx, out_sy, out_sn;
    long d_s1, d_sx, d_sy, p_s1;
};

__global__ __launch_bounds__(256) void dsc_fused(const float *__restrict__ in, const float *__restrict__ dw,
                                                const float *__restrict__ pw, const float *__restrict__ bias,
                                                float *__restrict__ out, DGeom g) {
    extern __shared__ float lds[];
    float *s_mid = lds;                              // [TP][IC]
    float *s_dw = s_mid + TP * g.IC;                 // [FH][FW][IC][CM]
    float *s_pw = s_dw + g.FH * g.FW * g.IC * g.CM;  // [IC][CO]
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * TP, y = blockIdx.y, n = blockIdx.z;
    for (int i = tid; i < g.FH * g.FW * g.IC * g.CM; i += 256) {
        const int rd = i % g.CM, d = (i / g.CM) % g.IC, rx = (i / (g.CM * g.IC)) % g.FW, ry = i / (g.CM * g.IC * g.FW);
        s_dw[i] = dw[rd + d * g.d_s1 + rx * g.d_sx + ry * g.d_sy];
    }
    for (int i = tid; i < g.IC * g.CO; i += 256) s_pw[i] = pw[(i % g.CO) + (long)(i / g.CO) * g.p_s1];
    __syncthreads();
    const int padw = g.FW / 2, padh = g.FH / 2;
    const int Y = g.oy0 + y;
    // ---- phase 1: depthwise_convolved(d, X, Y, n) for the tile's pixels
    for (int e = tid; e < TP * g.IC; e += 256) {
        const int px = e / g.IC, d = e - px * g.IC;
        const int X = g.ox0 + x0 + px;
        float acc = 0.0f;
        if (x0 + px < g.ow) {
            for (int ry = 0; ry < g.FH; ry++) {
                for (int rx = 0; rx < g.FW; rx++) {
                    const int xx = X + rx - padw, yy = Y + ry - padh;
                    // (:36-43): in_bounds tests against the EXTENTS, the read is clamped to [0, max]
                    const bool inb = xx >= 0 && xx < g.W && yy >= 0 && yy < g.H;
                    const int cx = min(max(xx, 0), g.ix0 + g.W - 1), cy = min(max(yy, 0), g.iy0 + g.H - 1);
                    const float v = inb ? in[(long)n * g.in_sn + (long)(cy - g.iy0) * g.in_sy + (long)(cx - g.ix0) * g.in_sx + d / g.CM] : 0.0f;
                    const float *f = s_dw + ((ry * g.FW + rx) * g.IC + d) * g.CM;
                    for (int rd = 0; rd < g.CM; rd++) acc = __builtin_fmaf(f[rd], v, acc);
                }
            }
        }
        s_mid[e] = acc;
    }
    __syncthreads();
    // ---- phase 2: pointwise + bias + ReLU
    for (int e = tid; e < TP * g.CO; e += 256) {
        const int px = e / g.CO, c = e - px * g.CO;
        if (x0 + px >= g.ow) continue;
        float acc = bias[c];
        const float *m = s_mid + px * g.IC;
        for (int rc = 0; rc < g.IC; rc++) acc = __builtin_fmaf(s_pw[rc * g.CO + c], m[rc], acc);
        out[(long)n * g.out_sn + (long)y * g.out_sy + (long)(x0 + px) * g.out_sx + c] = acc > 0.0f ? acc : 0.0f;
    }
}

// ---- dsc_fused_t: the same two phases with the filter shape, IC, CO compile-time (channel multiplier 1) and a wider tile.
// The generic kernel spends its time in run-time index arithmetic (five div / mod per staged filter element, per tap
// address chains) and in two LATENCY-bound fma chains per thread; here every index folds, a thread owns TPX IC / 256
// depthwise values and TPX CO / 256 outputs and walks their fma chains interleaved (independent accumulators), so that the
// dependent-issue latency of one chain hides behind the others.  Same operations in the same order: bit-identical.
// A32 (round 5, second pass): 32-bit addressing.  With `long` strides every one of the 63 input loads of a thread carried a
// 64-bit multiply-add chain (213 quarter-rate integer instructions + 82 64-bit adds of 890 VALU: 40 % of the launch's issue
// slots, and the launch was issue-bound: 3.5 waves per SIMD x ~1500 slots).  Here the 21 distinct column offsets of a thread
// (clamped column x pixel stride + channel, one 24-bit multiply each) are computed once and shared by the three rows, whose bases
// are scalar: a load is `row base (SGPRs) + lane byte offset`, no vector arithmetic; likewise filters and stores.  The host checks
// that strides are non-negative and below 2^23 and that every buffer spans less than 2^31 bytes.  Same fma chains, same bits.
__device__ __forceinline__ uint32_t dsc_mul24(uint32_t a, uint32_t b) {   // kept apart from a following add (else v_mad_u64_u32)
    uint32_t r = __umul24(a, b);
    asm("" : "+v"(r));
    return r;
}
template<int FW, int FH, int IC, int CO, int TPX, bool A32>
__global__ __launch_bounds__(256) void dsc_fused_t(const float *__restrict__ in, const float *__restrict__ dw, const float *__restrict__ pw,
                                                  const float *__restrict__ bias, float *__restrict__ out, DGeom g) {
    __shared__ float s_mid[TPX * IC], s_dw[FH * FW * IC], s_pw[IC * CO];
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * TPX, y = blockIdx.y, n = blockIdx.z;
    constexpr int padw = FW / 2, padh = FH / 2;
    const int Y = g.oy0 + y;
    // ---- phase 1: thread <-> (pixels px0 + k * (256 / IC), channel d)
    // Round 5: ONE memory round trip before the arithmetic instead of two.  The filters' elements and then every input value the
    // thread will multiply (FH FW N1 of them) are REQUESTED first; the filters go to LDS when they arrive (the inputs, requested
    // after them, are still in flight: a wave's loads return in order), the barrier follows, and the fma chains start on inputs that
    // arrived meanwhile.  Before, a workgroup waited for its filter elements, passed the barrier and only then asked for its pixels.
    constexpr int PPT = 256 / IC, N1 = (TPX + PPT - 1) / PPT;   // pixels per pass, passes
    constexpr int NDW = (FH * FW * IC + 255) / 256, NPW = (IC * CO + 255) / 256;
    float fdw[NDW], fpw[NPW];
#pragma unroll
    for (int j = 0; j < NDW; j++) {
        const int i = min(tid + 256 * j, FH * FW * IC - 1);
        if (A32) {   // i < 2^16: i / (IC FW) and i / IC as 24-bit multiplies by ceil(2^16 / n) would need a range proof per shape; IC is a power of two
            static_assert((IC & (IC - 1)) == 0 && FH * FW * IC < 4096, "dsc_fused_t<A32>: filter indexing");
            const uint32_t q = (uint32_t)i / IC, d = (uint32_t)i % IC, ry = (q * ((65536u + FW - 1) / FW)) >> 16, rx = q - ry * FW;   // q < 128: exact
            fdw[j] = dw[dsc_mul24(d, (uint32_t)g.d_s1) + dsc_mul24(rx, (uint32_t)g.d_sx) + dsc_mul24(ry, (uint32_t)g.d_sy)];
        } else {
            const int d = i % IC, rx = (i / IC) % FW, ry = i / (IC * FW);
            fdw[j] = dw[d * g.d_s1 + rx * g.d_sx + ry * g.d_sy];
        }
    }
#pragma unroll
    for (int j = 0; j < NPW; j++) {
        const int i = min(tid + 256 * j, IC * CO - 1);
        if (A32) fpw[j] = pw[(uint32_t)(i % CO) + dsc_mul24((uint32_t)(i / CO), (uint32_t)g.p_s1)];
        else fpw[j] = pw[(i % CO) + (long)(i / CO) * g.p_s1];
    }
    const int d = tid % IC, pxo = tid / IC;
    float v[FH][FW][N1];
    bool vin[FH][FW][N1];
    if (A32) {
        uint32_t xob[FW][N1];     // byte offset of (clamped column, channel d) in a row, per tap column and pass
        bool xin[FW][N1];
        const int xx0 = g.ox0 + x0 + pxo - padw;
#pragma unroll
        for (int rx = 0; rx < FW; rx++) {
#pragma unroll
            for (int k = 0; k < N1; k++) {
                const int xx = xx0 + rx + k * PPT;
                xin[rx][k] = xx >= 0 && xx < g.W;                                  // (:36-43): zero padding by the EXTENTS
                const int cx = min(max(xx, 0), g.ix0 + g.W - 1) - g.ix0;           // the read itself is clamped
                xob[rx][k] = (dsc_mul24((uint32_t)cx, (uint32_t)g.in_sx) + (uint32_t)d) << 2;
            }
        }
#pragma unroll
        for (int ry = 0; ry < FH; ry++) {
            const int yy = Y + ry - padh;
            const bool yin = yy >= 0 && yy < g.H;
            const char *rowp = reinterpret_cast<const char *>(in + (long)n * g.in_sn + (long)(min(max(yy, 0), g.iy0 + g.H - 1) - g.iy0) * g.in_sy);   // uniform
#pragma unroll
            for (int rx = 0; rx < FW; rx++) {
#pragma unroll
                for (int k = 0; k < N1; k++) {
                    v[ry][rx][k] = *reinterpret_cast<const float *>(rowp + xob[rx][k]);   // unconditional (the address is clamped); masked below
                    vin[ry][rx][k] = yin && xin[rx][k];
                }
            }
        }
    }
#pragma unroll
    for (int ry = 0; ry < (A32 ? 0 : FH); ry++) {
        const int yy = Y + ry - padh;
        const bool yin = yy >= 0 && yy < g.H;
        const float *rowp = in + (long)n * g.in_sn + (long)(min(max(yy, 0), g.iy0 + g.H - 1) - g.iy0) * g.in_sy + d;
#pragma unroll
        for (int rx = 0; rx < FW; rx++) {
#pragma unroll
            for (int k = 0; k < N1; k++) {
                const int px = pxo + k * PPT;
                const int xx = g.ox0 + x0 + px + rx - padw;
                const bool inb = yin && xx >= 0 && xx < g.W;                      // (:36-43): zero padding by the EXTENTS
                const int cx = min(max(xx, 0), g.ix0 + g.W - 1) - g.ix0;           // the read itself is clamped
                v[ry][rx][k] = inb ? rowp[(long)cx * g.in_sx] : 0.0f;
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < NDW; j++) {
        if (tid + 256 * j < FH * FW * IC) s_dw[tid + 256 * j] = fdw[j];
    }
#pragma unroll
    for (int j = 0; j < NPW; j++) {
        if (tid + 256 * j < IC * CO) s_pw[tid + 256 * j] = fpw[j];
    }
    __syncthreads();
    if (A32) {
        // every loaded value is USED here, in one place (an empty asm): left to itself the compiler turns `in bounds ? load : 0` into
        // a branch around each load with a wait behind it — 63 round trips in a row
#pragma unroll
        for (int ry = 0; ry < FH; ry++)
#pragma unroll
            for (int rx = 0; rx < FW; rx++)
#pragma unroll
                for (int k = 0; k < N1; k++) asm volatile("" : "+v"(v[ry][rx][k]));
#pragma unroll
        for (int ry = 0; ry < FH; ry++)
#pragma unroll
            for (int rx = 0; rx < FW; rx++)
#pragma unroll
                for (int k = 0; k < N1; k++) v[ry][rx][k] = vin[ry][rx][k] ? v[ry][rx][k] : 0.0f;
    }
    {
        float acc[N1];
#pragma unroll
        for (int k = 0; k < N1; k++) acc[k] = 0.0f;
#pragma unroll
        for (int ry = 0; ry < FH; ry++) {
#pragma unroll
            for (int rx = 0; rx < FW; rx++) {
                const float f = s_dw[(ry * FW + rx) * IC + d];
#pragma unroll
                for (int k = 0; k < N1; k++) acc[k] = __builtin_fmaf(f, v[ry][rx][k], acc[k]);   // pixels past the tile / the row: finite values nobody stores
            }
        }
#pragma unroll
        for (int k = 0; k < N1; k++) {
            const int px = pxo + k * PPT;
            if (px < TPX) s_mid[px * IC + d] = acc[k];
        }
    }
    __syncthreads();
    // ---- phase 2: thread <-> (pixels pxo + k * (256 / CO), output channel c)
    constexpr int QPT = 256 / CO, N2 = (TPX + QPT - 1) / QPT;
    {
        const int c = tid % CO, pxo = tid / CO;
        float acc[N2];
        const float b = bias[c];
#pragma unroll
        for (int k = 0; k < N2; k++) acc[k] = b;
#pragma unroll 8
        for (int rc = 0; rc < IC; rc++) {
            const float w = s_pw[rc * CO + c];
#pragma unroll
            for (int k = 0; k < N2; k++) {
                const int px = min(pxo + k * QPT, TPX - 1);
                acc[k] = __builtin_fmaf(w, s_mid[px * IC + rc], acc[k]);
            }
        }
#pragma unroll
        for (int k = 0; k < N2; k++) {
            const int px = pxo + k * QPT;
            if (px < TPX && x0 + px < g.ow) {
                const float rv = acc[k] > 0.0f ? acc[k] : 0.0f;
                if (A32) {
                    char *orow = reinterpret_cast<char *>(out + (long)n * g.out_sn + (long)y * g.out_sy);   // uniform
                    *reinterpret_cast<float *>(orow + ((dsc_mul24((uint32_t)(x0 + px), (uint32_t)g.out_sx) + (uint32_t)c) << 2)) = rv;
                } else {
                    out[(long)n * g.out_sn + (long)y * g.out_sy + (long)(x0 + px) * g.out_sx + c] = rv;
                }
            }
        }
    }
}

const halide_type_t ty_f32 = {(decltype(halide_type_t::code))2, 32, 0};
// estimates: generator :78-99 (N=4, CI=32, CO=16, CM=1, 112x112; depthwise_filter.dim(0) estimate is CI/CO as written)
const int64_t e0 = 0, e32 = 32, e112 = 112, e4 = 4, e2 = 2, e3 = 3, e16 = 16;
const int64_t *const est_in[8] = {&e0, &e32, &e0, &e112, &e0, &e112, &e0, &e4};
const int64_t *const est_dw[8] = {&e0, &e2, &e0, &e32, &e0, &e3, &e0, &e3};
const int64_t *const est_pw[4] = {&e0, &e16, &e0, &e32};
const int64_t *const est_b[2] = {&e0, &e16};
const int64_t *const est_o[8] = {&e0, &e16, &e0, &e112, &e0, &e112, &e0, &e4};
const halide_filter_argument_t dsc_args[5] = {
    {"input", halide_argument_kind_input_buffer, 4, ty_f32, nullptr, nullptr, nullptr, nullptr, est_in},
    {"depthwise_filter", halide_argument_kind_input_buffer, 4, ty_f32, nullptr, nullptr, nullptr, nullptr, est_dw},
    {"pointwise_filter", halide_argument_kind_input_buffer, 2, ty_f32, nullptr, nullptr, nullptr, nullptr, est_pw},
    {"bias", halide_argument_kind_input_buffer, 1, ty_f32, nullptr, nullptr, nullptr, nullptr, est_b},
    {"output", halide_argument_kind_output_buffer, 4, ty_f32, nullptr, nullptr, nullptr, nullptr, est_o},
};
const halide_filter_metadata_t dsc_md = {1, 5, dsc_args, kTargetString, "depthwise_separable_conv"};

}  // namespace

extern "C" int depthwise_separable_conv(halide_buffer_t *input, halide_buffer_t *depthwise_filter,
                                        halide_buffer_t *pointwise_filter, halide_buffer_t *bias, halide_buffer_t *output) {
    void *uc = nullptr;
    BufArg args[5] = {{"input", input, T_F32, 4, false}, {"depthwise_filter", depthwise_filter, T_F32, 4, false},
                      {"pointwise_filter", pointwise_filter, T_F32, 2, false}, {"bias", bias, T_F32, 1, false},
                      {"output", output, T_F32, 4, true}};
    int r = check_not_null(uc, args, 5);
    if (r) return r;
    if ((r = check_type_and_dims(uc, args, 5))) return r;
    auto real = [](halide_buffer_t *b) { return buffer_known(b); };   // real, or a query buffer the caller shaped
    if (any_bounds_query(args, 5)) {
        // shapes follow from whichever buffers are real or came shaped; the generator's estimates (:78-99, with CM = 1) fill the rest
        int ci = 32, w = 112, h = 112, n = 4, cm = 1, fw = 3, fh = 3, co = 16;
        if (real(input)) ci = input->dim[0].extent, w = input->dim[1].extent, h = input->dim[2].extent, n = input->dim[3].extent;
        else if (real(output)) w = output->dim[1].extent, h = output->dim[2].extent, n = output->dim[3].extent;
        // channel multiplier and filter size ARE the extents of depthwise_filter as passed (:27-29), also in query mode:
        // RunGen hands over query buffers shaped by the estimates (tools/RunGen.h:362-364), and :86 estimates CI / CO = 2
        if (real(depthwise_filter) || depthwise_filter->dim[0].extent > 0) cm = depthwise_filter->dim[0].extent;
        if (real(depthwise_filter) || depthwise_filter->dim[2].extent > 0) fw = depthwise_filter->dim[2].extent;
        if (real(depthwise_filter) || depthwise_filter->dim[3].extent > 0) fh = depthwise_filter->dim[3].extent;
        if (real(output)) co = output->dim[0].extent;
        else if (real(pointwise_filter)) co = pointwise_filter->dim[0].extent;
        else if (real(bias)) co = bias->dim[0].extent;
        const int ic = real(pointwise_filter) ? pointwise_filter->dim[1].extent : ci * cm;
        int z4[4] = {0, 0, 0, 0}, z2[2] = {0, 0}, z1[1] = {0};
        int ei[4] = {ci, w, h, n}, ed[4] = {cm, ic, fw, fh}, ep[2] = {co, ic}, eb[1] = {co}, eo[4] = {co, w, h, n};
        answer_query(input, z4, ei);
        answer_query(depthwise_filter, z4, ed);
        answer_query(pointwise_filter, z2, ep);
        answer_query(bias, z1, eb);
        answer_query(output, z4, eo);
        return 0;
    }
    for (int i = 0; i < 5; i++)
        if ((r = check_shape(uc, args[i]))) return r;
    DGeom g;
    g.CI = input->dim[0].extent, g.W = input->dim[1].extent, g.H = input->dim[2].extent, g.N = output->dim[3].extent;
    g.CM = depthwise_filter->dim[0].extent, g.FW = depthwise_filter->dim[2].extent, g.FH = depthwise_filter->dim[3].extent;
    g.IC = pointwise_filter->dim[1].extent, g.CO = output->dim[0].extent;
    // depthwise_filter.dim(1).stride == channel_multiplier (:283)
    if ((r = check_equal(uc, "depthwise_filter.stride.1", depthwise_filter->dim[1].stride, "depthwise_filter.extent.0", g.CM))) return r;
    // required regions (what the reference's bounds inference demands of each input; access_out_of_bounds otherwise)
    if ((r = check_covers(uc, args[1], 0, 0, g.CM)) || (r = check_covers(uc, args[1], 1, 0, g.IC)) ||
        (r = check_covers(uc, args[1], 2, 0, g.FW)) || (r = check_covers(uc, args[1], 3, 0, g.FH))) return r;
    if ((r = check_covers(uc, args[2], 0, output->dim[0].min, g.CO)) || (r = check_covers(uc, args[2], 1, 0, g.IC))) return r;
    if ((r = check_covers(uc, args[3], 0, output->dim[0].min, g.CO))) return r;
    if (g.CM > 0 && (r = check_covers(uc, args[0], 0, 0, (g.IC + g.CM - 1) / g.CM))) return r;
    if ((r = check_covers(uc, args[0], 3, output->dim[3].min, g.N))) return r;
    // clamp(x, 0, input.dim(1).max()) (:38-41) reads coordinate 0 whenever the image is not empty: the input must
    // start at or before 0 in x and y
    if (g.W > 0 && g.H > 0 && ((r = check_covers(uc, args[0], 1, 0, 1)) || (r = check_covers(uc, args[0], 2, 0, 1)))) return r;
    if (g.CM < 1 || g.FW < 1 || g.FH < 1) {
        return report(uc, halide_error_code_constraint_violated, "depthwise_filter extents must be >= 1");
    }
    DeviceCtx ctx;
    if ((r = acquire_device(uc, &ctx))) return r;
    for (int i = 0; i < 4; i++)
        if ((r = input_to_device(uc, ctx, args[i]))) return r;
    if ((r = output_on_device(uc, ctx, args[4]))) return r;
    g.ow = output->dim[1].extent, g.oh = output->dim[2].extent;
    if (g.ow > 0 && g.oh > 0 && g.N > 0 && g.CO > 0) {
        // in-bounds test of the generator (:36-37): 0 <= x < input.dim(1).extent(), i.e. W counts from the input's min
        g.ix0 = input->dim[1].min, g.iy0 = input->dim[2].min;
        g.ox0 = output->dim[1].min, g.oy0 = output->dim[2].min;
        g.in_sx = input->dim[1].stride, g.in_sy = input->dim[2].stride, g.in_sn = input->dim[3].stride;
        g.out_sx = output->dim[1].stride, g.out_sy = output->dim[2].stride, g.out_sn = output->dim[3].stride;
        g.d_s1 = depthwise_filter->dim[1].stride, g.d_sx = depthwise_filter->dim[2].stride, g.d_sy = depthwise_filter->dim[3].stride;
        g.p_s1 = pointwise_filter->dim[1].stride;
        const size_t sh = sizeof(float) * ((size_t)TP * g.IC + (size_t)g.FH * g.FW * g.IC * g.CM + (size_t)g.IC * g.CO);
        if (sh > 150 * 1024) {
            return report(uc, halide_error_code_constraint_violated,
                          "depthwise_separable_conv: filters of %zu bytes do not fit the 160 KB LDS of a CU", sh);
        }
        HLMI_HIP(uc, hipFuncSetAttribute(reinterpret_cast<const void *>(dsc_fused), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
        // pointers offset to the first element each kernel index addresses
        const float *d_in = dev_ptr<float>(input) + (long)(0 - input->dim[0].min) + (long)(output->dim[3].min - input->dim[3].min) * g.in_sn;
        const float *d_dw = dev_ptr<float>(depthwise_filter) + (long)(0 - depthwise_filter->dim[0].min) +
                            (long)(0 - depthwise_filter->dim[1].min) * g.d_s1 + (long)(0 - depthwise_filter->dim[2].min) * g.d_sx +
                            (long)(0 - depthwise_filter->dim[3].min) * g.d_sy;
        const float *d_pw = dev_ptr<float>(pointwise_filter) + (long)(output->dim[0].min - pointwise_filter->dim[0].min) +
                            (long)(0 - pointwise_filter->dim[1].min) * g.p_s1;
        const float *d_b = dev_ptr<float>(bias) + (long)(output->dim[0].min - bias->dim[0].min);
        timing_note_bytes(4.0 * ((double)g.CI * g.W * g.H * g.N + (double)g.CO * g.ow * g.oh * g.N));
        if (g.FW == 3 && g.FH == 3 && g.IC == 32 && g.CO == 16 && g.CM == 1) {
            constexpr int TPX = 56;   // the driver's MobileNet-v2 layer (process.cpp:13): two tiles per 112-pixel row
            // 32-bit addressing (see the kernel): small non-negative strides, every row / filter spanning less than 2^31 bytes
            const long lim = 1L << 23;
            auto ok = [&](long st) { return st >= 0 && st < lim; };
            const bool a32 = ok(g.in_sx) && ok(g.out_sx) && ok(g.d_s1) && ok(g.d_sx) && ok(g.d_sy) && ok(g.p_s1) && g.W < lim && g.ow < lim &&
                             (long)g.W * g.in_sx + g.IC < (1L << 29) && (long)g.ow * g.out_sx + g.CO < (1L << 29) &&
                             (long)g.IC * g.d_s1 + g.FW * g.d_sx + g.FH * g.d_sy < (1L << 29) && (long)g.IC * g.p_s1 + g.CO < (1L << 29) &&
                             !env_flag("HLMI_DSC_NO_A32");
            if (a32)
                HLMI_LAUNCH(uc, "dsc_fused", ctx.stream, (dsc_fused_t<3, 3, 32, 16, TPX, true>), dim3((g.ow + TPX - 1) / TPX, g.oh, g.N), dim3(256),
                            0, d_in, d_dw, d_pw, d_b, dev_ptr<float>(output), g);
            else
                HLMI_LAUNCH(uc, "dsc_fused", ctx.stream, (dsc_fused_t<3, 3, 32, 16, TPX, false>), dim3((g.ow + TPX - 1) / TPX, g.oh, g.N), dim3(256),
                            0, d_in, d_dw, d_pw, d_b, dev_ptr<float>(output), g);
        } else {
            dim3 grid((g.ow + TP - 1) / TP, g.oh, g.N);
            HLMI_LAUNCH(uc, "dsc_fused", ctx.stream, dsc_fused, grid, dim3(256), sh, d_in, d_dw, d_pw, d_b, dev_ptr<float>(output), g);
        }
    }
    mark_output_written(output);
    return 0;
}

extern "C" int depthwise_separable_conv_argv(void **a) {
    return depthwise_separable_conv((halide_buffer_t *)a[0], (halide_buffer_t *)a[1], (halide_buffer_t *)a[2],
                                    (halide_buffer_t *)a[3], (halide_buffer_t *)a[4]);
}
extern "C" const halide_filter_metadata_t *depthwise_separable_conv_metadata(void) { return &dsc_md; }
extern "C" int depthwise_separable_conv_auto_schedule(halide_buffer_t *input, halide_buffer_t *depthwise_filter,
                                                      halide_buffer_t *pointwise_filter, halide_buffer_t *bias,
                                                      halide_buffer_t *output) {
    return depthwise_separable_conv(input, depthwise_filter, pointwise_filter, bias, output);
}
