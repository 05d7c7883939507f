// unsharp.hip — gfx950 implementation of the reference's unsharp AOT pipeline (an adjacent app with the
// same boundary).  Algorithm: /root/reference/apps/unsharp/unsharp_generator.cpp:13-52 (sigma = 1.5, :7); boundary:
// `int unsharp(halide_buffer_t *input, halide_buffer_t *output)`, f32 [W,H,3] planar in and out (:9-10).
//
// gray -> 7-tap Gaussian down the columns -> 7-tap along the rows -> sharpen = 2 gray - blur -> ratio = sharpen / gray ->
// output = ratio * input.  Every operator rounds once in source order (the library is built with -ffp-contract=off), as
// oracle/unsharp_oracle.c fixes it; the four kernel taps are constants the reference's simplifier folds at compile time
// with the host's double exp (src/Simplify_Call.cpp:767-780) — computed the same way on the host here.
//
// HBM bound: 12 B/px read + 12 B/px written.  One workgroup = a 64 x 32 output tile: gray of the (64+6) x (32+6) window
// goes to LDS once (edge-clamped), blur_y of 70 x 32 to LDS, then each thread finishes 8 pixels (64 x 16 tiles: 5 % slower).
#include "hlmi_device_math.h"
#include "hlmi_internal.h"

#include <math.h>
#include <stdlib.h>

using namespace hlmi;

namespace {

constexpr int TW = 128, TH = 32, R = 3;
constexpr int GW = TW + 2 * R, GH = TH + 2 * R, GP = GW + 1;   // gray window and its LDS pitch

struct UGeom {
    int ix0, iy0, W, H;        // input origin and extents (clamp box)
    int ox0, oy0, ow, oh;      // output region
    long in_sy, in_sc, out_sy, out_sc;
    float k[4];
};

// ((k0 c + k1 s1) + k2 s2) + k3 s3 (:35-44): every product feeds an add (dev::mad2 / dev::mad fuse them under the fma canon)
__device__ __forceinline__ float tap7(const float (&k)[4], float c, float s1, float s2, float s3) {
    return dev::mad(k[3], s3, dev::mad(k[2], s2, dev::mad2(k[0], c, k[1], s1)));
}

__global__ __launch_bounds__(256) void unsharp_tile(const float *__restrict__ in, float *__restrict__ out, UGeom g) {
    __shared__ float s_gray[GH * GP];
    __shared__ float s_by[TH * GP];
    const int tid = threadIdx.x;
    const int X0 = g.ox0 + blockIdx.x * TW, Y0 = g.oy0 + blockIdx.y * TH;   // absolute origin of the tile
    {
        // every element of the thread is requested before the first one is used: left as a loop, each of the 11 iterations
        // waited for its own three loads — 11 memory round trips in a row at the head of every workgroup
        constexpr int N1 = (GW * GH + 255) / 256;
        float v[N1][3];
#pragma unroll
        for (int k = 0; k < N1; k++) {
            const int i = min(tid + 256 * k, GW * GH - 1);
            const int r = i / GW, c = i - r * GW;
            const int x = min(max(X0 - R + c, g.ix0), g.ix0 + g.W - 1) - g.ix0, y = min(max(Y0 - R + r, g.iy0), g.iy0 + g.H - 1) - g.iy0;
            const float *p = in + (long)y * g.in_sy + x;
            v[k][0] = p[0], v[k][1] = p[g.in_sc], v[k][2] = p[2 * g.in_sc];
        }
#pragma unroll
        for (int k = 0; k < N1; k++) {
            const int i = tid + 256 * k;
            if (i < GW * GH) {
                const int r = i / GW, c = i - r * GW;
                s_gray[r * GP + c] = dev::mad(0.114f, v[k][2], dev::mad2(0.299f, v[k][0], 0.587f, v[k][1]));
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < GW * TH; i += 256) {
        const int r = i / GW, c = i - r * GW;
        const float *q = s_gray + (r + R) * GP + c;
        s_by[r * GP + c] = tap7(g.k, q[0], q[-GP] + q[GP], q[-2 * GP] + q[2 * GP], q[-3 * GP] + q[3 * GP]);
    }
    __syncthreads();
    static_assert((TW * TH) % 256 == 0, "whole passes");
    constexpr int N3 = TW * TH / 256;
    float px[N3][3];
#pragma unroll
    for (int k = 0; k < N3; k++) {   // the pixels themselves, all requested first (pixels past the region re-read its last one)
        const int i = tid + 256 * k, r = i / TW, c = i - r * TW;
        const int x = min((int)blockIdx.x * TW + c, g.ow - 1), y = min((int)blockIdx.y * TH + r, g.oh - 1);
        const float *p = in + (long)(g.oy0 + y - g.iy0) * g.in_sy + (g.ox0 + x - g.ix0);
#pragma unroll
        for (int ch = 0; ch < 3; ch++) px[k][ch] = p[ch * g.in_sc];
    }
#pragma unroll
    for (int k = 0; k < N3; k++) {
        const int i = tid + 256 * k, r = i / TW, c = i - r * TW;
        const int x = blockIdx.x * TW + c, y = blockIdx.y * TH + r;
        if (x >= g.ow || y >= g.oh) continue;
        const float *q = s_by + r * GP + c + R;
        const float bx = tap7(g.k, q[0], q[-1] + q[1], q[-2] + q[2], q[-3] + q[3]);
        const float gr = s_gray[(r + R) * GP + c + R];
        const float ratio = (2.0f * gr - bx) / gr;
        float *o = out + (long)y * g.out_sy + x;
#pragma unroll
        for (int ch = 0; ch < 3; ch++) o[ch * g.out_sc] = ratio * px[k][ch];
    }
}

// unsharp_tile2 (round 5): the same tile, the same expressions, without the index arithmetic.  unsharp_tile is 1587 VALU
// instructions per thread of which 156 are quarter-rate integer multiplies and 178 64-bit adds (flat element numbers divided by
// the window width, `long` row products per element): ~2300 issue slots, in a launch whose workgroups all run in one round — load,
// compute, store in lock step — so the issue time adds to the memory time.  Here a thread owns a COLUMN: its clamped x offset is
// computed once, row numbers are wave-uniform (128 threads per row), so every global access is `scalar row pointer + lane byte
// offset` and LDS addresses are `lane base + constant`; the vertical 7-tap pass walks 16 rows of its column with a sliding window
// (22 LDS reads for 16 outputs instead of 112).  Rows narrower than 2^29 floats (the host checks; else unsharp_tile).
__global__ __launch_bounds__(256) void unsharp_tile2(const float *__restrict__ in, float *__restrict__ out, UGeom g) {
    static_assert(TW == 128 && TH == 32 && 2 * R <= 128, "unsharp_tile2: thread layout");
    __shared__ float s_gray[GH * GP];
    __shared__ float s_by[TH * GP];
    const int tid = threadIdx.x, cl = tid & 127;
    const int rh = __builtin_amdgcn_readfirstlane(tid >> 7);                // 0 / 1: wave-uniform
    const int X0 = g.ox0 + blockIdx.x * TW, Y0 = g.oy0 + blockIdx.y * TH;   // absolute origin of the tile
    auto gray_of = [](float r, float gg, float b) { return dev::mad(0.114f, b, dev::mad2(0.299f, r, 0.587f, gg)); };
    auto ld = [](const float *rowp, uint32_t byte_off) { return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(rowp) + byte_off); };
    {
        // window column cl (0..127) of rows rh, rh + 2, ...: all 3 x 19 loads requested before the first gray; columns 128..133 below
        constexpr int NR = GH / 2;
        static_assert(GH % 2 == 0, "row pairs");
        const uint32_t xb = (uint32_t)(min(max(X0 - R + cl, g.ix0), g.ix0 + g.W - 1) - g.ix0) << 2;
        float v[NR][3];
#pragma unroll
        for (int it = 0; it < NR; it++) {
            const int y = min(max(Y0 - R + 2 * it + rh, g.iy0), g.iy0 + g.H - 1) - g.iy0;   // scalar
            const float *rowp = in + (long)y * g.in_sy;
            v[it][0] = ld(rowp, xb), v[it][1] = ld(rowp + g.in_sc, xb), v[it][2] = ld(rowp + 2 * g.in_sc, xb);
        }
        // the six columns right of them: 6 x 38 elements, one per thread
        constexpr int NE = (GW - 128) * GH;
        static_assert(NE <= 256 && GW - 128 == 6, "edge columns");
        const int ei = min(tid, NE - 1), er = (ei * 10923) >> 16, ec = 128 + (ei - 6 * er);   // ei / 6 for ei < 228
        float ve[3];
        {
            const int x = min(max(X0 - R + ec, g.ix0), g.ix0 + g.W - 1) - g.ix0, y = min(max(Y0 - R + er, g.iy0), g.iy0 + g.H - 1) - g.iy0;
            const float *p = in + (long)y * g.in_sy + x;
            ve[0] = p[0], ve[1] = p[g.in_sc], ve[2] = p[2 * g.in_sc];
        }
#pragma unroll
        for (int it = 0; it < NR; it++) s_gray[(2 * it + rh) * GP + cl] = gray_of(v[it][0], v[it][1], v[it][2]);
        if (tid < NE) s_gray[er * GP + ec] = gray_of(ve[0], ve[1], ve[2]);
    }
    __syncthreads();
    {
        // blur_y: column cl, rows 16 rh .. 16 rh + 15 from gray rows 16 rh .. 16 rh + 21 of the column; then the six edge columns
        auto by_of = [&](float m3, float m2, float m1, float c0, float p1, float p2, float p3) {
            return tap7(g.k, c0, m1 + p1, m2 + p2, m3 + p3);
        };
        const float *q = s_gray + (16 * rh) * GP + cl;
        float w[22];
#pragma unroll
        for (int j = 0; j < 22; j++) w[j] = q[j * GP];
#pragma unroll
        for (int j = 0; j < 16; j++) s_by[(16 * rh + j) * GP + cl] = by_of(w[j], w[j + 1], w[j + 2], w[j + 3], w[j + 4], w[j + 5], w[j + 6]);
        constexpr int NE = (GW - 128) * TH;   // 192
        if (tid < NE) {
            const int er = (tid * 10923) >> 16, ec = 128 + (tid - 6 * er);
            const float *e = s_gray + (er + R) * GP + ec;
            s_by[er * GP + ec] = by_of(e[-3 * GP], e[-2 * GP], e[-GP], e[0], e[GP], e[2 * GP], e[3 * GP]);
        }
    }
    __syncthreads();
    constexpr int N3 = TW * TH / 256;   // 16 rows per thread: rh, rh + 2, ...
    const int xo = (int)blockIdx.x * TW + cl;
    uint32_t xib = (uint32_t)(g.ox0 + min(xo, g.ow - 1) - g.ix0) << 2, xob = (uint32_t)xo << 2;
    // (a lane offset defined in another basic block reaches the loads as a zero-extended 64-bit value and costs a 64-bit add per
    // access; redefined in place by an empty asm it stays 32-bit: scalar base + lane offset, no vector arithmetic)
    auto fresh = [](uint32_t &v) {
        asm volatile("" : "+v"(v));
        return v;
    };
    float px[N3][3];
#pragma unroll
    for (int k = 0; k < N3; k++) {   // the pixels themselves, all requested first (pixels past the region re-read its last one)
        const int y = min((int)blockIdx.y * TH + 2 * k + rh, g.oh - 1);   // scalar
        const float *rowp = in + (long)(g.oy0 + y - g.iy0) * g.in_sy;
#pragma unroll
        for (int ch = 0; ch < 3; ch++) px[k][ch] = ld(rowp + ch * g.in_sc, fresh(xib));
    }
#pragma unroll
    for (int k = 0; k < N3; k++) {
        const int r = 2 * k + rh, y = (int)blockIdx.y * TH + r;
        if (y >= g.oh) break;             // scalar
        const float *q = s_by + r * GP + cl + R;
        const float bx = tap7(g.k, q[0], q[-1] + q[1], q[-2] + q[2], q[-3] + q[3]);
        const float gr = s_gray[(r + R) * GP + cl + R];
        const float ratio = (2.0f * gr - bx) / gr;
        if (xo < g.ow) {
            float *orow = out + (long)y * g.out_sy;
#pragma unroll
            for (int ch = 0; ch < 3; ch++) *reinterpret_cast<float *>(reinterpret_cast<char *>(orow + ch * g.out_sc) + fresh(xob)) = ratio * px[k][ch];
        }
    }
}

const int64_t e0 = 0, ew = 1536, eh = 2560, ec = 3;
const int64_t *const est[6] = {&e0, &ew, &e0, &eh, &e0, &ec};
const halide_type_t ty_f32 = {(decltype(halide_type_t::code))2, 32, 0};
// estimates: generator :54-61
const halide_filter_argument_t us_args[2] = {
    {"input", halide_argument_kind_input_buffer, 3, ty_f32, nullptr, nullptr, nullptr, nullptr, est},
    {"output", halide_argument_kind_output_buffer, 3, ty_f32, nullptr, nullptr, nullptr, nullptr, est},
};
const halide_filter_metadata_t us_md = {1, 2, us_args, kTargetString, "unsharp"};

}  // namespace

extern "C" int unsharp(halide_buffer_t *input, halide_buffer_t *output) {
    void *uc = nullptr;
    BufArg args[2] = {{"input", input, T_F32, 3, false}, {"output", output, T_F32, 3, true}};
    int r = check_not_null(uc, args, 2);
    if (r) return r;
    if ((r = check_type_and_dims(uc, args, 2))) return r;
    if (any_bounds_query(args, 2)) {
        // the final tap input(x, y, c) is unclamped (:52): the input must cover the output's x, y region and channels 0..2
        int mins[3] = {output->dim[0].min, output->dim[1].min, 0}, ext[3] = {output->dim[0].extent, output->dim[1].extent, 3};
        answer_query(input, mins, ext);
        answer_query(output, mins, ext);
        return 0;
    }
    if ((r = check_shape(uc, args[0])) || (r = check_shape(uc, args[1]))) return r;
    const int ow = output->dim[0].extent, oh = output->dim[1].extent, oc = output->dim[2].extent;
    if ((r = check_covers(uc, args[0], 0, output->dim[0].min, ow)) || (r = check_covers(uc, args[0], 1, output->dim[1].min, oh))) return r;
    // gray reads channels 0, 1, 2 (through repeat_edge: any non-empty channel range is legal); the output's channels are
    // read unclamped
    if ((r = check_covers(uc, args[0], 2, output->dim[2].min, oc))) return r;
    if (output->dim[2].min < 0 || output->dim[2].min + oc > 3) {
        return report(uc, halide_error_code_constraint_violated, "Output buffer output: channels must lie in [0, 3)");
    }
    if (input->dim[2].min > 0 || input->dim[2].min + input->dim[2].extent < 3) {
        return report(uc, halide_error_code_constraint_violated, "Input buffer input must hold channels 0..2 (clamped channel reads are not supported)");
    }
    DeviceCtx ctx;
    if ((r = acquire_device(uc, &ctx))) return r;
    if ((r = input_to_device(uc, ctx, args[0]))) return r;
    if ((r = output_on_device(uc, ctx, args[1]))) return r;
    if (ow > 0 && oh > 0 && oc > 0) {
        UGeom g;
        g.ix0 = input->dim[0].min, g.iy0 = input->dim[1].min, g.W = input->dim[0].extent, g.H = input->dim[1].extent;
        g.ox0 = output->dim[0].min, g.oy0 = output->dim[1].min, g.ow = ow, g.oh = oh;
        g.in_sy = input->dim[1].stride, g.in_sc = input->dim[2].stride;
        g.out_sy = output->dim[1].stride, g.out_sc = output->dim[2].stride;
        const float kPi = 3.14159265358979310000f, sigma = 1.5f;
        const float den = sqrtf(2 * kPi) * sigma;
        for (int i = 0; i < 4; i++) g.k[i] = (float)exp((double)((float)(-i * i) / (2 * sigma * sigma))) / den;
        if (oc != 3 || output->dim[2].min != 0) {
            return report(uc, halide_error_code_constraint_violated, "Output buffer output: all three channels are produced together");
        }
        const float *din = dev_ptr<float>(input) + (long)(0 - input->dim[2].min) * g.in_sc;
        timing_note_bytes(24.0 * ow * oh);
        const bool narrow = g.W < (1 << 29) && ow < (1 << 29) && !env_flag("HLMI_UNSHARP_REF");   // 32-bit lane byte offsets inside a row
        if (narrow)
            HLMI_LAUNCH(uc, "unsharp_tile", ctx.stream, unsharp_tile2, dim3((ow + TW - 1) / TW, (oh + TH - 1) / TH), dim3(256), 0, din,
                        dev_ptr<float>(output), g);
        else
            HLMI_LAUNCH(uc, "unsharp_tile", ctx.stream, unsharp_tile, dim3((ow + TW - 1) / TW, (oh + TH - 1) / TH), dim3(256), 0, din,
                        dev_ptr<float>(output), g);
    }
    mark_output_written(output);
    return 0;
}

extern "C" int unsharp_argv(void **a) { return unsharp((halide_buffer_t *)a[0], (halide_buffer_t *)a[1]); }
extern "C" const halide_filter_metadata_t *unsharp_metadata(void) { return &us_md; }
extern "C" int unsharp_auto_schedule(halide_buffer_t *input, halide_buffer_t *output) { return unsharp(input, output); }
