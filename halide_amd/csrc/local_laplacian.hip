// local_laplacian.hip — gfx950 implementation of the reference's local_laplacian AOT pipeline.
//
// Algorithm: /root/reference/apps/local_laplacian/local_laplacian_generator.cpp:18-87 (+ downsample
// :267-273, upsample :276-282); boundary: `int local_laplacian(halide_buffer_t*, int32_t, float, float,
// halide_buffer_t*)` (:12-16, :287).  pyramid_levels J = 8 is a compile-time GeneratorParam (:10).
//
// Layout in HBM.  All Funcs of the reference are total functions on Z^2 and only the input is
// edge-clamped (:28), so level j must be known a few pixels OUTSIDE ceil(W/2^j).  Because the clamp
// makes level 0 constant beyond the image edge, level j is constant beyond
//     lo_{j+1} = floor((lo_j - 2)/2),   hi_{j+1} = ceil((hi_j + 1)/2)      (lo_0, hi_0 = input min/max)
// so each level is stored on [lo_j, hi_j]^2 only and reads are clamped to that box — bit-identical
// to evaluating on the unbounded regions the reference's bounds inference demands.
//   level j (1..7):  float G[j][K+1][h_j][w_j]  — planes 0..K-1 = gPyramid[j](.,.,k), plane K = inGPyramid[j]
//                    float OUT[j][h_j][w_j]     — outGPyramid[j] on R_j
//   remap LUT: 2*(K-1)*256+1 floats, built on the device by ll_remap_lut.
// Launch chain (v1, one frame): lut, level0->1 (LDS tiled), 6x generic down, top, 6x generic up,
// level-0 collapse+recolour = 16 launches on one stream.
#include "hlmi_device_math.h"
#include "hlmi_internal.h"

using namespace hlmi;

namespace {

constexpr int J = 8;        // pyramid_levels (local_laplacian_generator.cpp:10)
constexpr int MAX_K = 32;   // largest `levels` the LUT sizing below admits

struct Level {
    int lox, loy, w, h;     // stored box: x in [lox, lox+w-1]
    int rx0, rx1, ry0, ry1; // R_j: region of outGPyramid[j] that is needed
    float *g;               // (K+1) planes
    float *out;             // outGPyramid[j]
};

struct Geometry {
    int K, half;            // levels, (K-1)*256
    float Km1, inv_Km1;
    int ix0, ix1, iy0, iy1; // clamp box of the input (absolute coordinates)
    int ic0, ic1;           // channel clamp box
};

// ---------------------------------------------------------------------------------------------------
// remap(i) = alpha * fx * exp(-fx*fx/2), fx = i/256  (generator :23-25)
__global__ void ll_remap_lut(float *lut, int half, float alpha) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > 2 * half) return;
    float fx = (float)(i - half) * (1.0f / 256.0f);
    lut[i] = (alpha * fx) * dev::halide_exp(((-fx) * fx) * 0.5f);
}

__device__ __forceinline__ float down4(float a, float b, float c, float d) {
    return ((a + 3.0f * (b + c)) + d) * 0.125f;  // (:270-271), "/ 8.0f" == "* 0.125f"
}

__device__ __forceinline__ float gray_of(const uint16_t *in, long sy, long sc, int x, int y, const Geometry &gm) {
    // clamped(x,y,c) / 65535.0f -> * (1/65535.0f); gray = 0.299 f0 + 0.587 f1 + 0.114 f2  (:28-36)
    int xc = dev::clampi(x, gm.ix0, gm.ix1) - gm.ix0, yc = dev::clampi(y, gm.iy0, gm.iy1) - gm.iy0;
    const float r = 1.0f / 65535.0f;
    long o = (long)yc * sy + xc;
    int c0 = dev::clampi(0, gm.ic0, gm.ic1) - gm.ic0, c1 = dev::clampi(1, gm.ic0, gm.ic1) - gm.ic0,
        c2 = dev::clampi(2, gm.ic0, gm.ic1) - gm.ic0;
    float f0 = (float)in[o + c0 * sc] * r, f1 = (float)in[o + c1 * sc] * r, f2 = (float)in[o + c2 * sc] * r;
    return (0.299f * f0 + 0.587f * f1) + 0.114f * f2;
}

// gPyramid[0](x,y,k) = beta*(gray - level) + level + remap(idx - 256*k)   (:41-44)
__device__ __forceinline__ float g0_of(float gray, int idx, int k, float beta, float inv_Km1, const float *lut, int half) {
    float level = (float)k * inv_Km1;
    return (beta * (gray - level) + level) + lut[idx - 256 * k + half];
}
__device__ __forceinline__ int idx_of(float gray, float Km1, int half) {
    return dev::clampi((int)((gray * Km1) * 256.0f), 0, half);
}

// ---------------------------------------------------------------------------------------------------
// level 0 -> level 1: gray + gPyramid[0] are never materialised (the reference's CPU schedule does the
// same, :184-188).  One workgroup produces a TX x TY tile of all K+1 planes of level 1.
constexpr int TX = 64, TY = 8, GW = 2 * TX + 2, GH = 2 * TY + 2, KC = 5;

template<bool LUT_IN_LDS>
__global__ __launch_bounds__(256) void ll_level0_down(const uint16_t *__restrict__ in, long in_sy, long in_sc,
                                                     Geometry gm, float beta, const float *__restrict__ lut_g,
                                                     float *__restrict__ g1, int lox, int loy, int w1, int h1) {
    extern __shared__ float smem[];
    float *sgray = smem;                       // GH*GW
    float *sdy = smem + GH * GW;               // KC*TY*GW
    float *slut = sdy + KC * TY * GW;          // 2*half+1 (only if LUT_IN_LDS)
    uint16_t *sidx = (uint16_t *)(slut + (LUT_IN_LDS ? 2 * gm.half + 1 : 0));  // GH*GW, idx <= 31*256
    const int tid = threadIdx.x;
    const int X0 = lox + blockIdx.x * TX, Y0 = loy + blockIdx.y * TY;  // level-1 coords of the tile origin
    const int gx0 = 2 * X0 - 1, gy0 = 2 * Y0 - 1;                      // level-0 coords of the window origin

    if (LUT_IN_LDS) {
        for (int i = tid; i <= 2 * gm.half; i += 256) slut[i] = lut_g[i];
    }
    for (int i = tid; i < GH * GW; i += 256) {
        int r = i / GW, c = i - r * GW;
        float gr = gray_of(in, in_sy, in_sc, gx0 + c, gy0 + r, gm);
        sgray[i] = gr;
        sidx[i] = (uint16_t)idx_of(gr, gm.Km1, gm.half);
    }
    __syncthreads();
    const float *lut = LUT_IN_LDS ? slut : lut_g;
    const size_t plane = (size_t)w1 * h1;

    for (int k0 = 0; k0 <= gm.K; k0 += KC) {
        const int nk = min(KC, gm.K + 1 - k0);
        // phase 1: vertical 1-3-3-1 on columns; thread <-> (plane kk, column c), walks the TY outputs
        for (int it = tid; it < nk * GW; it += 256) {
            int kk = it / GW, c = it - kk * GW;
            int k = k0 + kk;
            bool is_in = (k == gm.K);  // plane K: inGPyramid, gPyramid[0] replaced by gray itself (:58)
            float v0, v1, v2, v3;
            auto eval = [&](int r) -> float {
                float gr = sgray[r * GW + c];
                return is_in ? gr : g0_of(gr, sidx[r * GW + c], k, beta, gm.inv_Km1, lut, gm.half);
            };
            v0 = eval(0);
            v1 = eval(1);
#pragma unroll
            for (int t = 0; t < TY; t++) {
                v2 = eval(2 * t + 2);
                v3 = eval(2 * t + 3);
                sdy[(kk * TY + t) * GW + c] = down4(v0, v1, v2, v3);
                v0 = v2;
                v1 = v3;
            }
        }
        __syncthreads();
        // phase 2: horizontal 1-3-3-1, write level 1
        for (int it = tid; it < nk * TY * TX; it += 256) {
            int x = it % TX, t = (it / TX) % TY, kk = it / (TX * TY);
            int ox = X0 + x - lox, oy = Y0 + t - loy;
            if (ox < w1 && oy < h1) {
                const float *d = &sdy[(kk * TY + t) * GW + 2 * x];
                g1[(size_t)(k0 + kk) * plane + (size_t)oy * w1 + ox] = down4(d[0], d[1], d[2], d[3]);
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------
// generic level j -> j+1 (j >= 1): one thread per output element of one plane; 16 clamped taps
__global__ __launch_bounds__(256) void ll_down(const float *__restrict__ src, int slox, int sloy, int sw, int sh,
                                               float *__restrict__ dst, int dlox, int dloy, int dw, int dh) {
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= dw) return;
    const float *s = src + (size_t)blockIdx.z * sw * sh;
    int X = dlox + x, Y = dloy + y;
    int c[4], r[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        c[i] = dev::clampi(2 * X - 1 + i - slox, 0, sw - 1);
        r[i] = dev::clampi(2 * Y - 1 + i - sloy, 0, sh - 1);
    }
    float dy[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        dy[i] = down4(s[(size_t)r[0] * sw + c[i]], s[(size_t)r[1] * sw + c[i]], s[(size_t)r[2] * sw + c[i]],
                      s[(size_t)r[3] * sw + c[i]]);
    }
    dst[(size_t)blockIdx.z * dw * dh + (size_t)y * dw + x] = down4(dy[0], dy[1], dy[2], dy[3]);
}

// upsample(f)(X,Y) (:276-282) of a stored level plane `f` (origin lox/loy, width w)
__device__ __forceinline__ float up_at(const float *__restrict__ f, int lox, int loy, int w, int X, int Y) {
    int xa = dev::fdiv2(X + 1) - lox, xb = dev::fdiv2(X - 1) - lox;
    int ya = dev::fdiv2(Y + 1) - loy, yb = dev::fdiv2(Y - 1) - loy;
    float wx = (float)(dev::fmod2(X) * 2 + 1) * 0.25f, wy = (float)(dev::fmod2(Y) * 2 + 1) * 0.25f;
    float ua = dev::lerpf(f[(size_t)ya * w + xa], f[(size_t)ya * w + xb], wx);
    float ub = dev::lerpf(f[(size_t)yb * w + xa], f[(size_t)yb * w + xb], wx);
    return dev::lerpf(ua, ub, wy);
}

// outGPyramid[J-1] = outLPyramid[J-1] (:76, :63-72 with lPyramid[J-1] = gPyramid[J-1], :51)
__global__ void ll_top(const float *__restrict__ g, int w, int h, int lox, int loy, int rx0, int ry0, int rw, int rh,
                       int K, float Km1, float *__restrict__ out) {
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= rw || y >= rh) return;
    size_t o = (size_t)(ry0 + y - loy) * w + (rx0 + x - lox), plane = (size_t)w * h;
    float level = g[(size_t)K * plane + o] * Km1;
    int li = dev::clampi((int)level, 0, K - 2);
    float lf = level - (float)li;
    out[o] = (1.0f - lf) * g[(size_t)li * plane + o] + lf * g[(size_t)(li + 1) * plane + o];
}

// outGPyramid[j] = upsample(outGPyramid[j+1]) + outLPyramid[j], 1 <= j <= J-2 (:50-54, :63-79)
__global__ __launch_bounds__(256) void ll_up(const float *__restrict__ g, int w, int h, int lox, int loy,
                                             const float *__restrict__ gc, const float *__restrict__ outc, int cw,
                                             int ch, int clox, int cloy, int rx0, int ry0, int rw, int rh, int K,
                                             float Km1, float *__restrict__ out) {
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= rw || y >= rh) return;
    int X = rx0 + x, Y = ry0 + y;
    size_t o = (size_t)(Y - loy) * w + (X - lox), plane = (size_t)w * h, cplane = (size_t)cw * ch;
    float level = g[(size_t)K * plane + o] * Km1;
    int li = dev::clampi((int)level, 0, K - 2);
    float lf = level - (float)li;
    float l0 = g[(size_t)li * plane + o] - up_at(gc + (size_t)li * cplane, clox, cloy, cw, X, Y);
    float l1 = g[(size_t)(li + 1) * plane + o] - up_at(gc + (size_t)(li + 1) * cplane, clox, cloy, cw, X, Y);
    float outL = (1.0f - lf) * l0 + lf * l1;
    out[o] = up_at(outc, clox, cloy, cw, X, Y) + outL;
}

// level 0: outGPyramid[0], colour, u16 (:63-87).  gray / gPyramid[0] recomputed pointwise.
template<bool LUT_IN_LDS>
__global__ __launch_bounds__(256) void ll_level0_up(const uint16_t *__restrict__ in, long in_sy, long in_sc, Geometry gm,
                                                   float beta, const float *__restrict__ lut_g,
                                                   const float *__restrict__ g1, const float *__restrict__ out1, int w1,
                                                   int h1, int lox1, int loy1, uint16_t *__restrict__ out, long out_sy,
                                                   long out_sc, int ox0, int oy0, int ow, int oh, int oc0, int nc) {
    extern __shared__ float slut[];
    if (LUT_IN_LDS) {
        for (int i = threadIdx.x; i <= 2 * gm.half; i += 256) slut[i] = lut_g[i];
        __syncthreads();
    }
    const float *lut = LUT_IN_LDS ? slut : lut_g;
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= ow) return;
    int X = ox0 + x, Y = oy0 + y;
    float gray = gray_of(in, in_sy, in_sc, X, Y, gm);
    float level = gray * gm.Km1;
    int li = dev::clampi((int)level, 0, gm.K - 2);
    float lf = level - (float)li;
    int idx = idx_of(gray, gm.Km1, gm.half);
    size_t plane1 = (size_t)w1 * h1;
    float l0 = g0_of(gray, idx, li, beta, gm.inv_Km1, lut, gm.half) - up_at(g1 + (size_t)li * plane1, lox1, loy1, w1, X, Y);
    float l1 = g0_of(gray, idx, li + 1, beta, gm.inv_Km1, lut, gm.half) -
               up_at(g1 + (size_t)(li + 1) * plane1, lox1, loy1, w1, X, Y);
    float outL = (1.0f - lf) * l0 + lf * l1;
    float og = (up_at(out1, lox1, loy1, w1, X, Y) + outL) + 0.01f;
    float gr = gray + 0.01f;
    long io = (long)(Y - gm.iy0) * in_sy + (X - gm.ix0), oo = (long)y * out_sy + x;
    for (int c = 0; c < nc; c++) {
        // color = input * (outG0 + eps) / (gray + eps); input is the UNclamped input here (:84)
        float v = ((float)in[io + (long)(oc0 + c - gm.ic0) * in_sc] * og) / gr;
        out[oo + (long)c * out_sc] = (uint16_t)dev::clampf(v, 0.0f, 65535.0f);
    }
}

// ---------------------------------------------------------------------------------------------------
const int64_t est_zero = 0, est_w = 1536, est_h = 2560, est_c = 3;
const int64_t *const buf_est[6] = {&est_zero, &est_w, &est_zero, &est_h, &est_zero, &est_c};
const halide_scalar_value_t est_levels = [] { halide_scalar_value_t v{}; v.u.i32 = 8; return v; }();
const halide_scalar_value_t est_one = [] { halide_scalar_value_t v{}; v.u.f32 = 1.0f; return v; }();
const halide_type_t ty_u16 = {(decltype(halide_type_t::code))1, 16, 0};
const halide_type_t ty_i32 = {(decltype(halide_type_t::code))0, 32, 0};
const halide_type_t ty_f32 = {(decltype(halide_type_t::code))2, 32, 0};
// estimates: generator :92-98
const halide_filter_argument_t ll_args[5] = {
    {"input", halide_argument_kind_input_buffer, 3, ty_u16, nullptr, nullptr, nullptr, nullptr, buf_est},
    {"levels", halide_argument_kind_input_scalar, 0, ty_i32, nullptr, nullptr, nullptr, &est_levels, nullptr},
    {"alpha", halide_argument_kind_input_scalar, 0, ty_f32, nullptr, nullptr, nullptr, &est_one, nullptr},
    {"beta", halide_argument_kind_input_scalar, 0, ty_f32, nullptr, nullptr, nullptr, &est_one, nullptr},
    {"output", halide_argument_kind_output_buffer, 3, ty_u16, nullptr, nullptr, nullptr, nullptr, buf_est},
};
const halide_filter_metadata_t ll_md = {1, 5, ll_args, kTargetString, "local_laplacian"};

}  // namespace

extern "C" int local_laplacian(halide_buffer_t *input, int32_t levels, float alpha, float beta, halide_buffer_t *output) {
    void *uc = nullptr;
    BufArg args[2] = {{"input", input, T_U16, 3, false}, {"output", output, T_U16, 3, true}};
    int r = check_not_null(uc, args, 2);
    if (r) return r;
    if ((r = check_type_and_dims(uc, args, 2))) return r;
    if (any_bounds_query(args, 2)) {
        // input is needed exactly on the output's region: every other tap goes through repeat_edge (:28, :84)
        int mins[3], ext[3];
        for (int d = 0; d < 3; d++) mins[d] = output->dim[d].min, ext[d] = output->dim[d].extent;
        answer_query(input, mins, ext);
        answer_query(output, mins, ext);
        return 0;
    }
    if ((r = check_shape(uc, args[0])) || (r = check_shape(uc, args[1]))) return r;
    for (int d = 0; d < 3; d++) {
        if ((r = check_covers(uc, args[0], d, output->dim[d].min, output->dim[d].extent))) return r;
    }
    if (levels < 2 || levels > MAX_K) {
        return report(uc, levels < 2 ? halide_error_code_param_too_small : halide_error_code_param_too_large,
                      "Parameter levels is %d but must be in [2, %d]", levels, MAX_K);
    }
    const int ow = output->dim[0].extent, oh = output->dim[1].extent, nc = output->dim[2].extent;

    DeviceCtx ctx;
    if ((r = acquire_device(uc, &ctx))) return r;
    if ((r = input_to_device(uc, ctx, args[0]))) return r;
    if ((r = output_on_device(uc, ctx, args[1]))) return r;
    if (ow == 0 || oh == 0 || nc == 0) {
        mark_output_written(output);
        return 0;
    }

    Geometry gm;
    gm.K = levels;
    gm.half = (levels - 1) * 256;
    gm.Km1 = (float)(levels - 1);
    gm.inv_Km1 = 1.0f / gm.Km1;
    gm.ix0 = input->dim[0].min, gm.ix1 = gm.ix0 + input->dim[0].extent - 1;
    gm.iy0 = input->dim[1].min, gm.iy1 = gm.iy0 + input->dim[1].extent - 1;
    gm.ic0 = input->dim[2].min, gm.ic1 = gm.ic0 + input->dim[2].extent - 1;

    // per-level boxes
    Level lv[J];
    int lox = gm.ix0, hix = gm.ix1, loy = gm.iy0, hiy = gm.iy1;
    int rx0 = output->dim[0].min, rx1 = rx0 + ow - 1, ry0 = output->dim[1].min, ry1 = ry0 + oh - 1;
    size_t ws_floats = (size_t)(2 * gm.half + 1 + 63) & ~(size_t)63;
    size_t off[J], off_out[J];
    for (int j = 0; j < J; j++) {
        lv[j].lox = lox, lv[j].loy = loy, lv[j].w = hix - lox + 1, lv[j].h = hiy - loy + 1;
        lv[j].rx0 = rx0, lv[j].rx1 = rx1, lv[j].ry0 = ry0, lv[j].ry1 = ry1;
        if (j >= 1) {
            off[j] = ws_floats;
            ws_floats += ((size_t)(levels + 1) * lv[j].w * lv[j].h + 63) & ~(size_t)63;
            off_out[j] = ws_floats;
            ws_floats += ((size_t)lv[j].w * lv[j].h + 63) & ~(size_t)63;
        }
        lox = floor_div(lox - 2, 2), hix = floor_div(hix + 2, 2);
        loy = floor_div(loy - 2, 2), hiy = floor_div(hiy + 2, 2);
        rx0 = floor_div(rx0 - 1, 2), rx1 = floor_div(rx1 + 1, 2);
        ry0 = floor_div(ry0 - 1, 2), ry1 = floor_div(ry1 + 1, 2);
    }
    void *ws = nullptr;
    if ((r = get_workspace(uc, ctx, ws_floats * sizeof(float), &ws))) return r;
    float *wsf = (float *)ws;
    float *lut = wsf;
    for (int j = 1; j < J; j++) lv[j].g = wsf + off[j], lv[j].out = wsf + off_out[j];

    const uint16_t *din = dev_ptr<uint16_t>(input);
    uint16_t *dout = dev_ptr<uint16_t>(output);
    const long in_sy = input->dim[1].stride, in_sc = input->dim[2].stride;
    const long out_sy = output->dim[1].stride, out_sc = output->dim[2].stride;
    hipStream_t st = ctx.stream;
    const bool lut_lds = levels <= 15;  // LUT + tile must fit the default 64 KB dynamic-LDS window
    const int nlut = 2 * gm.half + 1;

    HLMI_LAUNCH(uc, "ll_remap_lut", st, ll_remap_lut, dim3((nlut + 255) / 256), dim3(256), 0, lut, gm.half, alpha);
    {
        dim3 grid((lv[1].w + TX - 1) / TX, (lv[1].h + TY - 1) / TY);
        size_t sh = sizeof(float) * (GH * GW + KC * TY * GW + (lut_lds ? nlut : 0)) + sizeof(uint16_t) * GH * GW;
        if (lut_lds) {
            HLMI_LAUNCH(uc, "ll_level0_down", st, ll_level0_down<true>, grid, dim3(256), sh, din, in_sy, in_sc, gm, beta,
                        lut, lv[1].g, lv[1].lox, lv[1].loy, lv[1].w, lv[1].h);
        } else {
            HLMI_LAUNCH(uc, "ll_level0_down", st, ll_level0_down<false>, grid, dim3(256), sh, din, in_sy, in_sc, gm, beta,
                        lut, lv[1].g, lv[1].lox, lv[1].loy, lv[1].w, lv[1].h);
        }
    }
    for (int j = 1; j + 1 < J; j++) {
        dim3 grid((lv[j + 1].w + 255) / 256, lv[j + 1].h, levels + 1);
        HLMI_LAUNCH(uc, "ll_down", st, ll_down, grid, dim3(256), 0, lv[j].g, lv[j].lox, lv[j].loy, lv[j].w, lv[j].h,
                    lv[j + 1].g, lv[j + 1].lox, lv[j + 1].loy, lv[j + 1].w, lv[j + 1].h);
    }
    {
        const Level &t = lv[J - 1];
        int rw = t.rx1 - t.rx0 + 1, rh = t.ry1 - t.ry0 + 1;
        HLMI_LAUNCH(uc, "ll_top", st, ll_top, dim3((rw + 63) / 64, rh), dim3(64), 0, t.g, t.w, t.h, t.lox, t.loy, t.rx0,
                    t.ry0, rw, rh, levels, gm.Km1, t.out);
    }
    for (int j = J - 2; j >= 1; j--) {
        const Level &a = lv[j], &c = lv[j + 1];
        int rw = a.rx1 - a.rx0 + 1, rh = a.ry1 - a.ry0 + 1;
        HLMI_LAUNCH(uc, "ll_up", st, ll_up, dim3((rw + 255) / 256, rh), dim3(256), 0, a.g, a.w, a.h, a.lox, a.loy, c.g,
                    c.out, c.w, c.h, c.lox, c.loy, a.rx0, a.ry0, rw, rh, levels, gm.Km1, a.out);
    }
    {
        dim3 grid((ow + 255) / 256, oh);
        size_t sh = lut_lds ? sizeof(float) * nlut : 0;
        const Level &c = lv[1];
        if (lut_lds) {
            HLMI_LAUNCH(uc, "ll_level0_up", st, ll_level0_up<true>, grid, dim3(256), sh, din, in_sy, in_sc, gm, beta, lut,
                        c.g, c.out, c.w, c.h, c.lox, c.loy, dout, out_sy, out_sc, output->dim[0].min, output->dim[1].min,
                        ow, oh, output->dim[2].min, nc);
        } else {
            HLMI_LAUNCH(uc, "ll_level0_up", st, ll_level0_up<false>, grid, dim3(256), sh, din, in_sy, in_sc, gm, beta, lut,
                        c.g, c.out, c.w, c.h, c.lox, c.loy, dout, out_sy, out_sc, output->dim[0].min, output->dim[1].min,
                        ow, oh, output->dim[2].min, nc);
        }
    }
    mark_output_written(output);
    return 0;
}

extern "C" int local_laplacian_argv(void **a) {
    return local_laplacian((halide_buffer_t *)a[0], *(int32_t *)a[1], *(float *)a[2], *(float *)a[3],
                           (halide_buffer_t *)a[4]);
}
extern "C" const halide_filter_metadata_t *local_laplacian_metadata(void) { return &ll_md; }
extern "C" int local_laplacian_auto_schedule(halide_buffer_t *input, int32_t levels, float alpha, float beta,
                                             halide_buffer_t *output) {
    return local_laplacian(input, levels, alpha, beta, output);
}
