// conv_layer_bf16.hip — bf16 matrix-core variant of the conv_layer pipeline (BASELINE.json configs[4]:
// N=16, CI=CO=128, H=W=56, k=3, "bf16 MFMA im2col-GEMM").
//
// Same algorithm, boundary and layouts as conv_layer.hip (reference: /root/reference/apps/conv_layer/
// conv_layer_generator.cpp:21-27, :35-50): f32 halide buffers input [CI, W+2, H+2, N], filter [CO, 3, 3, CI],
// bias [CO], relu [CO, W, H, N].  Only the arithmetic differs: input and filter are rounded to bfloat16
// (round-to-nearest-even, v_cvt_pk_bf16_f32), products are exact in f32 and are accumulated in f32 by
// `v_mfma_f32_32x32x16_bf16`, starting from the f32 bias.  Parity is therefore by tolerance against
// oracle_conv_layer_bf16 (same rounding of the operands, double accumulation), not bit-exact; the exact f32
// entry point is `conv_layer`.
//
// Implicit GEMM  D[pixel][co] = bias[co] + sum_k A[pixel][k] B[k][co],  k = (ky, kx, ci):
//   conv_filter_bf16   filter f32 [ci][ky][kx][co] -> bf16 wB[ky*3+kx][co][ci] (k contiguous per output channel,
//                      the order an MFMA B operand wants: lane j = column, 8 consecutive k)
//   conv3x3_bf16_mfma  workgroup = 4 waves = 128 pixels x 128 output channels, wave = 64 x 64 (2x2 accumulators of
//                      32x32).  Per (ky, kx) and 64-ci chunk: A (128 px x 64 ci, 256 B contiguous per pixel, f32 ->
//                      bf16 on the way) and B (128 co x 64 ci bf16) are staged in LDS with rows padded to 144 B;
//                      operands are read with ds_read_b128 (lane l: row l&31, k = 8 (l>>5) .. +7).  The next
//                      chunk's global loads are issued before the current chunk's 16 MFMAs (register double
//                      buffer + two LDS buffers, one barrier per chunk).
//   Algorithmic bytes: input + output once (f32) + the bf16 filter: 27.6 + 25.7 + 0.3 MB at configs[4];
//   flops 2 N H W CI CO 9 = 14.8 G.
#include "hlmi_internal.h"

using namespace hlmi;

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int TP = 128;       // pixels per workgroup
constexpr int TC = 128;       // output channels per workgroup
constexpr int KC = 64;        // ci per staged chunk
constexpr int PA = KC + 8;    // LDS row pitch in bf16 elements (144 B: ds_read_b128 rows fall on distinct banks)

struct CGeom {
    int CI, CO, W, H, N;
    long npix;
};

__device__ __forceinline__ uint32_t pk_bf16(float lo, float hi) {  // {bf16(lo), bf16(hi)}, round to nearest even
    uint32_t r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}

// filter f32 [ci][ky][kx][co] (co fastest) -> bf16 wB[kk][co][ci]
__global__ void conv_filter_bf16(const float *__restrict__ filt, uint16_t *__restrict__ wb, int CI, int CO) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;  // one thread per (kk, co, ci pair)
    const int half = CI / 2, total = 9 * CO * half;
    if (e >= total) return;
    const int cp = e % half, co = (e / half) % CO, kk = e / (half * CO);
    const float a = filt[((size_t)(2 * cp) * 9 + kk) * CO + co], b = filt[((size_t)(2 * cp + 1) * 9 + kk) * CO + co];
    reinterpret_cast<uint32_t *>(wb)[((size_t)kk * CO + co) * half + cp] = pk_bf16(a, b);
}

__global__ __launch_bounds__(256, 2) void conv3x3_bf16_mfma(const float *__restrict__ in, const uint16_t *__restrict__ wb,
                                                           const float *__restrict__ bias, float *__restrict__ out, CGeom g) {
    extern __shared__ uint16_t smem[];                       // [2][A: TP x PA | B: TC x PA] bf16
    constexpr int BUF = (TP + TC) * PA;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const long p0 = (long)blockIdx.x * TP;
    const int co0 = blockIdx.y * TC;

    // loader roles.  A: float4 index aq of the 64-ci chunk, pixels ap + 16 i.  B: 16-byte piece bq, rows bp + 32 i.
    const int aq = tid & 15, ap = tid >> 4;
    const int bq = tid & 7, bp = tid >> 3;
    long a_base[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        long p = p0 + ap + 16 * i;
        if (p >= g.npix) p = g.npix - 1;                     // padded rows are computed but never stored
        const int x = (int)(p % g.W);
        const long t = p / g.W;
        const int y = (int)(t % g.H), n = (int)(t / g.H);
        a_base[i] = (((long)n * (g.H + 2) + y) * (g.W + 2) + x) * g.CI + 4 * aq;
    }
    const long in_row = (long)(g.W + 2) * g.CI;
    const int cpk = g.CI / KC, nchunk = 9 * cpk;             // chunks per (ky, kx); total

    float4 ra[8];
    uint4 rb[4];
    auto gload = [&](int c) {
        const int kk = c / cpk, ci0 = (c - kk * cpk) * KC;
        const int ky = kk / 3, kx = kk - 3 * ky;
        const long a_off = (long)ky * in_row + (long)kx * g.CI + ci0;
#pragma unroll
        for (int i = 0; i < 8; i++) ra[i] = *reinterpret_cast<const float4 *>(in + a_base[i] + a_off);
        const uint16_t *bsrc = wb + ((size_t)kk * g.CO + co0) * g.CI + ci0 + 8 * bq;
#pragma unroll
        for (int i = 0; i < 4; i++) rb[i] = *reinterpret_cast<const uint4 *>(bsrc + (size_t)(bp + 32 * i) * g.CI);
    };
    auto lstore = [&](int buf) {
        uint16_t *sA = smem + buf * BUF, *sB = sA + TP * PA;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            uint2 v;
            v.x = pk_bf16(ra[i].x, ra[i].y), v.y = pk_bf16(ra[i].z, ra[i].w);
            *reinterpret_cast<uint2 *>(sA + (ap + 16 * i) * PA + 4 * aq) = v;
        }
#pragma unroll
        for (int i = 0; i < 4; i++) *reinterpret_cast<uint4 *>(sB + (bp + 32 * i) * PA + 8 * bq) = rb[i];
    };

    floatx16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++) {
            const float bv = bias[co0 + 64 * wn + 32 * b + (lane & 31)];  // C init = bias broadcast down the rows
#pragma unroll
            for (int r = 0; r < 16; r++) acc[a][b][r] = bv;
        }

    gload(0);
    lstore(0);
    __syncthreads();
#pragma unroll 1
    for (int c = 0; c < nchunk; c++) {
        if (c + 1 < nchunk) gload(c + 1);
        const uint16_t *sA = smem + (c & 1) * BUF, *sB = sA + TP * PA;
        const uint16_t *pa = sA + (64 * wm + (lane & 31)) * PA + 8 * (lane >> 5);
        const uint16_t *pb = sB + (64 * wn + (lane & 31)) * PA + 8 * (lane >> 5);
#pragma unroll
        for (int ks = 0; ks < KC / 16; ks++) {
            const bf16x8 a0 = *reinterpret_cast<const bf16x8 *>(pa + 16 * ks);
            const bf16x8 a1 = *reinterpret_cast<const bf16x8 *>(pa + 32 * PA + 16 * ks);
            const bf16x8 b0 = *reinterpret_cast<const bf16x8 *>(pb + 16 * ks);
            const bf16x8 b1 = *reinterpret_cast<const bf16x8 *>(pb + 32 * PA + 16 * ks);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (c + 1 < nchunk) lstore((c + 1) & 1);             // the other buffer: last read before the previous barrier
        __syncthreads();
    }
    // ---- epilogue: relu = max(0, conv), store (co fastest); C/D map: row = (r&3) + 8 (r>>2) + 4 (lane>>5), col = lane&31
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
// estimates: BASELINE.json configs[4] (N=16, 56x56 output, 128 -> 128 channels)
const int64_t e0 = 0, e128 = 128, e3 = 3, e58 = 58, e56 = 56, e16 = 16;
const int64_t *const est_in[8] = {&e0, &e128, &e0, &e58, &e0, &e58, &e0, &e16};
const int64_t *const est_f[8] = {&e0, &e128, &e0, &e3, &e0, &e3, &e0, &e128};
const int64_t *const est_b[2] = {&e0, &e128};
const int64_t *const est_o[8] = {&e0, &e128, &e0, &e56, &e0, &e56, &e0, &e16};
const halide_filter_argument_t conv_args[4] = {
    {"input", halide_argument_kind_input_buffer, 4, ty_f32, nullptr, nullptr, nullptr, nullptr, est_in},
    {"filter", halide_argument_kind_input_buffer, 4, ty_f32, nullptr, nullptr, nullptr, nullptr, est_f},
    {"bias", halide_argument_kind_input_buffer, 1, ty_f32, nullptr, nullptr, nullptr, nullptr, est_b},
    {"relu", halide_argument_kind_output_buffer, 4, ty_f32, nullptr, nullptr, nullptr, nullptr, est_o},
};
const halide_filter_metadata_t conv_md = {1, 4, conv_args, kTargetString, "conv_layer_bf16"};

}  // namespace

extern "C" int conv_layer_bf16(halide_buffer_t *input, halide_buffer_t *filter, halide_buffer_t *bias, halide_buffer_t *relu) {
    void *uc = nullptr;
    BufArg args[4] = {{"input", input, T_F32, 4, false}, {"filter", filter, T_F32, 4, false}, {"bias", bias, T_F32, 1, false},
                      {"relu", relu, T_F32, 4, true}};
    CGeom g;
    bool query;
    int r = conv_check_args(uc, args, &g.CI, &g.CO, &g.W, &g.H, &g.N, &query);
    if (r || query) return r;
    if (g.CI % KC != 0) {
        return report(uc, halide_error_code_constraint_violated,
                      "Constraint violated: conv_layer_bf16 needs input channels (%d) to be a multiple of %d", g.CI, KC);
    }
    DeviceCtx ctx;
    if ((r = acquire_device(uc, &ctx))) return r;
    for (int i = 0; i < 3; i++)
        if ((r = input_to_device(uc, ctx, args[i]))) return r;
    if ((r = output_on_device(uc, ctx, args[3]))) return r;
    g.npix = (long)g.W * g.H * g.N;
    if (g.npix > 0) {
        void *ws = nullptr;
        const size_t wb_bytes = (size_t)9 * g.CO * g.CI * sizeof(uint16_t);
        if ((r = get_workspace(uc, ctx, wb_bytes, &ws))) return r;
        uint16_t *wb = (uint16_t *)ws;
        const int pairs = 9 * g.CO * (g.CI / 2);
        timing_note_bytes(6.0 * 9 * g.CO * g.CI);
        HLMI_LAUNCH(uc, "conv_filter_bf16", ctx.stream, conv_filter_bf16, dim3((pairs + 255) / 256), dim3(256), 0,
                    dev_ptr<float>(filter), wb, g.CI, g.CO);
        const size_t sh = (size_t)2 * (TP + TC) * PA * sizeof(uint16_t);  // 73.7 KB: above the 64 KB default window
        HLMI_HIP(uc, hipFuncSetAttribute(reinterpret_cast<const void *>(conv3x3_bf16_mfma),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
        dim3 grid((unsigned)((g.npix + TP - 1) / TP), g.CO / TC);
        timing_note_bytes(4.0 * ((double)g.N * (g.H + 2) * (g.W + 2) * g.CI + (double)g.npix * g.CO) + (double)wb_bytes);
        HLMI_LAUNCH(uc, "conv3x3_bf16_mfma", ctx.stream, conv3x3_bf16_mfma, grid, dim3(256), sh, dev_ptr<float>(input), wb,
                    dev_ptr<float>(bias), dev_ptr<float>(relu), g);
    }
    mark_output_written(relu);
    return 0;
}

extern "C" int conv_layer_bf16_argv(void **a) {
    return conv_layer_bf16((halide_buffer_t *)a[0], (halide_buffer_t *)a[1], (halide_buffer_t *)a[2], (halide_buffer_t *)a[3]);
}
extern "C" const halide_filter_metadata_t *conv_layer_bf16_metadata(void) { return &conv_md; }
