// hist.hip — gfx950 implementation of the reference's hist AOT pipeline (histogram equalisation; SURVEY.md §8 f3: an
// adjacent app with the same boundary).  Algorithm: /root/reference/apps/hist/hist_generator.cpp:13-56; boundary:
// `int hist(halide_buffer_t *input, halide_buffer_t *output)`, u8 [W,H,3] planar in and out (:9-10).
//
//   hist_count   luma histogram of the WHOLE input (:28-36).  Counts are integers — exact and order-free — so this is
//                where private copies belong: 32 copies of the 256-bin histogram per workgroup in LDS, one per lane
//                modulo 32 (ds_add_u32 without bank or address collisions, see hist_count), summed per workgroup and
//                added to the global histogram with one atomic per bin.  4 pixels per lane per step (aligned dword loads).
//                The global histogram itself is kept as 8 copies (workgroup mod 8): same-address atomics serialise in L2.
//   hist_apply   pointwise: luma, Cr, Cb, eq = clamp(float(cdf[bin]) * 255 / (W H)), recolour, u8 — one rounding per
//                operator in source order (oracle/hist_oracle.c); cdf in LDS; 4 pixels per lane
// HBM: 3 B/px read twice + 3 B/px written.
#include "hlmi_device_math.h"
#include "hlmi_internal.h"

#include <stdlib.h>

using namespace hlmi;

namespace {

__device__ __forceinline__ float luma(uint8_t r, uint8_t g, uint8_t b) {
    return dev::mad(0.114f, (float)b, dev::mad2(0.299f, (float)r, 0.587f, (float)g));
}

// in: channel 0 of the input's element (0, 0); W x H pixels.  VEC: rows start 4-byte aligned and W % 4 == 0
// Private copies of the histogram keep the LDS atomics of a wave apart: 32 copies per workgroup, laid out [bin][copy] with
// copy = lane % 32 — a lane's counter always sits in bank `copy`, whatever the bin, so the 64 ds_add_u32 of a wave instruction
// never collide on a bank by more than two (lanes l and l + 32), and only those two can hit the same address.  With one
// copy per wave a smooth image (neighbouring pixels in one bin) serialised all 64 lanes on one address.
constexpr int HCOPY = 32;
constexpr int HSUB = 8;    // copies of the global histogram
template<bool VEC>
__global__ __launch_bounds__(256) void hist_count(const uint8_t *__restrict__ in, long in_sy, long in_sc, int W, int H,
                                                 int rows_per_block, unsigned *__restrict__ ghist) {
    __shared__ unsigned wh[256 * HCOPY];
    const int tid = threadIdx.x;
    for (int i = tid; i < 256 * HCOPY; i += 256) wh[i] = 0;
    __syncthreads();
    unsigned *mine = wh + (tid & (HCOPY - 1));
    const int y0 = blockIdx.x * rows_per_block, y1 = min(y0 + rows_per_block, H);
    if (VEC) {
        // a thread's (row, 1024-pixel column block) slots, eight at a time with all 24 loads in flight: taken one by one, each
        // slot's loads were a full HBM round trip that nothing overlapped (16 us for 12 MB)
        const int xiters = (W + 1023) / 1024, slots = (y1 - y0) * xiters;
        for (int s0 = 0; s0 < slots; s0 += 8) {
            uint32_t a[8], b[8], c[8];
            bool ok[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int sl = s0 + k, r = sl / xiters, x = 4 * tid + 1024 * (sl - r * xiters);
                ok[k] = sl < slots && x < W;
                if (ok[k]) {
                    const uint8_t *r0 = in + (long)(y0 + r) * in_sy + x;
                    a[k] = *reinterpret_cast<const uint32_t *>(r0), b[k] = *reinterpret_cast<const uint32_t *>(r0 + in_sc),
                    c[k] = *reinterpret_cast<const uint32_t *>(r0 + 2 * in_sc);
                }
            }
#pragma unroll
            for (int k = 0; k < 8; k++) {
                if (!ok[k]) continue;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const float Y = luma((uint8_t)(a[k] >> (8 * j)), (uint8_t)(b[k] >> (8 * j)), (uint8_t)(c[k] >> (8 * j)));
                    atomicAdd(&mine[(int)dev::clampf(Y, 0.0f, 255.0f) * HCOPY], 1u);
                }
            }
        }
    } else {
        for (int y = y0; y < y1; y++) {
            const uint8_t *r0 = in + (long)y * in_sy, *r1 = r0 + in_sc, *r2 = r0 + 2 * in_sc;
            for (int x = tid; x < W; x += 256) atomicAdd(&mine[(int)dev::clampf(luma(r0[x], r1[x], r2[x]), 0.0f, 255.0f) * HCOPY], 1u);
        }
    }
    __syncthreads();
    unsigned s = 0;                                    // thread = bin; the copies are read rotated so that the 64 lanes of a
#pragma unroll 8                                       // wave read 32 different banks
    for (int k = 0; k < HCOPY; k++) s += wh[tid * HCOPY + ((k + tid) & (HCOPY - 1))];
    // HSUB copies of the global histogram (hist_apply adds them up): 80 workgroups per address instead of 640 — same-address
    // global atomics are serialised in L2 and were most of this kernel's time
    if (s) atomicAdd(&ghist[(blockIdx.x % HSUB) * 256 + tid], s);
}

struct HGeom {
    int ox0, oy0, ow, oh;
    long in_sy, in_sc, out_sy, out_sc;
    float scale;                              // 255.0f / float(H * W)
};

template<bool VEC>
__global__ __launch_bounds__(256) void hist_apply(const uint8_t *__restrict__ in, const unsigned *__restrict__ ghist,
                                                 uint8_t *__restrict__ out, HGeom g) {
    // the cdf is a 256-term integer prefix sum of the histogram (exact, order-free): every workgroup forms it itself from
    // the 1 KB histogram (L2-resident) instead of waiting for a one-workgroup hist_cdf launch in between
    __shared__ int s_cdf[256];
    __shared__ unsigned s_wsum[4];
    {
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        unsigned run = 0;
#pragma unroll
        for (int k = 0; k < HSUB; k++) run += ghist[k * 256 + threadIdx.x];
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned up = __shfl_up(run, d, 64);
            if (lane >= d) run += up;
        }
        if (lane == 63) s_wsum[wv] = run;
        __syncthreads();
        unsigned base = 0;
#pragma unroll
        for (int k = 0; k < 3; k++) base += (k < wv) ? s_wsum[k] : 0u;
        s_cdf[threadIdx.x] = (int)(base + run);
    }
    __syncthreads();
    const int y = blockIdx.y;
    const uint8_t *r0 = in + (long)(g.oy0 + y) * g.in_sy + g.ox0, *r1 = r0 + g.in_sc, *r2 = r0 + 2 * g.in_sc;
    uint8_t *o0 = out + (long)y * g.out_sy, *o1 = o0 + g.out_sc, *o2 = o0 + 2 * g.out_sc;
    auto pixel = [&](uint8_t rr, uint8_t gg, uint8_t bb, uint8_t &red, uint8_t &green, uint8_t &blue) {
        const float Y = luma(rr, gg, bb);
        const float Cr = dev::mad((float)rr - Y, 0.713f, 128.0f), Cb = dev::mad((float)bb - Y, 0.564f, 128.0f);
        const uint8_t bin = (uint8_t)dev::clampf(Y, 0.0f, 255.0f);
        const float eq = dev::clampf((float)s_cdf[bin] * g.scale, 0.0f, 255.0f);
        red = (uint8_t)dev::clampf(dev::mad(Cr - 128.0f, 1.4f, eq), 0.0f, 255.0f);
        green = (uint8_t)dev::clampf(dev::msub(dev::msub(eq, 0.343f, Cb - 128.0f), 0.711f, Cr - 128.0f), 0.0f, 255.0f);
        blue = (uint8_t)dev::clampf(dev::mad(1.765f, Cb - 128.0f, eq), 0.0f, 255.0f);
    };
    if (VEC) {
        const int x = 4 * (blockIdx.x * 256 + threadIdx.x);
        if (x >= g.ow) return;
        const uint32_t a = *reinterpret_cast<const uint32_t *>(r0 + x), b = *reinterpret_cast<const uint32_t *>(r1 + x),
                       c = *reinterpret_cast<const uint32_t *>(r2 + x);
        uint32_t pr = 0, pg = 0, pb = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            uint8_t red, green, blue;
            pixel((uint8_t)(a >> (8 * k)), (uint8_t)(b >> (8 * k)), (uint8_t)(c >> (8 * k)), red, green, blue);
            pr |= (uint32_t)red << (8 * k), pg |= (uint32_t)green << (8 * k), pb |= (uint32_t)blue << (8 * k);
        }
        *reinterpret_cast<uint32_t *>(o0 + x) = pr;
        *reinterpret_cast<uint32_t *>(o1 + x) = pg;
        *reinterpret_cast<uint32_t *>(o2 + x) = pb;
    } else {
        const int x = blockIdx.x * 256 + threadIdx.x;
        if (x >= g.ow) return;
        pixel(r0[x], r1[x], r2[x], o0[x], o1[x], o2[x]);
    }
}

const int64_t e0 = 0, ew = 1536, eh = 2560, ec = 3;
const int64_t *const est[6] = {&e0, &ew, &e0, &eh, &e0, &ec};
const halide_type_t ty_u8 = {(decltype(halide_type_t::code))1, 8, 0};
// estimates: generator :58-65
const halide_filter_argument_t h_args[2] = {
    {"input", halide_argument_kind_input_buffer, 3, ty_u8, nullptr, nullptr, nullptr, nullptr, est},
    {"output", halide_argument_kind_output_buffer, 3, ty_u8, nullptr, nullptr, nullptr, nullptr, est},
};
const halide_filter_metadata_t h_md = {1, 2, h_args, kTargetString, "hist"};

}  // namespace

extern "C" int hist(halide_buffer_t *input, halide_buffer_t *output) {
    void *uc = nullptr;
    BufArg args[2] = {{"input", input, T_U8, 3, false}, {"output", output, T_U8, 3, true}};
    int r = check_not_null(uc, args, 2);
    if (r) return r;
    if ((r = check_type_and_dims(uc, args, 2))) return r;
    if (any_bounds_query(args, 2)) {
        // every tap is unclamped and the histogram spans the input's own extent: propose the output's shape, 3 channels
        int mins[3] = {0, 0, 0}, ext[3] = {output->dim[0].min + output->dim[0].extent, output->dim[1].min + output->dim[1].extent, 3};
        answer_query(input, mins, ext);
        int omins[3] = {output->dim[0].min, output->dim[1].min, 0}, oext[3] = {output->dim[0].extent, output->dim[1].extent, 3};
        answer_query(output, omins, oext);
        return 0;
    }
    if ((r = check_shape(uc, args[0])) || (r = check_shape(uc, args[1]))) return r;
    const int W = input->dim[0].extent, H = input->dim[1].extent;
    const int ow = output->dim[0].extent, oh = output->dim[1].extent, oc = output->dim[2].extent;
    // the histogram's RDom starts at 0 (:31, :36) and reads channels 0..2
    if (W > 0 && H > 0) {
        if ((r = check_covers(uc, args[0], 0, 0, W)) || (r = check_covers(uc, args[0], 1, 0, H)) || (r = check_covers(uc, args[0], 2, 0, 3))) return r;
    }
    if ((r = check_covers(uc, args[0], 0, output->dim[0].min, ow)) || (r = check_covers(uc, args[0], 1, output->dim[1].min, oh))) return r;
    if (oc > 0 && (output->dim[2].min != 0 || oc != 3)) {
        return report(uc, halide_error_code_constraint_violated, "Output buffer output: the three channels [0, 3) are produced together");
    }
    DeviceCtx ctx;
    if ((r = acquire_device(uc, &ctx))) return r;
    if ((r = input_to_device(uc, ctx, args[0]))) return r;
    if ((r = output_on_device(uc, ctx, args[1]))) return r;
    if (ow > 0 && oh > 0 && oc > 0) {
        void *ws = nullptr;
        if ((r = get_workspace(uc, ctx, HSUB * 256 * sizeof(int), &ws))) return r;
        unsigned *ghist = (unsigned *)ws;
        hipStream_t st = ctx.stream;
        const long in_sy = input->dim[1].stride, in_sc = input->dim[2].stride;
        // element (0, 0, 0) of the input
        const uint8_t *din = dev_ptr<uint8_t>(input) + (long)(0 - input->dim[1].min) * in_sy + (0 - input->dim[0].min) +
                             (long)(0 - input->dim[2].min) * in_sc;
        HLMI_HIP(uc, hipMemsetAsync(ghist, 0, HSUB * 256 * sizeof(unsigned), st));
        const bool vec_in = (uintptr_t)din % 4 == 0 && in_sy % 4 == 0 && in_sc % 4 == 0;
        const int rpb = 4, nblk = (H + rpb - 1) / rpb;   // rows per block (2 / 8 / 16 measured slower)
        timing_note_bytes(3.0 * W * H);
        if (vec_in && W % 4 == 0) HLMI_LAUNCH(uc, "hist_count", st, hist_count<true>, dim3(nblk), dim3(256), 0, din, in_sy, in_sc, W, H, rpb, ghist);
        else HLMI_LAUNCH(uc, "hist_count", st, hist_count<false>, dim3(nblk), dim3(256), 0, din, in_sy, in_sc, W, H, rpb, ghist);
        HGeom g;
        g.ox0 = output->dim[0].min, g.oy0 = output->dim[1].min, g.ow = ow, g.oh = oh;
        g.in_sy = in_sy, g.in_sc = in_sc, g.out_sy = output->dim[1].stride, g.out_sc = output->dim[2].stride;
        g.scale = 255.0f / (float)(H * W);
        uint8_t *dout = dev_ptr<uint8_t>(output);
        const bool vec = vec_in && g.ox0 % 4 == 0 && ow % 4 == 0 && (uintptr_t)dout % 4 == 0 && g.out_sy % 4 == 0 && g.out_sc % 4 == 0;
        timing_note_bytes(6.0 * ow * oh);
        if (vec) HLMI_LAUNCH(uc, "hist_apply", st, hist_apply<true>, dim3((ow / 4 + 255) / 256, oh), dim3(256), 0, din, ghist, dout, g);
        else HLMI_LAUNCH(uc, "hist_apply", st, hist_apply<false>, dim3((ow + 255) / 256, oh), dim3(256), 0, din, ghist, dout, g);
    }
    mark_output_written(output);
    return 0;
}

extern "C" int hist_argv(void **a) { return hist((halide_buffer_t *)a[0], (halide_buffer_t *)a[1]); }
extern "C" const halide_filter_metadata_t *hist_metadata(void) { return &h_md; }
extern "C" int hist_auto_schedule(halide_buffer_t *input, halide_buffer_t *output) { return hist(input, output); }
