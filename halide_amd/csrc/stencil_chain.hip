// stencil_chain.hip — gfx950 implementation of the reference's stencil_chain AOT pipeline.
//
// Algorithm: /root/reference/apps/stencil_chain/stencil_chain_generator.cpp:18-34 — 32 chained 5x5
// stencils with weights (i+3)*(j+3) in wrapping uint16 arithmetic; only stage 0 is edge-clamped, so stage
// s is needed on the output grown by 2*(32-s).  Boundary: `int stencil_chain(halide_buffer_t *input,
// halide_buffer_t *output)` (:9-10, :150).
//
// Design.  mod-2^16 arithmetic is a commutative ring, so (a) the weight matrix (i+3)(j+3) factors into a vertical
// [1 2 3 4 5] pass and a horizontal [1 2 3 4 5] pass, in either order, with bit-identical results, and (b) v_pk_mad_u16
// on packed pixel pairs IS the reference's wrapping uint16 arithmetic.  FUSE = 8 stages are computed per launch inside
// LDS (temporal tiling, like the reference's CPU schedule fuses groups of stages per tile, :115-143): a workgroup loads a
// 128 x 96 u16 window, ping-pongs it through 8 stages (the valid box shrinks by 2 per stage) and writes the central
// 96 x 64.  Intermediates between launches live on the grown domain in a scratch arena.  HBM traffic: 4 launches x
// (2 B read + 2 B written)/px (+halo); the kernel is bound by instruction issue (10 packed multiply-adds per pixel and
// stage), priced in bench_apps.py against the packed-16-bit VALU rate.
#include "hlmi_internal.h"

using namespace hlmi;

namespace {

constexpr int STENCILS = 32;  // GeneratorParam stencils (:7)
constexpr int FUSE = 8;
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

// ---- lane layout of the fused kernel: a WAVE owns whole rows of the 128-column window, lane l holds the column pair
// (2l, 2l+1), so the horizontal neighbours are the adjacent lanes and come by DPP wave shifts; all 64 lanes are always active:
// what the edge lanes compute from their missing neighbours is garbage that moves inwards 2 pixels per stage — exactly the
// shrinking valid box — and is never stored.  (Round 2's stencil_fused8w kept the window in LDS — one dependent ds_read and
// ds_write per row and stage, a barrier per stage: 0.117 ms at 1536 x 2560 — and was retired in round 4 together with its
// switch HLMI_SC_LDS, as was the eight-thin-waves variant HLMI_SC_THIN; the register-window kernel below runs 0.086 ms.)
constexpr int WTW = 96, WRW = WTW + 4 * FUSE;   // output tile width; window width = 128
static_assert(WRW == 128, "one wave = one window row: 64 lanes x 2 columns");
__device__ __forceinline__ uint32_t sc_lane_prev(uint32_t v) {   // lane-1's value (0 for lane 0): DPP wave_shr:1
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, true);
}
__device__ __forceinline__ uint32_t sc_lane_next(uint32_t v) {   // lane+1's value (0 for lane 63): DPP wave_shl:1
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xf, 0xf, true);
}
// PIN / POUT: the window / the tile moves as aligned dwords (even row strides and origins: always true between launches
// when the width is even)
// ---- stencil_fused8r: 8 fused stages with the window in REGISTERS (the LDS-window kernel spent more time waiting for LDS
// round trips than computing: 0.117 ms where the executed multiply-adds need 0.035 ms).  A wave
// keeps its RW rows of the window — lane l = column pair (2l, 2l+1), one dword per row — in RW registers for all 8 stages
// and updates them in place, top to bottom, with the two previous ORIGINAL rows carried in temporaries; the four waves of a
// workgroup are stacked vertically (window = 128 columns x 4 RW rows) and only exchange their two top and two bottom rows
// through LDS once per stage (8 LDS instructions per wave and stage instead of 2 per row; double-buffered by stage parity:
// one barrier per stage).  The window comes from global memory straight into the registers and the central
// 96 x (4 RW - 32) block goes straight back: no staging pass.  All lanes and rows are always computed;
// what the edge lanes / edge rows make of their missing neighbours is garbage that moves inwards 2 pixels per stage — the
// shrinking valid box — and is never stored.  RW = 32: 96 x 96 outputs per workgroup (1.78x halo recomputation instead of the
// LDS version's 2.0x), 476 workgroups at 1536 x 2560.
// NW waves of RW rows: 4 x 32 and 8 x 16 make the same 128-row window; eight thinner waves double the waves per SIMD (the
// workgroup count is fixed by the tile, 476 at 1536 x 2560: 1.9 waves per SIMD with four) at twice the exchanges.
template<bool CLAMP, bool PIN, bool POUT, int RW, int NW = 4>
__global__ __launch_bounds__(64 * NW) void stencil_fused8r(const uint16_t *__restrict__ src, long src_sy, int sx0, int sy0, int sw,
                                                       int sh, uint16_t *__restrict__ dst, long dst_sy, int dx0, int dy0,
                                                       int dw, int dh) {
    constexpr int WR = NW * RW, OR = WR - 4 * FUSE;                      // window rows, output rows
    __shared__ uint32_t xch[2][NW][4][64];                                // [stage parity][wave][top0, top1, bot0, bot1][lane]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ox = dx0 + blockIdx.x * WTW, oy = dy0 + blockIdx.y * OR;   // absolute coords of the output tile
    const int gx = ox - 2 * FUSE, gy = oy - 2 * FUSE + wave * RW;        // absolute coords of the wave's first row / column
    auto U = [](uint32_t v) { return __builtin_bit_cast(u16x2, v); };
    uint32_t R[RW];
    // ---- load: rows of the wave, column pair of the lane
    {
        const int x = gx + 2 * lane - sx0;                                // column of the pair's first element
        // CLAMP (stage 0, the caller's input): dwords only for windows that lie inside the input (no clamp binds)
        const bool inside = gx >= sx0 && gx + WRW <= sx0 + sw && gy >= sy0 && gy + RW <= sy0 + sh;   // wave-uniform
        if (PIN && inside) {               // the common case: every row and column of the wave's window exists
            const uint16_t *p = src + (long)(gy - sy0) * src_sy + x;
#pragma unroll
            for (int r = 0; r < RW; r++) R[r] = *reinterpret_cast<const uint32_t *>(p + (long)r * src_sy);
        } else if (PIN && !CLAMP) {
#pragma unroll
            for (int r = 0; r < RW; r++) {
                const int y = gy + r - sy0;
                const bool ok = x >= 0 && x + 1 < sw && y >= 0 && y < sh;
                R[r] = ok ? *reinterpret_cast<const uint32_t *>(src + (long)y * src_sy + x) : 0u;
            }
        } else {
#pragma unroll
            for (int r = 0; r < RW; r++) {
                int y = gy + r - sy0, x0 = x, x1 = x + 1;
                uint32_t a, b;
                if (CLAMP) {
                    y = min(max(y, 0), sh - 1), x0 = min(max(x0, 0), sw - 1), x1 = min(max(x1, 0), sw - 1);
                    a = src[(long)y * src_sy + x0], b = src[(long)y * src_sy + x1];
                } else {
                    // windows of edge tiles may poke outside the producer's domain; those cells only ever feed outputs
                    // outside the destination domain, which are not stored
                    const bool oky = y >= 0 && y < sh;
                    a = (oky && x0 >= 0 && x0 < sw) ? src[(long)y * src_sy + x0] : (uint16_t)0;
                    b = (oky && x1 >= 0 && x1 < sw) ? src[(long)y * src_sy + x1] : (uint16_t)0;
                }
                R[r] = a | (b << 16);
            }
        }
    }
    // the weights live in scalar registers the compiler cannot see through: with literal 2 and 4 it strength-reduces the
    // products to v_pk_lshlrev_b16 + v_pk_add_u16 (12 packed ops per row instead of 8 v_pk_mad_u16)
    uint32_t w2 = 0x00020002u, w3 = 0x00030003u, w4 = 0x00040004u, w5 = 0x00050005u;
    asm volatile("" : "+s"(w2), "+s"(w3), "+s"(w4), "+s"(w5));
    const u16x2 k2 = U(w2), k3 = U(w3), k4 = U(w4), k5 = U(w5);
#pragma unroll 1
    for (int m = 0; m < FUSE; m++) {
        uint32_t (*x)[4][64] = xch[m & 1];
        x[wave][0][lane] = R[0], x[wave][1][lane] = R[1], x[wave][2][lane] = R[RW - 2], x[wave][3][lane] = R[RW - 1];
        __syncthreads();
        // rows -2, -1 (the wave above's bottom rows) and RW, RW + 1 (the wave below's top rows); the first / last wave of the
        // window has no such neighbour: zeros, i.e. garbage in rows that are outside the valid box anyway
        const int wu = max(wave - 1, 0), wd = min(wave + 1, NW - 1);
        uint32_t p2 = x[wu][2][lane], p1 = x[wu][3][lane], b0 = x[wd][0][lane], b1 = x[wd][1][lane];
        if (wave == 0) p2 = 0u, p1 = 0u;
        if (wave == NW - 1) b0 = 0u, b1 = 0u;
#pragma unroll
        for (int r = 0; r < RW; r++) {
            const uint32_t cur = R[r];
            const uint32_t n1 = r + 1 < RW ? R[r + 1] : b0, n2 = r + 2 < RW ? R[r + 2] : (r + 2 == RW ? b0 : b1);
            const u16x2 v = U(p2) + k2 * U(p1) + k3 * U(cur) + k4 * U(n1) + k5 * U(n2);   // pair (x, x+1), vertical sums
            const uint32_t vb = __builtin_bit_cast(uint32_t, v);
            const uint32_t va = sc_lane_prev(vb), vc = sc_lane_next(vb);                   // (x-2, x-1), (x+2, x+3)
            const u16x2 S1 = U(__builtin_amdgcn_alignbit(vb, va, 16)), S2 = U(__builtin_amdgcn_alignbit(vc, vb, 16));
            R[r] = __builtin_bit_cast(uint32_t, (u16x2)(U(va) + k2 * S1 + k3 * v + k4 * S2 + k5 * U(vc)));
            p2 = p1, p1 = cur;
        }
    }
    // ---- store: window rows [2 FUSE, WR - 2 FUSE), pairs [FUSE, 64 - FUSE)
    const bool whole = ox + WTW <= dx0 + dw && oy + OR <= dy0 + dh;      // workgroup-uniform: the tile lies inside the destination
    if (POUT && whole) {
        if (lane >= FUSE && lane < 64 - FUSE) {
            uint16_t *p = dst + (long)(oy + wave * RW - 2 * FUSE - dy0) * dst_sy + (ox + 2 * (lane - FUSE) - dx0);
#pragma unroll
            for (int r = 0; r < RW; r++) {
                const int wr = wave * RW + r;                           // window row (wave-uniform condition)
                if (wr >= 2 * FUSE && wr < WR - 2 * FUSE) *reinterpret_cast<uint32_t *>(p + (long)r * dst_sy) = R[r];
            }
        }
    } else if (lane >= FUSE && lane < 64 - FUSE) {
        const int X = ox + 2 * (lane - FUSE) - dx0;
#pragma unroll
        for (int r = 0; r < RW; r++) {
            const int wr = wave * RW + r;                               // window row
            const int Y = oy + wr - 2 * FUSE - dy0;
            if (wr >= 2 * FUSE && wr < WR - 2 * FUSE && Y < dh) {
                if (POUT && X + 1 < dw) {
                    *reinterpret_cast<uint32_t *>(dst + (long)Y * dst_sy + X) = R[r];
                } else {
                    if (X < dw) dst[(long)Y * dst_sy + X] = (uint16_t)(R[r] & 0xffffu);
                    if (X + 1 < dw) dst[(long)Y * dst_sy + X + 1] = (uint16_t)(R[r] >> 16);
                }
            }
        }
    }
}

const int64_t e0 = 0, ew = 1536, eh = 2560;
const int64_t *const est[4] = {&e0, &ew, &e0, &eh};
const halide_type_t ty_u16 = {(decltype(halide_type_t::code))1, 16, 0};
const halide_filter_argument_t sc_args[2] = {
    {"input", halide_argument_kind_input_buffer, 2, ty_u16, nullptr, nullptr, nullptr, nullptr, est},
    {"output", halide_argument_kind_output_buffer, 2, ty_u16, nullptr, nullptr, nullptr, nullptr, est},
};
const halide_filter_metadata_t sc_md = {1, 2, sc_args, kTargetString, "stencil_chain"};

}  // namespace

extern "C" int stencil_chain(halide_buffer_t *input, halide_buffer_t *output) {
    void *uc = nullptr;
    BufArg args[2] = {{"input", input, T_U16, 2, false}, {"output", output, T_U16, 2, true}};
    int r = check_not_null(uc, args, 2);
    if (r) return r;
    if ((r = check_type_and_dims(uc, args, 2))) return r;
    if (any_bounds_query(args, 2)) {
        // every tap of the input goes through repeat_edge (:18): any non-empty input is acceptable; propose
        // the output's region (what tools/RunGen.h:1212-1250 then allocates)
        int mins[2] = {output->dim[0].min, output->dim[1].min}, ext[2] = {output->dim[0].extent, output->dim[1].extent};
        answer_query(input, mins, ext);
        answer_query(output, mins, ext);
        return 0;
    }
    if ((r = check_shape(uc, args[0])) || (r = check_shape(uc, args[1]))) return r;
    const int W = output->dim[0].extent, H = output->dim[1].extent;
    if (W > 0 && H > 0 && (input->dim[0].extent < 1 || input->dim[1].extent < 1)) {
        return report(uc, halide_error_code_access_out_of_bounds, "Input buffer input is empty but is accessed (clamped) at 0");
    }
    DeviceCtx ctx;
    if ((r = acquire_device(uc, &ctx))) return r;
    if ((r = input_to_device(uc, ctx, args[0]))) return r;
    if ((r = output_on_device(uc, ctx, args[1]))) return r;
    if (W == 0 || H == 0) {
        mark_output_written(output);
        return 0;
    }
    constexpr int NL = STENCILS / FUSE;  // launches
    const int ox0 = output->dim[0].min, oy0 = output->dim[1].min;
    // domain after launch L (0-based): output grown by g_L = 2*FUSE*(NL-1-L)
    auto grow = [&](int L) { return 2 * FUSE * (NL - 1 - L); };
    size_t plane = (size_t)(W + 2 * grow(0)) * (H + 2 * grow(0));
    void *ws = nullptr;
    if ((r = get_workspace(uc, ctx, 2 * ((plane * 2 + 255) & ~(size_t)255), &ws))) return r;
    uint16_t *tmp[2] = {(uint16_t *)ws, (uint16_t *)((char *)ws + ((plane * 2 + 255) & ~(size_t)255))};

    const uint16_t *src = dev_ptr<uint16_t>(input);
    long src_sy = input->dim[1].stride;
    int sx0 = input->dim[0].min, sy0 = input->dim[1].min, sw = input->dim[0].extent, sh = input->dim[1].extent;
    for (int L = 0; L < NL; L++) {
        const int g = grow(L);
        const int dx0 = ox0 - g, dy0 = oy0 - g, dw = W + 2 * g, dh = H + 2 * g;
        uint16_t *dst = (L == NL - 1) ? dev_ptr<uint16_t>(output) : tmp[L & 1];
        long dst_sy = (L == NL - 1) ? (long)output->dim[1].stride : (long)dw;
        constexpr int RW = 32, ORW = 4 * RW - 4 * FUSE;   // stencil_fused8r: rows per wave, output rows per workgroup
        dim3 gridr((dw + WTW - 1) / WTW, (dh + ORW - 1) / ORW);
        // dword moves: the intermediates are dense planes on even origins (their offsets from the tile grid are
        // multiples of 2 FUSE), so an even width makes every pair an aligned dword; the user's output needs checking
        const bool pin = L > 0 ? (sw % 2 == 0 && src_sy % 2 == 0)
                               : (src_sy % 2 == 0 && (uintptr_t)src % 4 == 0 && ((dx0 - 2 * FUSE - sx0) & 1) == 0);
        const bool pout = dst_sy % 2 == 0 && (uintptr_t)dst % 4 == 0;
#define SC_R(C, I, O)                                                                                                          \
HLMI_LAUNCH(uc, "stencil_fused8", ctx.stream, (stencil_fused8r<C, I, O, RW>), gridr, dim3(256), 0, src, src_sy, sx0, sy0, sw, sh, dst, \
            dst_sy, dx0, dy0, dw, dh)
        if (L == 0) {
            if (pin) { if (pout) SC_R(true, true, true); else SC_R(true, true, false); }
            else { if (pout) SC_R(true, false, true); else SC_R(true, false, false); }
        }
        else if (pin) { if (pout) SC_R(false, true, true); else SC_R(false, true, false); }
        else { if (pout) SC_R(false, false, true); else SC_R(false, false, false); }
#undef SC_R
        src = dst, src_sy = dst_sy, sx0 = dx0, sy0 = dy0, sw = dw, sh = dh;
    }
    mark_output_written(output);
    return 0;
}

extern "C" int stencil_chain_argv(void **a) { return stencil_chain((halide_buffer_t *)a[0], (halide_buffer_t *)a[1]); }
extern "C" const halide_filter_metadata_t *stencil_chain_metadata(void) { return &sc_md; }
extern "C" int stencil_chain_auto_schedule(halide_buffer_t *input, halide_buffer_t *output) { return stencil_chain(input, output); }
