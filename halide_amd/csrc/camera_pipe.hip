// camera_pipe.hip — gfx950 implementation of the reference's camera_pipe AOT pipeline (raw Bayer -> RGB8).
//
// Algorithm: /root/reference/apps/camera_pipe/camera_pipe_generator.cpp — shift (16,12) :406-413, hot-pixel
// suppression :240-250, deinterleave :252-263, Demosaic :47-152, colour matrix (Q8.8) :265-299, tone curve LUT
// :301-366, 1-2-1 unsharp mask :368-404.  Boundary: `int camera_pipe(input, matrix_3200, matrix_7000, color_temp,
// gamma, contrast, sharpen_strength, blackLevel, whiteLevel, processed)` (:219-228, :622).
// All per-pixel stages are integer (u16/i16/i32/u8 with the reference's wrap / floor-division rules) and therefore
// exact; only the 12 matrix coefficients and the 1024-entry curve are float set-up work (cp_setup kernel).
// HBM: 2 B read + 3 B written per output pixel; v1 materialises the curved u8 planes once (3+3 B/px extra).
//   cp_setup    1 block: colour matrix, tone curve, sharpen strength
//   cp_demosaic one thread per Bayer quad: 10x10 raw window -> hot-pixel clamp -> 3x3x4 deinterleaved values in
//               registers -> demosaic 4 pixels -> matrix -> curve -> u8 planes on [-1, W] x [-1, H]
//   cp_sharpen  one thread per output pixel, 3 channels
#include <mutex>
#include <string.h>

#include "hlmi_device_math.h"
#include "hlmi_internal.h"

#include <stdio.h>

#include <stdlib.h>

using namespace hlmi;

namespace {

struct CPSetup {          // lives in the scratch arena, written by cp_setup
    int16_t matrix[12];   // [row c][col j] = matrix(j, c)
    uint8_t strength_x32;
    uint8_t pad[7];
    uint8_t curve[1024];
};

__global__ void cp_setup(const float *__restrict__ m3200, long m3_sy, const float *__restrict__ m7000, long m7_sy,
                         float color_temp, float gamma, float contrast, float sharpen_strength, int blackLevel, int whiteLevel,
                         CPSetup *__restrict__ s) {
    const int t = threadIdx.x;
    if (t < 12) {
        const float k1 = 1.0f / 3200, k2 = 1.0f / 7000;
        const float inv_den = 1.0f / (k2 - k1);
        float alpha = (1.0f / color_temp - k1) * inv_den;
        int c = t / 4, j = t - 4 * c;
        float val = dev::mad2(m3200[c * m3_sy + j], alpha, m7000[c * m7_sy + j], 1.0f - alpha);
        s->matrix[t] = (int16_t)(val * 256.0f);
    }
    if (t == 12) s->strength_x32 = (uint8_t)dev::clampf(sharpen_strength * 32.0f, 0.0f, 255.0f);
    if (t < 1024) {
        const int minRaw = 0 + blackLevel, maxRaw = whiteLevel;
        float invRange = 1.0f / (float)(maxRaw - minRaw);
        float b = 2.0f - dev::halide_pow(2.0f, contrast * (1.0f / 100.0f));
        float a = 2.0f - 2.0f * b;
        float xf = dev::clampf((float)(t - minRaw) * invRange, 0.0f, 1.0f);
        float g = dev::halide_pow(xf, 1.0f / gamma);
        float z = g > 0.5f ? 1.0f - dev::mad2(a * (1.0f - g), 1.0f - g, b, 1.0f - g) : dev::mad2(a * g, g, b, g);
        uint8_t val = (uint8_t)dev::clampf(dev::mad(z, 255.0f, 0.5f), 0.0f, 255.0f);
        s->curve[t] = t <= minRaw ? (uint8_t)0 : (t > maxRaw ? (uint8_t)255 : val);
    }
}

__device__ __forceinline__ uint16_t avg16(uint16_t a, uint16_t b) { return (uint16_t)(((uint32_t)a + b + 1u) >> 1); }
__device__ __forceinline__ uint8_t avg8(uint8_t a, uint8_t b) { return (uint8_t)(((uint32_t)a + b + 1u) >> 1); }
__device__ __forceinline__ uint16_t absd16(uint16_t a, uint16_t b) { return a > b ? (uint16_t)(a - b) : (uint16_t)(b - a); }
__device__ __forceinline__ uint16_t u16(uint32_t v) { return (uint16_t)v; }

// raw: pointer to shifted(0,0) = input(16,12) relative to the buffer's own min; cv: u8 planes [3][CH][CW] of curved on
// [-1, W] x [-1, H]
template<bool PAIRS>
__global__ __launch_bounds__(256) void cp_demosaic(const uint16_t *__restrict__ raw, long in_sy, const CPSetup *__restrict__ s,
                                                  uint8_t *__restrict__ cv, int CW, int CH, int CWL, int nqx, int nqy) {
    const int qi = blockIdx.x * blockDim.x + threadIdx.x, qj = blockIdx.y;
    if (qi >= nqx) return;
    const int qx = qi - 1, qy = qj - 1;  // quads start at fdiv(-1, 2) = -1
    // 10x10 raw window, origin (2qx-4, 2qy-4)
    uint16_t R[10][10];
    const uint16_t *base = raw + (long)(2 * qy - 4) * in_sy + (2 * qx - 4);
    if (PAIRS) {
        // the window starts at an even column: with an even row stride and a 4-byte aligned image every (2k, 2k+1)
        // column pair is one aligned dword — 44 loads instead of 84
#pragma unroll
        for (int j = 0; j < 10; j++)
#pragma unroll
            for (int i = 0; i < 10; i += 2) {
                const bool used = !((i < 2 || i > 7) && (j < 2 || j > 7));
                const uint32_t w = used ? *reinterpret_cast<const uint32_t *>(base + (long)j * in_sy + i) : 0u;
                R[j][i] = (uint16_t)(w & 0xffffu), R[j][i + 1] = (uint16_t)(w >> 16);
            }
    } else {
#pragma unroll
        for (int j = 0; j < 10; j++)
#pragma unroll
            for (int i = 0; i < 10; i++) {
                // only the plus-shaped footprint of the 6x6 centre is ever used; skip the 2x2 corners
                bool used = !((i < 2 || i > 7) && (j < 2 || j > 7));
                R[j][i] = used ? base[(long)j * in_sy + i] : (uint16_t)0;
            }
    }
    // hot-pixel suppression (:240-250) + deinterleave (:252-263): D[c][dy+1][dx+1]
    uint16_t D[4][3][3];
#pragma unroll
    for (int c = 0; c < 4; c++)
#pragma unroll
        for (int dy = 0; dy < 3; dy++)
#pragma unroll
            for (int dx = 0; dx < 3; dx++) {
                const int i = 2 * (dx - 1) + (c & 1) + 4, j = 2 * (dy - 1) + (c >> 1) + 4;
                uint16_t a = max(max(R[j][i - 2], R[j][i + 2]), max(R[j - 2][i], R[j + 2][i]));
                D[c][dy][dx] = min(R[j][i], a);
            }
#define G_GR(dx, dy) D[0][(dy) + 1][(dx) + 1]
#define R_R(dx, dy) D[1][(dy) + 1][(dx) + 1]
#define B_B(dx, dy) D[2][(dy) + 1][(dx) + 1]
#define G_GB(dx, dy) D[3][(dy) + 1][(dx) + 1]
    auto g_r = [&](int dx, int dy) -> uint16_t {
        uint16_t gv = avg16(G_GB(dx, dy - 1), G_GB(dx, dy)), gvd = absd16(G_GB(dx, dy - 1), G_GB(dx, dy));
        uint16_t gh = avg16(G_GR(dx + 1, dy), G_GR(dx, dy)), ghd = absd16(G_GR(dx + 1, dy), G_GR(dx, dy));
        return ghd < gvd ? gh : gv;
    };
    auto g_b = [&](int dx, int dy) -> uint16_t {
        uint16_t gv = avg16(G_GR(dx, dy + 1), G_GR(dx, dy)), gvd = absd16(G_GR(dx, dy + 1), G_GR(dx, dy));
        uint16_t gh = avg16(G_GB(dx - 1, dy), G_GB(dx, dy)), ghd = absd16(G_GB(dx - 1, dy), G_GB(dx, dy));
        return ghd < gvd ? gh : gv;
    };
    const uint16_t gr00 = g_r(0, 0), grm0 = g_r(-1, 0), gr01 = g_r(0, 1), grm1 = g_r(-1, 1);
    const uint16_t gb00 = g_b(0, 0), gb0m = g_b(0, -1), gb10 = g_b(1, 0), gb1m = g_b(1, -1);

    uint16_t px[4][3];  // [site: gr, r, b, gb][r, g, b]
    // green-red site (even, even)
    px[0][1] = G_GR(0, 0);
    px[0][0] = u16(u16(G_GR(0, 0) - avg16(gr00, grm0)) + avg16(R_R(-1, 0), R_R(0, 0)));
    px[0][2] = u16(u16(G_GR(0, 0) - avg16(gb00, gb0m)) + avg16(B_B(0, 0), B_B(0, -1)));
    // red site (odd, even)
    px[1][0] = R_R(0, 0);
    px[1][1] = gr00;
    {
        uint16_t bp = u16(u16(gr00 - avg16(gb00, gb1m)) + avg16(B_B(0, 0), B_B(1, -1)));
        uint16_t bpd = absd16(B_B(0, 0), B_B(1, -1));
        uint16_t bn = u16(u16(gr00 - avg16(gb10, gb0m)) + avg16(B_B(1, 0), B_B(0, -1)));
        uint16_t bnd = absd16(B_B(1, 0), B_B(0, -1));
        px[1][2] = bpd < bnd ? bp : bn;
    }
    // blue site (even, odd)
    px[2][2] = B_B(0, 0);
    px[2][1] = gb00;
    {
        uint16_t rp = u16(u16(gb00 - avg16(gr00, grm1)) + avg16(R_R(0, 0), R_R(-1, 1)));
        uint16_t rpd = absd16(R_R(0, 0), R_R(-1, 1));
        uint16_t rn = u16(u16(gb00 - avg16(grm0, gr01)) + avg16(R_R(-1, 0), R_R(0, 1)));
        uint16_t rnd = absd16(R_R(-1, 0), R_R(0, 1));
        px[2][0] = rpd < rnd ? rp : rn;
    }
    // green-blue site (odd, odd)
    px[3][1] = G_GB(0, 0);
    px[3][0] = u16(u16(G_GB(0, 0) - avg16(gr00, gr01)) + avg16(R_R(0, 0), R_R(0, 1)));
    px[3][2] = u16(u16(G_GB(0, 0) - avg16(gb00, gb10)) + avg16(B_B(0, 0), B_B(1, 0)));
#undef G_GR
#undef R_R
#undef B_B
#undef G_GB
    // colour matrix (Q8.8, floor /256) + tone curve -> u8, store to the curved planes
    const size_t plane = (size_t)CW * CH;
#pragma unroll
    for (int site = 0; site < 4; site++) {
        const int X = 2 * qx + (site & 1), Y = 2 * qy + (site >> 1);
        const int cxx = X + 1, cyy = Y + 1;
        if (cxx < 0 || cxx >= CWL || cyy < 0 || cyy >= CH) continue;   // CWL = logical width W+2, CW = row pitch
        const int32_t ir = (int16_t)px[site][0], ig = (int16_t)px[site][1], ib = (int16_t)px[site][2];
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const int16_t *m = &s->matrix[4 * c];
            int32_t v = (((int32_t)m[3] + (int32_t)m[0] * ir) + (int32_t)m[1] * ig) + (int32_t)m[2] * ib;
            int16_t cc = (int16_t)(v >> 8);
            cv[(size_t)c * plane + (size_t)cyy * CW + cxx] = s->curve[dev::clampi(cc, 0, 1023)];
        }
    }
}

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
constexpr int TQX = 32;                               // quad columns of output per tile (64 pixels)
// ---- cp_fused_tile: every stage in ONE launch, a tile of 64 x 40 OUTPUT pixels per workgroup, staged through LDS so that every
// raw pixel is loaded once and every hot-pixel clamp is evaluated once (cp_demosaic loads a 10 x 10 window and clamps 36 pixels
// per quad: 9x redundant):
//   raw window 76 x 52 (aligned dword pairs) -> LDS;  clamped pairs (sites {Gr,R} and {B,Gb} of a quad are adjacent pixels:
//   v_pk_max_u16 / v_pk_min_u16) for 36 x 24 quads -> LDS;  a thread per quad, 34 x 22 quads in three passes: 18 LDS dwords ->
//   demosaic -> matrix -> curve (LUT in LDS) -> staged u8 rows of the curved planes (66 x 42 used: the tile and one pixel
//   around it);  sharpen from the staged rows, four pixels per thread as cp_sharpen4 does -> the output.
// Round 4: the curved planes used to go out to a 14.7 MB workspace from a 32 x 8-quad tile kernel and come back in a sharpen
// launch (0.036 ms per call); making the one-pixel ring of curved values again in the neighbour tiles costs 1.13 x the quad
// arithmetic and saves a launch and both passes over that workspace: 0.027 ms.  Per-workgroup time stamps (1920 workgroups, one
// round of eight per CU): raw window 2.4 us, clamp 4.1 (with the wait for the window), quads 3.4, sharpen + stores 1.6,
// 14 us in all at the median and 25 for the launch — the youngest workgroups of a CU issue last; reading ONE window from every
// tile changes nothing, i.e. it is not the raw reads.  Tiles of 64 x 16 / 64 x 24 pixels (2.3 / 1.6 rounds): 0.028.
// Needs the PAIRS alignment (even row stride, 4-byte aligned origin) and an even W.  Raw pixels outside the footprint the
// boundary guarantees ([-6, W+5] x [-6, H+5] around the output) are read as 0; they only feed pixels that are not stored.
constexpr int FTY = 20;                                                      // quad rows of OUTPUT per tile: 64 x 40 pixels
constexpr int FQX = TQX + 2, FQY = FTY + 2;                                  // quads made per tile: 34 x 22 = 748 on 256 threads, three passes
constexpr int FRWD = (2 * FQX + 8) / 2, FRH = 2 * FQY + 8, FRPD = FRWD + 1;  // raw window: 38 dwords (76 pixels) x 52 rows, pitch 39
constexpr int FDQX = FQX + 2, FDQY = FQY + 2, FDPD = FDQX + 1;               // clamped quads 36 x 24, pitch 37
constexpr int FOP = 72;                                                      // staged row pitch in bytes (column k at byte k + 3)
typedef short i16x2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void cp_fused_tile(const uint16_t *__restrict__ raw, long in_sy, const CPSetup *__restrict__ s,
                                                    uint8_t *__restrict__ out, long out_sy, long out_sc, int W, int H, int dwords, int gx) {
    // the raw window is dead once the clamped quads exist, the staged output rows are first written after that: one array
    constexpr int RAW_DW = FRH * FRPD, OUT_DW = 3 * 2 * FQY * FOP / 4;
    __shared__ uint32_t s_buf[RAW_DW > OUT_DW ? RAW_DW : OUT_DW];
    __shared__ uint32_t s_d[2 * FDQY * FDPD];           // [cy][j][i]: sites (2 cy, 2 cy + 1) of clamped quad (i, j)
    __shared__ uint32_t s_curve[256];
    uint32_t *s_raw = s_buf;
    uint8_t *s_out = reinterpret_cast<uint8_t *>(s_buf);
    const int tid = threadIdx.x;
    // tiles in row-major order, a contiguous run of them per XCD (blocks are dealt round-robin over the 8 XCDs): a tile's raw
    // window shares its outer cache lines and its halo rows with its neighbours' — they should meet in one L2
    const int nb8 = gridDim.x >> 3, lb = (int)blockIdx.x < (nb8 << 3) ? ((int)blockIdx.x & 7) * nb8 + ((int)blockIdx.x >> 3) : (int)blockIdx.x;
    const int tbx = lb % gx, tby = lb / gx;
    const int QX0 = -1 + TQX * tbx, QY0 = -1 + FTY * tby;   // first quad made: quads start at fdiv(-1, 2) = -1
    s_curve[tid] = reinterpret_cast<const uint32_t *>(s->curve)[tid];
    const int rx0 = 2 * QX0 - 4, ry0 = 2 * QY0 - 4;   // raw window origin (even)
    {
        // the thread's window dwords are requested together (a loop waited for each before asking for the next); sites outside
        // the raw image read its first dword and become 0
        constexpr int N1 = (FRH * FRWD + 255) / 256;
        uint32_t v[N1];
        bool ok[N1];
#pragma unroll
        for (int k = 0; k < N1; k++) {
            const int i = min(tid + 256 * k, FRH * FRWD - 1), r = i / FRWD, cdw = i - r * FRWD;
            const int x = rx0 + 2 * cdw, y = ry0 + r;
            ok[k] = x >= -6 && x + 1 <= W + 5 && y >= -6 && y <= H + 5;
            v[k] = *reinterpret_cast<const uint32_t *>(raw + (ok[k] ? (long)y * in_sy + x : 0l));
        }
#pragma unroll
        for (int k = 0; k < N1; k++) {
            const int i = tid + 256 * k;
            if (i < FRH * FRWD) s_raw[(i / FRWD) * FRPD + (i - (i / FRWD) * FRWD)] = ok[k] ? v[k] : 0u;
        }
    }
    __syncthreads();
    // hot-pixel suppression (:240-250) on pixel pairs: clamped quad (i, j), pair cy -> raw window dword (i + 1, 2 j + cy + 2)
    for (int it = tid; it < 2 * FDQY * FDQX; it += 256) {
        const int cy = it / (FDQY * FDQX), rem = it - cy * (FDQY * FDQX), j = rem / FDQX, i = rem - j * FDQX;
        const uint32_t *p = s_raw + (2 * j + cy + 2) * FRPD + (i + 1);
        const u16x2 c0 = __builtin_bit_cast(u16x2, p[0]), l = __builtin_bit_cast(u16x2, p[-1]), r = __builtin_bit_cast(u16x2, p[1]),
                    u = __builtin_bit_cast(u16x2, p[-2 * FRPD]), d = __builtin_bit_cast(u16x2, p[2 * FRPD]);
        const u16x2 a = __builtin_elementwise_max(__builtin_elementwise_max(l, r), __builtin_elementwise_max(u, d));
        s_d[(cy * FDQY + j) * FDPD + i] = __builtin_bit_cast(uint32_t, __builtin_elementwise_min(c0, a));
    }
    __syncthreads();
    const uint8_t *curve = reinterpret_cast<const uint8_t *>(s_curve);
#pragma unroll 1
    for (int qi = tid; qi < FQX * FQY; qi += 256) {
    const int ty = qi / FQX, tx = qi - ty * FQX;
    // deinterleave (:252-263): D[c][dy+1][dx+1]
    uint16_t D[4][3][3];
#pragma unroll
    for (int cy = 0; cy < 2; cy++)
#pragma unroll
        for (int dy = 0; dy < 3; dy++)
#pragma unroll
            for (int dx = 0; dx < 3; dx++) {
                const uint32_t w = s_d[(cy * FDQY + ty + dy) * FDPD + tx + dx];
                D[2 * cy][dy][dx] = (uint16_t)(w & 0xffffu), D[2 * cy + 1][dy][dx] = (uint16_t)(w >> 16);
            }
#define G_GR(dx, dy) D[0][(dy) + 1][(dx) + 1]
#define R_R(dx, dy) D[1][(dy) + 1][(dx) + 1]
#define B_B(dx, dy) D[2][(dy) + 1][(dx) + 1]
#define G_GB(dx, dy) D[3][(dy) + 1][(dx) + 1]
    auto g_r = [&](int dx, int dy) -> uint16_t {
        uint16_t gv = avg16(G_GB(dx, dy - 1), G_GB(dx, dy)), gvd = absd16(G_GB(dx, dy - 1), G_GB(dx, dy));
        uint16_t gh = avg16(G_GR(dx + 1, dy), G_GR(dx, dy)), ghd = absd16(G_GR(dx + 1, dy), G_GR(dx, dy));
        return ghd < gvd ? gh : gv;
    };
    auto g_b = [&](int dx, int dy) -> uint16_t {
        uint16_t gv = avg16(G_GR(dx, dy + 1), G_GR(dx, dy)), gvd = absd16(G_GR(dx, dy + 1), G_GR(dx, dy));
        uint16_t gh = avg16(G_GB(dx - 1, dy), G_GB(dx, dy)), ghd = absd16(G_GB(dx - 1, dy), G_GB(dx, dy));
        return ghd < gvd ? gh : gv;
    };
    const uint16_t gr00 = g_r(0, 0), grm0 = g_r(-1, 0), gr01 = g_r(0, 1), grm1 = g_r(-1, 1);
    const uint16_t gb00 = g_b(0, 0), gb0m = g_b(0, -1), gb10 = g_b(1, 0), gb1m = g_b(1, -1);
    uint16_t px[4][3];  // [site: gr, r, b, gb][r, g, b]
    px[0][1] = G_GR(0, 0);
    px[0][0] = u16(u16(G_GR(0, 0) - avg16(gr00, grm0)) + avg16(R_R(-1, 0), R_R(0, 0)));
    px[0][2] = u16(u16(G_GR(0, 0) - avg16(gb00, gb0m)) + avg16(B_B(0, 0), B_B(0, -1)));
    px[1][0] = R_R(0, 0);
    px[1][1] = gr00;
    {
        uint16_t bp = u16(u16(gr00 - avg16(gb00, gb1m)) + avg16(B_B(0, 0), B_B(1, -1)));
        uint16_t bpd = absd16(B_B(0, 0), B_B(1, -1));
        uint16_t bn = u16(u16(gr00 - avg16(gb10, gb0m)) + avg16(B_B(1, 0), B_B(0, -1)));
        uint16_t bnd = absd16(B_B(1, 0), B_B(0, -1));
        px[1][2] = bpd < bnd ? bp : bn;
    }
    px[2][2] = B_B(0, 0);
    px[2][1] = gb00;
    {
        uint16_t rp = u16(u16(gb00 - avg16(gr00, grm1)) + avg16(R_R(0, 0), R_R(-1, 1)));
        uint16_t rpd = absd16(R_R(0, 0), R_R(-1, 1));
        uint16_t rn = u16(u16(gb00 - avg16(grm0, gr01)) + avg16(R_R(-1, 0), R_R(0, 1)));
        uint16_t rnd = absd16(R_R(-1, 0), R_R(0, 1));
        px[2][0] = rpd < rnd ? rp : rn;
    }
    px[3][1] = G_GB(0, 0);
    px[3][0] = u16(u16(G_GB(0, 0) - avg16(gr00, gr01)) + avg16(R_R(0, 0), R_R(0, 1)));
    px[3][2] = u16(u16(G_GB(0, 0) - avg16(gb00, gb10)) + avg16(B_B(0, 0), B_B(1, 0)));
#undef G_GR
#undef R_R
#undef B_B
#undef G_GB
    // colour matrix (Q8.8, floor /256) + tone curve -> u8, staged: pixel (2 tx + sx, 2 ty + sy) of channel c
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const int16_t *m = &s->matrix[4 * c];
#pragma unroll
        for (int sy = 0; sy < 2; sy++) {
            uint8_t o[2];
#pragma unroll
            for (int sx = 0; sx < 2; sx++) {
                const int site = 2 * sy + sx;
                const int32_t ir = (int16_t)px[site][0], ig = (int16_t)px[site][1], ib = (int16_t)px[site][2];
                const int32_t v = (((int32_t)m[3] + (int32_t)m[0] * ir) + (int32_t)m[1] * ig) + (int32_t)m[2] * ib;
                o[sx] = curve[dev::clampi((int16_t)(v >> 8), 0, 1023)];
            }
            // column 2 tx sits at byte 2 tx + 3: an odd address, two byte stores
            uint8_t *q = s_out + (c * 2 * FQY + 2 * ty + sy) * FOP + 2 * tx + 3;
            q[0] = o[0], q[1] = o[1];
        }
    }
    }
    __syncthreads();
    // sharpen (:398-404) from the staged planes: staged row r = curved row 40 by - 2 + r (2 FTY = 40 output rows per tile, quads
    // start one quad above), the pixel of curved column 64 bx - 2 + k sits at byte k + 3 of its row; output pixel (64 bx + j, 40 by + i)
    // reads staged rows i + 1 .. i + 3 and the bytes j + 4 .. j + 6 (its own value at byte j + 5).  A thread owns
    // four adjacent pixels of one row, as cp_sharpen4 does: two aligned dwords per row, v_lerp_u8 rounding averages, packed i16.
    const short st = (short)s->strength_x32;
    const i16x2 strength = {st, st}, zero = {0, 0}, top = {255, 255};
    const uint32_t one = 0x01010101u;
    auto avg4 = [&](uint32_t a, uint32_t b) { return __builtin_amdgcn_lerp(a, b, one); };
    for (int t = tid; t < 16 * 2 * FTY; t += 256) {
    const int i = t >> 4, mg = t & 15;
    const int x = 64 * tbx + 4 * mg, y = 2 * FTY * tby + i;
    if (x >= W || y >= H) continue;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const uint32_t *r0 = reinterpret_cast<const uint32_t *>(s_out + (c * 2 * FQY + i + 1) * FOP + 4 + 4 * mg);
        const uint32_t *r1 = r0 + FOP / 4, *r2 = r0 + 2 * (FOP / 4);
        const uint32_t a0 = r0[0], b0 = r0[1], a1 = r1[0], b1 = r1[1], a2 = r2[0], b2 = r2[1];
        const uint32_t uya = avg4(avg4(a0, a2), a1), uyb = avg4(avg4(b0, b2), b1);
        const uint32_t L = uya, C = __builtin_amdgcn_alignbyte(uyb, uya, 1), R = __builtin_amdgcn_alignbyte(uyb, uya, 2);
        const uint32_t un = avg4(avg4(L, R), C);
        const uint32_t P = __builtin_amdgcn_alignbyte(b1, a1, 1);
        auto sharpen2 = [&](uint32_t p2, uint32_t u2) -> uint32_t {
            const i16x2 p = __builtin_bit_cast(i16x2, p2), u = __builtin_bit_cast(i16x2, u2);
            const i16x2 mask = p - u;
            const i16x2 q = (i16x2)(mask * strength) >> 5;   // int16 product wraps (src/IROperator.cpp:769-816); floor /32
            i16x2 v = p + q;
            v = v < zero ? zero : v;
            v = v > top ? top : v;
            return __builtin_bit_cast(uint32_t, v);
        };
        const uint32_t even = sharpen2(P & 0x00ff00ffu, un & 0x00ff00ffu), odd = sharpen2((P >> 8) & 0x00ff00ffu, (un >> 8) & 0x00ff00ffu);
        const uint32_t v4 = even | (odd << 8);
        uint8_t *o = out + (long)y * out_sy + x + (long)c * out_sc;
        if (dwords && x + 3 < W) {
            *reinterpret_cast<uint32_t *>(o) = v4;
        } else {
#pragma unroll
            for (int b = 0; b < 4; b++)
                if (x + b < W) o[b] = (uint8_t)(v4 >> (8 * b));
        }
    }
    }
}

__global__ __launch_bounds__(256) void cp_sharpen(const uint8_t *__restrict__ cv, int CW, int CH, const CPSetup *__restrict__ s,
                                                 uint8_t *__restrict__ out, long out_sy, long out_sc, int W, int H) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    const int16_t strength = (int16_t)s->strength_x32;
    const size_t plane = (size_t)CW * CH;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const uint8_t *p = cv + (size_t)c * plane + (size_t)(y + 1) * CW + (x + 1);
        uint8_t uy[3];
#pragma unroll
        for (int d = -1; d <= 1; d++) uy[d + 1] = avg8(avg8(p[d - CW], p[d + CW]), p[d]);
        uint8_t unsharp = avg8(avg8(uy[0], uy[2]), uy[1]);
        int16_t mask = (int16_t)((int16_t)p[0] - (int16_t)unsharp);
        int16_t prod = (int16_t)(mask * strength);      // int16 (x) uint8 -> int16, wraps (src/IROperator.cpp:769-816)
        int16_t q = (int16_t)(prod >> 5);               // floor division by 32
        int16_t v = (int16_t)((int16_t)p[0] + q);
        out[(long)y * out_sy + x + (long)c * out_sc] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
    }
}

// cp_sharpen for aligned geometry (row pitch and output strides multiples of 4): a lane owns 4 adjacent pixels.
// Rows arrive as aligned dwords; the rounding byte averages avg8(a, b) = (a + b + 1) >> 1 of :398-404 are v_lerp_u8 with
// the rounding bit set in every byte; the horizontal neighbours are v_alignbyte shifts of the two dwords of a row; the
// int16 part (mask, product, floor /32, saturating cast) runs on packed i16 pairs (even bytes / odd bytes).
__global__ __launch_bounds__(256) void cp_sharpen4(const uint8_t *__restrict__ cv, int CW, int CH, const CPSetup *__restrict__ s,
                                                  uint8_t *__restrict__ out, long out_sy, long out_sc, int W, int H) {
    const int x = 4 * (blockIdx.x * blockDim.x + threadIdx.x), y = blockIdx.y;
    if (x >= W) return;
    const short st = (short)s->strength_x32;
    const i16x2 strength = {st, st}, zero = {0, 0}, top = {255, 255};
    const uint32_t one = 0x01010101u;
    const size_t plane = (size_t)CW * CH;
    auto avg4 = [&](uint32_t a, uint32_t b) { return __builtin_amdgcn_lerp(a, b, one); };
#pragma unroll
    for (int c = 0; c < 3; c++) {
        // cv columns x .. x+7 of rows y, y+1, y+2 (output pixel j sits at cv column x+1+j, row y+1)
        const uint32_t *r0 = reinterpret_cast<const uint32_t *>(cv + (size_t)c * plane + (size_t)y * CW + x);
        const uint32_t *r1 = reinterpret_cast<const uint32_t *>(cv + (size_t)c * plane + (size_t)(y + 1) * CW + x);
        const uint32_t *r2 = reinterpret_cast<const uint32_t *>(cv + (size_t)c * plane + (size_t)(y + 2) * CW + x);
        const uint32_t a0 = r0[0], b0 = r0[1], a1 = r1[0], b1 = r1[1], a2 = r2[0], b2 = r2[1];
        const uint32_t uya = avg4(avg4(a0, a2), a1), uyb = avg4(avg4(b0, b2), b1);            // unsharp_y, columns x..x+7
        const uint32_t L = uya, C = __builtin_amdgcn_alignbyte(uyb, uya, 1), R = __builtin_amdgcn_alignbyte(uyb, uya, 2);
        const uint32_t un = avg4(avg4(L, R), C);                                              // unsharp, 4 pixels
        const uint32_t P = __builtin_amdgcn_alignbyte(b1, a1, 1);                             // curved, 4 pixels
        auto sharpen2 = [&](uint32_t p2, uint32_t u2) -> uint32_t {                           // two pixels as i16 pairs
            const i16x2 p = __builtin_bit_cast(i16x2, p2), u = __builtin_bit_cast(i16x2, u2);
            const i16x2 mask = p - u;
            const i16x2 q = (i16x2)(mask * strength) >> 5;   // int16 product wraps (src/IROperator.cpp:769-816); floor /32
            i16x2 v = p + q;
            v = v < zero ? zero : v;
            v = v > top ? top : v;
            return __builtin_bit_cast(uint32_t, v);
        };
        const uint32_t even = sharpen2(P & 0x00ff00ffu, un & 0x00ff00ffu), odd = sharpen2((P >> 8) & 0x00ff00ffu, (un >> 8) & 0x00ff00ffu);
        *reinterpret_cast<uint32_t *>(out + (long)y * out_sy + x + (long)c * out_sc) = even | (odd << 8);
    }
}

const halide_type_t ty_u16 = {(decltype(halide_type_t::code))1, 16, 0};
const halide_type_t ty_u8 = {(decltype(halide_type_t::code))1, 8, 0};
const halide_type_t ty_f32 = {(decltype(halide_type_t::code))2, 32, 0};
const halide_type_t ty_i32 = {(decltype(halide_type_t::code))0, 32, 0};
const int64_t e0 = 0, e2592 = 2592, e1968 = 1968, e4 = 4, e3 = 3;
const int64_t *const est_in[4] = {&e0, &e2592, &e0, &e1968};
const int64_t *const est_m[4] = {&e0, &e4, &e0, &e3};
const int64_t *const est_out[6] = {&e0, &e2592, &e0, &e1968, &e0, &e3};
halide_scalar_value_t fval(float f) { halide_scalar_value_t v{}; v.u.f32 = f; return v; }
halide_scalar_value_t ival(int i) { halide_scalar_value_t v{}; v.u.i32 = i; return v; }
const halide_scalar_value_t est_ct = fval(3700), est_gamma = fval(2.0f), est_contrast = fval(50), est_sharp = fval(1.0f),
                            est_black = ival(25), est_white = ival(1023);
// estimates: generator :430-439
const halide_filter_argument_t cp_args[10] = {
    {"input", halide_argument_kind_input_buffer, 2, ty_u16, nullptr, nullptr, nullptr, nullptr, est_in},
    {"matrix_3200", halide_argument_kind_input_buffer, 2, ty_f32, nullptr, nullptr, nullptr, nullptr, est_m},
    {"matrix_7000", halide_argument_kind_input_buffer, 2, ty_f32, nullptr, nullptr, nullptr, nullptr, est_m},
    {"color_temp", halide_argument_kind_input_scalar, 0, ty_f32, nullptr, nullptr, nullptr, &est_ct, nullptr},
    {"gamma", halide_argument_kind_input_scalar, 0, ty_f32, nullptr, nullptr, nullptr, &est_gamma, nullptr},
    {"contrast", halide_argument_kind_input_scalar, 0, ty_f32, nullptr, nullptr, nullptr, &est_contrast, nullptr},
    {"sharpen_strength", halide_argument_kind_input_scalar, 0, ty_f32, nullptr, nullptr, nullptr, &est_sharp, nullptr},
    {"blackLevel", halide_argument_kind_input_scalar, 0, ty_i32, nullptr, nullptr, nullptr, &est_black, nullptr},
    {"whiteLevel", halide_argument_kind_input_scalar, 0, ty_i32, nullptr, nullptr, nullptr, &est_white, nullptr},
    {"processed", halide_argument_kind_output_buffer, 3, ty_u8, nullptr, nullptr, nullptr, nullptr, est_out},
};
const halide_filter_metadata_t cp_md = {1, 10, cp_args, kTargetString, "camera_pipe"};

// ---- cache of set-up blocks ---------------------------------------------------------------------------------------
// cp_setup's output (matrix, curve, strength) is a function of the two matrix buffers' contents and six scalars; a video
// stream calls with the same ones frame after frame.  The block is kept per (matrix allocations and versions, scalars) in
// memory of its own (the per-stream arena is shared with other pipelines) and the launch is skipped when nothing changed.
// Matrices in memory the runtime does not own (version 0) are never cached.
struct SetupKey {
    int device;
    uint64_t h3, v3, h7, v7;
    long s3, s7, o3, o7;
    float color_temp, gamma, contrast, sharpen;
    int black, white;
};
struct SetupImage {
    SetupKey key;
    bool valid = false;
    CPSetup *dev = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ready = nullptr;
    uint64_t used = 0;
};
std::mutex g_si_mu;
SetupImage g_si[4];
uint64_t g_si_clock = 0;

}  // namespace

extern "C" int camera_pipe(halide_buffer_t *input, halide_buffer_t *matrix_3200, halide_buffer_t *matrix_7000, float color_temp,
                           float gamma, float contrast, float sharpen_strength, int32_t blackLevel, int32_t whiteLevel,
                           halide_buffer_t *processed) {
    void *uc = nullptr;
    BufArg args[4] = {{"input", input, T_U16, 2, false}, {"matrix_3200", matrix_3200, T_F32, 2, false},
                      {"matrix_7000", matrix_7000, T_F32, 2, false}, {"processed", processed, T_U8, 3, true}};
    int r = check_not_null(uc, args, 4);
    if (r) return r;
    if ((r = check_type_and_dims(uc, args, 4))) return r;
    // footprint of a W x H output (no boundary condition anywhere in the pipeline): sharpen +-1 px, demosaic +-1 Bayer
    // quad, hot-pixel +-2 raw px, all after the (16, 12) shift  =>  x in [ox+10, ox+W+21], y in [oy+6, oy+H+17]
    const int W = processed->dim[0].extent, H = processed->dim[1].extent;
    const int ox = processed->dim[0].min, oy = processed->dim[1].min;
    if (any_bounds_query(args, 4)) {
        int imin[2] = {ox + 10, oy + 6}, iext[2] = {W + 12, H + 12};
        int mmin[2] = {0, 0}, mext[2] = {4, 3};
        int omin[3] = {ox, oy, 0}, oext[3] = {W, H, 3};
        answer_query(input, imin, iext);
        answer_query(matrix_3200, mmin, mext);
        answer_query(matrix_7000, mmin, mext);
        answer_query(processed, omin, oext);
        return 0;
    }
    for (int i = 0; i < 4; i++)
        if ((r = check_shape(uc, args[i]))) return r;
    if ((r = check_equal(uc, "processed.min.2", processed->dim[2].min, "0", 0))) return r;          // bound(c, 0, 3), :454
    if ((r = check_equal(uc, "processed.extent.2", processed->dim[2].extent, "3", 3))) return r;
    if ((ox & 1) || (oy & 1)) {
        return report(uc, halide_error_code_constraint_violated,
                      "Constraint violated: processed.min.0 (%d) and processed.min.1 (%d) must be even (Bayer phase)", ox, oy);
    }
    if (W > 0 && H > 0) {
        if ((r = check_covers(uc, args[0], 0, ox + 10, W + 12)) || (r = check_covers(uc, args[0], 1, oy + 6, H + 12))) return r;
    }
    for (int i = 1; i <= 2; i++) {
        if ((r = check_covers(uc, args[i], 0, 0, 4)) || (r = check_covers(uc, args[i], 1, 0, 3))) return r;
    }
    DeviceCtx ctx;
    if ((r = acquire_device(uc, &ctx))) return r;
    for (int i = 0; i < 3; i++)
        if ((r = input_to_device(uc, ctx, args[i]))) return r;
    if ((r = output_on_device(uc, ctx, args[3]))) return r;
    if (W == 0 || H == 0) {
        mark_output_written(processed);
        return 0;
    }
    const int CWL = W + 2, CW = (CWL + 3 + 4) & ~3, CH = H + 2;   // logical width; row pitch: a multiple of 4 with >= 4 spare
                                                                 // columns (cp_sharpen4 reads two dwords per row)
    const size_t setup_bytes = (sizeof(CPSetup) + 255) & ~(size_t)255;
    void *ws = nullptr;
    // shifted(x, y) = input(x + 16, y + 12) in ABSOLUTE coordinates; output pixel (ox, oy) is X = 0 of the kernels
    const long in_sy = input->dim[1].stride;
    const uint16_t *raw = dev_ptr<uint16_t>(input) + (long)(oy + 12 - input->dim[1].min) * in_sy + (ox + 16 - input->dim[0].min);
    const bool one_launch = in_sy % 2 == 0 && (uintptr_t)raw % 4 == 0 && W % 2 == 0;   // cp_fused_tile's preconditions
    if ((r = get_workspace(uc, ctx, setup_bytes + (one_launch ? 0 : (size_t)3 * CW * CH + 256), &ws))) return r;   // the curved planes: only on the two-launch path
    CPSetup *setup = (CPSetup *)ws;
    uint8_t *cv = (uint8_t *)ws + setup_bytes;
    hipStream_t st = ctx.stream;
    const long mo3 = (long)matrix_3200->dim[1].min * matrix_3200->dim[1].stride + matrix_3200->dim[0].min;
    const long mo7 = (long)matrix_7000->dim[1].min * matrix_7000->dim[1].stride + matrix_7000->dim[0].min;
    const float *m3 = dev_ptr<float>(matrix_3200) - mo3;
    const float *m7 = dev_ptr<float>(matrix_7000) - mo7;
    bool have_setup = false;
    SetupKey key;
    memset(&key, 0, sizeof key);
    key.device = ctx.device;
    key.h3 = matrix_3200->device, key.v3 = buffer_version(matrix_3200), key.h7 = matrix_7000->device, key.v7 = buffer_version(matrix_7000);
    key.s3 = matrix_3200->dim[1].stride, key.s7 = matrix_7000->dim[1].stride, key.o3 = mo3, key.o7 = mo7;
    key.color_temp = color_temp, key.gamma = gamma, key.contrast = contrast, key.sharpen = sharpen_strength;
    key.black = blackLevel, key.white = whiteLevel;
    const bool cacheable = key.v3 != 0 && key.v7 != 0 && !env_flag("HLMI_CP_NO_SETUP_CACHE");
    std::unique_lock<std::mutex> si_lock(g_si_mu, std::defer_lock);
    SetupImage *slot = nullptr;
    if (cacheable) {
        si_lock.lock();
        for (auto &e : g_si) {
            if (e.valid && memcmp(&e.key, &key, sizeof key) == 0) {
                e.used = ++g_si_clock;
                if (e.stream != st) HLMI_HIP(uc, wait_done(st, e.ready));
                setup = e.dev, have_setup = true;
                break;
            }
        }
        if (!have_setup) {
            slot = &g_si[0];
            for (auto &e : g_si) {
                if (!e.dev) { slot = &e; break; }
                if (e.used < slot->used) slot = &e;
            }
            slot->valid = false;
            if (slot->dev && slot->key.device != ctx.device) {
                (void)hipFree(slot->dev);
                slot->dev = nullptr;
            }
            if (!slot->dev) HLMI_HIP(uc, hipMalloc((void **)&slot->dev, setup_bytes));
            else HLMI_HIP(uc, hipDeviceSynchronize());   // evicting a block (rare): launches on any stream may still read it
            if (!slot->ready) HLMI_HIP(uc, hipEventCreateWithFlags(&slot->ready, hipEventDisableTiming));
            slot->key = key, slot->stream = st, slot->used = ++g_si_clock;
            setup = slot->dev;
        } else {
            si_lock.unlock();
        }
    }
    if (!have_setup) {
        HLMI_LAUNCH(uc, "cp_setup", st, cp_setup, dim3(1), dim3(1024), 0, m3, (long)matrix_3200->dim[1].stride, m7,
                    (long)matrix_7000->dim[1].stride, color_temp, gamma, contrast, sharpen_strength, blackLevel, whiteLevel, setup);
        if (slot) {
            HLMI_HIP(uc, record_done(slot->ready, st));
            slot->valid = true;
            si_lock.unlock();
        }
    }
    const int nqx = floor_div(W, 2) + 2, nqy = floor_div(H, 2) + 2;
    const long o_sy = processed->dim[1].stride, o_sc = processed->dim[2].stride;
    uint8_t *dout = dev_ptr<uint8_t>(processed);
    if (one_launch) {
        const int dwords = o_sy % 4 == 0 && o_sc % 4 == 0 && (uintptr_t)dout % 4 == 0;
        HLMI_LAUNCH(uc, "cp_fused", st, cp_fused_tile, dim3(((W + 63) / 64) * ((H + 2 * FTY - 1) / (2 * FTY))), dim3(256), 0, raw, in_sy, setup, dout, o_sy, o_sc,
                    W, H, dwords, (W + 63) / 64);
        mark_output_written(processed);
        return 0;
    }
    // any other geometry: one thread per quad from global memory, the curved planes through the workspace, a sharpen launch
    if (in_sy % 2 == 0 && (uintptr_t)raw % 4 == 0) {
        HLMI_LAUNCH(uc, "cp_demosaic", st, cp_demosaic<true>, dim3((nqx + 255) / 256, nqy), dim3(256), 0, raw, in_sy, setup, cv, CW, CH, CWL, nqx, nqy);
    } else {
        HLMI_LAUNCH(uc, "cp_demosaic", st, cp_demosaic<false>, dim3((nqx + 255) / 256, nqy), dim3(256), 0, raw, in_sy, setup, cv, CW, CH, CWL, nqx, nqy);
    }
    if (W % 4 == 0 && o_sy % 4 == 0 && o_sc % 4 == 0 && (uintptr_t)dout % 4 == 0) {
        HLMI_LAUNCH(uc, "cp_sharpen", st, cp_sharpen4, dim3((W / 4 + 255) / 256, H), dim3(256), 0, cv, CW, CH, setup, dout, o_sy, o_sc, W, H);
    } else {
        HLMI_LAUNCH(uc, "cp_sharpen", st, cp_sharpen, dim3((W + 255) / 256, H), dim3(256), 0, cv, CW, CH, setup, dout, o_sy, o_sc, W, H);
    }
    mark_output_written(processed);
    return 0;
}

extern "C" int camera_pipe_argv(void **a) {
    return camera_pipe((halide_buffer_t *)a[0], (halide_buffer_t *)a[1], (halide_buffer_t *)a[2], *(float *)a[3], *(float *)a[4],
                       *(float *)a[5], *(float *)a[6], *(int32_t *)a[7], *(int32_t *)a[8], (halide_buffer_t *)a[9]);
}
extern "C" const halide_filter_metadata_t *camera_pipe_metadata(void) { return &cp_md; }
extern "C" int camera_pipe_auto_schedule(halide_buffer_t *input, halide_buffer_t *matrix_3200, halide_buffer_t *matrix_7000,
                                         float color_temp, float gamma, float contrast, float sharpen_strength,
                                         int32_t blackLevel, int32_t whiteLevel, halide_buffer_t *processed) {
    return camera_pipe(input, matrix_3200, matrix_7000, color_temp, gamma, contrast, sharpen_strength, blackLevel, whiteLevel,
                       processed);
}
