// bgu.hip — gfx950 implementation of the reference's bgu AOT pipeline (bilateral-guided upsampling).
// Algorithm: /root/reference/apps/bgu/bgu_generator.cpp:268-488; boundary:
//   int bgu(float r_sigma, int32_t s_sigma, halide_buffer_t *splat_loc, halide_buffer_t *values,
//           halide_buffer_t *slice_loc, halide_buffer_t *output)                       f32 [W,H,3] planar everywhere
// The reference's GPU schedule (:571-669) is seven launches; its histogram accumulates with atomics (:589-597), i.e. in
// no defined order.  Here the float sums run in the serial order of the CPU schedule (oracle/bgu_oracle.c), which
// makes the result reproducible bit for bit, and the pipeline is three launches:
//   bgu_hist   one workgroup per grid cell: the cell's s_sigma^2 low-res samples are staged in LDS, then one thread per
//              (intensity bin, Gram-matrix term) walks them in order and adds the ones that land in its bin; the cell's
//              histogram stays in LDS and leaves the kernel already blurred in z
//   bgu_fit    one workgroup per (32 cells of a grid row, intensity plane): the 7-tap blurs in y (rows straight from
//              L2) and x (LDS), then one thread per (cell, right-hand side) runs the 4x4 sqrt-free LDL' solve in registers
//   bgu_slice  the only stage that touches the full-resolution image (2 x 47 MB at 1536 x 2560): a workgroup owns a
//              256-pixel-wide strip of rows inside ONE grid row; the two rows of transforms it needs live in LDS, per
//              image row they are interpolated in y once (the reference's interpolated_matrix_y) into a table
//              [cell][z][12] that every pixel then reads at its own (z, z+1) with ds_read_b128; the x and z lerps run two
//              channels per instruction (v_pk_mul_f32 / v_pk_add_f32)
// fast_inverse (:170) is 1/x correctly rounded, as on the reference's CUDA path (src/runtime/ptx_dev.ll:61-66).
#include "hlmi_device_math.h"
#include "hlmi_internal.h"

using namespace hlmi;
using namespace hlmi::dev;

namespace {

constexpr int NC = 22;    // accumulated terms per cell and bin (:307-315)
constexpr int CH = 256;   // samples staged per pass of bgu_hist
constexpr int XT = 32;    // cells per bgu_fit workgroup
constexpr int TW = 256;   // pixels per bgu_slice workgroup row
constexpr int RB = 8;     // rows per bgu_slice workgroup

// a low-res input: edge-clamped in EVERY dimension to its own box (BoundaryConditions::repeat_edge, :270-271)
struct LowRes {
    const float *p;   // element (x0, y0, c0)
    int x0, y0, c0, w, h, c;
    long sy, sc;
};
struct BGeom {
    int s, big, nb, nz, zmax, nhz;
    float inv_r;
    int cx0, cy0, ncx, ncy, nhx, nhy;   // line / blurx cells; the histogram's box is 3 cells larger on every side
};

__device__ __forceinline__ float lr_at(const LowRes &L, int x, int y, int c) {
    x = clampi(x, L.x0, L.x0 + L.w - 1) - L.x0;
    y = clampi(y, L.y0, L.y0 + L.h - 1) - L.y0;
    c = clampi(c, L.c0, L.c0 + L.c - 1) - L.c0;
    return L.p[(long)c * L.sc + (long)y * L.sy + x];
}

// operands of term c as rows of the staged sample table {sr, sg, sb, vr, vg, vb, 1}: term = row[IA] * row[IB]
// (x * 1.0f is x, so the single-factor terms of :307-315 are products too)
__device__ const unsigned char IA[NC] = {0, 0, 0, 0, 1, 1, 1, 2, 2, 6, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5};
__device__ const unsigned char IB[NC] = {0, 1, 2, 6, 1, 2, 6, 2, 6, 6, 0, 1, 2, 6, 0, 1, 2, 6, 0, 1, 2, 6};

// the seven-tap filter (:333-359), in the generator's order; the centre weight is 1
__device__ __forceinline__ float tap7(float a, float b, float c, float d, float e, float f, float h) {
    const float t0 = 1.0f / 64, t1 = 1.0f / 27, t2 = 1.0f / 8;
    return dev::mad(h, t0, dev::mad(f, t1, dev::mad(e, t2, dev::mad(c, t2, dev::mad2(a, t0, b, t1)) + d)));   // left to right (:333-359)
}

// bz: [nhy][nhx][nz][22] = blurz on the histogram's box.  The cell's histogram [nhz][22] never leaves LDS (hb, dynamic).
constexpr int RS = CH + 4;   // row stride of the staged table: rows 16 bytes apart in bank space, so lanes reading different rows with
                             // ds_read_b128 do not collide
__global__ __launch_bounds__(256) void bgu_hist(LowRes S, LowRes V, BGeom g, float *__restrict__ bzg) {
    __shared__ __attribute__((aligned(16))) float smp[7][RS];
    __shared__ __attribute__((aligned(16))) int zis[CH];
    extern __shared__ float hb[];
    const int tid = threadIdx.x;
    const int cxa = g.cx0 - 3 + (int)blockIdx.x, cya = g.cy0 - 3 + (int)blockIdx.y;
    const int nsamp = g.s * g.s, npairs = g.nhz * NC;
    for (int pb = 0; pb < npairs; pb += 256) {
        const int p = pb + tid;
        const int z = p / NC, c = p - z * NC;                   // p >= npairs: z > zmax, never matches
        const float4 *ra = reinterpret_cast<const float4 *>(smp[IA[c]]), *rb = reinterpret_cast<const float4 *>(smp[IB[c]]);
        const int4 *zq = reinterpret_cast<const int4 *>(zis);
        float acc = 0.0f;                                       // :291
        for (int s0 = 0; s0 < nsamp; s0 += CH) {
            __syncthreads();
            const int si = s0 + tid;
            float sr = 0.0f, sg = 0.0f, sb = 0.0f, vr = 0.0f, vg = 0.0f, vb = 0.0f;
            int zi = -1;                                        // past the last sample: lands in no bin
            if (si < nsamp) {                                   // sample (r.x, r.y) = (si % s, si / s): r.x innermost (:529-532)
                const int ry = si / g.s, rx = si - ry * g.s;
                const int sx = cxa * g.s + rx - g.s / 2, sy = cya * g.s + ry - g.s / 2;   // :294
                sr = lr_at(S, sx, sy, 0), sg = lr_at(S, sx, sy, 1), sb = lr_at(S, sx, sy, 2);
                vr = lr_at(V, sx, sy, 0), vg = lr_at(V, sx, sy, 1), vb = lr_at(V, sx, sy, 2);
                const float pos = clampf(((sr + sg * 2.0f) + sb) * 0.25f, 0.0f, 1.0f);   // :281-284 as the simplifier folds it
                zi = (int)__builtin_rintf(pos * g.inv_r);                                 // :297, ties to even
            }
            smp[0][tid] = sr, smp[1][tid] = sg, smp[2][tid] = sb, smp[3][tid] = vr, smp[4][tid] = vg, smp[5][tid] = vb, smp[6][tid] = 1.0f;
            zis[tid] = zi;
            __syncthreads();
            const int n4 = (min(CH, nsamp - s0) + 3) >> 2;
#pragma unroll 4
            for (int q = 0; q < n4; q++) {                      // four samples per LDS instruction, added in order.  The sum is
                const float4 a = ra[q], b = rb[q];              // never -0 (it starts at +0), so adding +0 for the samples of
                const int4 zz = zq[q];                          // other bins leaves it as it is
                acc = acc + (zz.x == z ? a.x * b.x : 0.0f);
                acc = acc + (zz.y == z ? a.y * b.y : 0.0f);
                acc = acc + (zz.z == z ? a.z * b.z : 0.0f);
                acc = acc + (zz.w == z ? a.w * b.w : 0.0f);
            }
        }
        if (p < npairs) hb[p] = acc;
    }
    __syncthreads();
    // blurz (:336-342) for the planes that are sliced, z = 0 .. nb + 1; the histogram is 0 outside 0 .. zmax
    float *out = bzg + ((long)blockIdx.y * g.nhx + blockIdx.x) * (g.nz * NC);
    for (int t = tid; t < g.nz * NC; t += 256) {
        const int z = t / NC, c = t - z * NC;
        float v[7];
#pragma unroll
        for (int d = 0; d < 7; d++) {
            const int zz = z + d - 3;
            v[d] = (zz >= 0 && zz <= g.zmax) ? hb[zz * NC + c] : 0.0f;
        }
        out[t] = tap7(v[0], v[1], v[2], v[3], v[4], v[5], v[6]);
    }
}

// the solve of one cell for ONE of its three right-hand sides: solve_symmetric<4, 3> (:131-238) statement by statement —
// the LDL' factorisation of A (shared by the three, recomputed by each of the three threads of a cell) and the
// substitutions for column k.  b: the 22 blurred terms of the cell.
__device__ __forceinline__ void solve_column(const float *b, int k, float (&x)[4]) {
    const float lambda = 1e-1f;                                  // :406-414
    float A[4][4];
    A[0][0] = b[0] + lambda, A[0][1] = b[1], A[0][2] = b[2], A[0][3] = b[3];
    A[1][0] = b[1], A[1][1] = b[4] + lambda, A[1][2] = b[5], A[1][3] = b[6];
    A[2][0] = b[2], A[2][1] = b[5], A[2][2] = b[7] + lambda, A[2][3] = b[8];
    A[3][0] = b[3], A[3][1] = b[6], A[3][2] = b[8], A[3][3] = b[9] + lambda;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const float v = b[10 + 4 * k + j];
        x[j] = (j == k) ? v + lambda : v;                        // b(0,0), b(1,1), b(2,2) += lambda
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
        A[j][j] = 1.0f / A[j][j];                                // fast_inverse as on the CUDA path
#pragma unroll
        for (int i = j + 1; i < 4; i++) A[i][j] = A[i][j] * A[j][j];
#pragma unroll
        for (int i = j + 1; i < 4; i++) {
#pragma unroll
            for (int kk = j + 1; kk < 4; kk++) {
                if (kk < i) A[i][kk] = A[kk][i];
                else A[i][kk] = dev::msub(A[i][kk], A[kk][j], A[j][i]);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
#pragma unroll
        for (int i = 0; i < j; i++) x[j] = dev::msub(x[j], A[j][i], x[i]);
    }
#pragma unroll
    for (int j = 0; j < 4; j++) x[j] = x[j] * A[j][j];
#pragma unroll
    for (int j = 3; j >= 0; j--) {
#pragma unroll
        for (int i = j + 1; i < 4; i++) x[j] = dev::msub(x[j], A[i][j], x[i]);
    }
}

// line: [ncy][ncx][nz][12].  Small, latency-bound and executed once per compute unit: the code is kept short (rolled
// loops, one right-hand side per thread) because every instruction is also a cold instruction-cache fetch.
__global__ __launch_bounds__(512) void bgu_fit(const float *__restrict__ bzg, BGeom g, float *__restrict__ line) {
    __shared__ float by[XT + 6][NC];
    __shared__ float bx[XT][NC + 1];
    const int tid = threadIdx.x;
    const int x0c = blockIdx.x * XT, cyi = blockIdx.y, z = blockIdx.z;
    const int nx = min(XT, g.ncx - x0c);
    const long rs = (long)g.nhx * g.nz * NC;                     // one grid row of bzg
    for (int it = tid; it < (nx + 6) * NC; it += 512) {          // blury (:343-349): the seven rows straight from L2
        const int xx = it / NC, c = it - xx * NC;
        const float *q = bzg + (long)cyi * rs + ((long)(x0c + xx) * g.nz + z) * NC + c;
        by[xx][c] = tap7(q[0], q[rs], q[2 * rs], q[3 * rs], q[4 * rs], q[5 * rs], q[6 * rs]);
    }
    __syncthreads();
    for (int it = tid; it < nx * NC; it += 512) {                // blurx (:350-356)
        const int xx = it / NC, c = it - xx * NC;
        bx[xx][c] = tap7(by[xx][c], by[xx + 1][c], by[xx + 2][c], by[xx + 3][c], by[xx + 4][c], by[xx + 5][c], by[xx + 6][c]);
    }
    __syncthreads();
    if (tid < 3 * nx) {
        const int xx = tid / 3, k = tid - 3 * xx;
        float x[4];
        solve_column(bx[xx], k, x);
        float *l = line + (((long)cyi * g.ncx + (x0c + xx)) * g.nz + z) * 12 + 4 * k;   // :417-433: c = 4 k + j
#pragma unroll
        for (int j = 0; j < 4; j++) l[j] = x[j];
    }
}

struct SGeom {
    int ox0, oy0, ow, oh, nsub, nlc;       // nlc: grid cells a 256-pixel strip can touch
    long s_sy, s_sc, o_sy, o_sc;
};

typedef float f2 __attribute__((ext_vector_type(2)));   // v_pk_mul_f32 / v_pk_add_f32: two lerps per instruction, each lane
                                                         // rounding exactly as the scalar operation does
__device__ __forceinline__ f2 lerp2(f2 zero, f2 one, float w, float iw) {   // dev::lerpf on a pair
    return dev::CANON_FMA ? __builtin_elementwise_fma(zero, f2{iw, iw}, one * w) : zero * iw + one * w;
}

__device__ __forceinline__ void slice_pixel(const float *__restrict__ tab, int xs, int xl, float xf, float s0, float s1, float s2, int nb,
                                            float (&o)[3]) {
    const float val = clampf(((s0 + s1 * 2.0f) + s2) * 0.25f, 0.0f, 1.0f);   // :286-289 as folded, :459-460
    const float zv = val * (float)nb;
    const int zi = (int)zv;
    const float zf = zv - (float)zi;
    const float4 *t0 = reinterpret_cast<const float4 *>(tab + xl * xs + zi * 12);   // cell xl: planes zi, zi + 1 (24 floats in a row)
    const float4 *t1 = reinterpret_cast<const float4 *>(tab + (xl + 1) * xs + zi * 12);
    f2 a[12], b[12];                       // [cell xl: 0..5 | cell xl + 1: 6..11] x channel pairs, planes zi (a) and zi + 1 (b)
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const float4 u0 = t0[i], u1 = t0[3 + i], w0 = t1[i], w1 = t1[3 + i];
        a[2 * i] = f2{u0.x, u0.y}, a[2 * i + 1] = f2{u0.z, u0.w};
        b[2 * i] = f2{u1.x, u1.y}, b[2 * i + 1] = f2{u1.z, u1.w};
        a[6 + 2 * i] = f2{w0.x, w0.y}, a[6 + 2 * i + 1] = f2{w0.z, w0.w};
        b[6 + 2 * i] = f2{w1.x, w1.y}, b[6 + 2 * i + 1] = f2{w1.z, w1.w};
    }
    const float ixf = 1.0f - xf, izf = 1.0f - zf;
    f2 m[6];
#pragma unroll
    for (int i = 0; i < 6; i++) m[i] = lerp2(lerp2(a[i], a[6 + i], xf, ixf), lerp2(b[i], b[6 + i], xf, ixf), zf, izf);   // :452-455, :467-470
    const f2 s01 = f2{s0, s1}, s2one = f2{s2, 1.0f};
#pragma unroll
    for (int c = 0; c < 3; c++) {
        if (dev::CANON_FMA) {   // ((m0 s0 + m1 s1) + m2 s2) + m3 (:473-477): the first and the third product fused
            o[c] = clampf(dev::mad(m[2 * c + 1].x, s2, dev::mad2(m[2 * c].x, s0, m[2 * c].y, s1)) + m[2 * c + 1].y, 0.0f, 1.0f);
        } else {
            const f2 p = m[2 * c] * s01, q = m[2 * c + 1] * s2one;           // :473-477; m * 1 is m
            o[c] = clampf(p.x + p.y + q.x + q.y, 0.0f, 1.0f);
        }
    }
}

// slice_loc: element (ox0, oy0, 0); out likewise.  grid: (ceil(ow / TW), (ncy - 1) * nsub).  LDS tables are laid out like
// `line` itself, [cell][z][12]: filling them is a straight copy, the per-row y interpolation needs no index arithmetic, and
// a pixel's two planes of a cell are 24 consecutive floats; lanes with different z are 3 z 16-byte groups apart, which
// never collide (3 is odd).
__global__ __launch_bounds__(TW) void bgu_slice(const float *__restrict__ line, const float *__restrict__ sl, BGeom g, SGeom q,
                                                float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int xs = g.nz * 12, tsz = q.nlc * xs;
    float *lin = lds;                       // [2][nlc][nz][12]: the two grid rows this strip interpolates between
    float *tab = lds + 2 * tsz;             // [2][nlc][nz][12]: interpolated_matrix_y of one image row, double buffered
    const int tid = threadIdx.x;
    const int ci = blockIdx.y / q.nsub, sub = blockIdx.y - ci * q.nsub;
    const int yi = g.cy0 + ci;
    const int ya = max(q.oy0, yi * g.big + sub * RB), yb = min(min(q.oy0 + q.oh, (yi + 1) * g.big), yi * g.big + (sub + 1) * RB);
    if (ya >= yb) return;
    const int xa = q.ox0 + blockIdx.x * TW, xn = min(TW, q.ox0 + q.ow - xa);
    const int xi_lo = (int)floorf((float)xa / (float)g.big);
    {   // cells xi_lo .. xi_lo + nlc - 1, clipped to the grid (the clipped ones are never read)
        const int cxl = xi_lo - g.cx0, n4 = min(q.nlc, g.ncx - cxl) * (xs / 4);
        for (int dy = 0; dy < 2; dy++) {
            const float4 *src = reinterpret_cast<const float4 *>(line + ((long)(ci + dy) * g.ncx + cxl) * xs);
            float4 *dst = reinterpret_cast<float4 *>(lin + dy * tsz);
            for (int it = tid; it < n4; it += TW) dst[it] = src[it];
        }
    }
    const int x = xa + tid;
    float xf = (float)x / (float)g.big;                          // :449-451
    const int xi = (int)floorf(xf);
    xf = xf - (float)xi;
    const int xl = xi - xi_lo;
    const bool live = tid < xn;
    const float *sp = sl + (long)(ya - q.oy0) * q.s_sy + (x - q.ox0);
    float *op = out + (long)(ya - q.oy0) * q.o_sy + (x - q.ox0);
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f;
    if (live) s0 = sp[0], s1 = sp[q.s_sc], s2 = sp[2 * q.s_sc];
    __syncthreads();
    const float4 *l0 = reinterpret_cast<const float4 *>(lin), *l1 = reinterpret_cast<const float4 *>(lin + tsz);
    for (int y = ya; y < yb; y++) {
        float yf = (float)y / (float)g.big;                      // :441-443
        yf = yf - (float)(int)floorf(yf);
        const float iyf = 1.0f - yf;
        float *tb = tab + ((y - ya) & 1) * tsz;
        for (int it = tid; it < tsz / 4; it += TW) {             // interpolated_matrix_y for this image row (:444-447)
            const float4 u = l0[it], w = l1[it];
            reinterpret_cast<float4 *>(tb)[it] = float4{dev::mad2(u.x, iyf, w.x, yf), dev::mad2(u.y, iyf, w.y, yf), dev::mad2(u.z, iyf, w.z, yf), dev::mad2(u.w, iyf, w.w, yf)};
        }
        float n0 = 0.0f, n1 = 0.0f, n2 = 0.0f;                   // next row's pixel, in flight across the barrier
        if (live && y + 1 < yb) n0 = sp[q.s_sy], n1 = sp[q.s_sy + q.s_sc], n2 = sp[q.s_sy + 2 * q.s_sc];
        __syncthreads();
        if (live) {
            float o[3];
            slice_pixel(tb, xs, xl, xf, s0, s1, s2, g.nb, o);
            op[0] = o[0], op[q.o_sc] = o[1], op[2 * q.o_sc] = o[2];
        }
        s0 = n0, s1 = n1, s2 = n2;
        sp += q.s_sy, op += q.o_sy;
    }
}

// any grid / image geometry: every pixel fetches its 96 transform entries from `line` itself
__global__ __launch_bounds__(256) void bgu_slice_direct(const float *__restrict__ line, const float *__restrict__ sl, BGeom g, SGeom q,
                                                        float *__restrict__ out) {
    const int xo = blockIdx.x * 256 + threadIdx.x, yo = blockIdx.y;
    if (xo >= q.ow) return;
    const int x = q.ox0 + xo, y = q.oy0 + yo;
    float yf = (float)y / (float)g.big, xf = (float)x / (float)g.big;
    const int yi = (int)floorf(yf), xi = (int)floorf(xf);
    yf = yf - (float)yi, xf = xf - (float)xi;
    const float *sp = sl + (long)yo * q.s_sy + xo;
    const float s0 = sp[0], s1 = sp[q.s_sc], s2 = sp[2 * q.s_sc];
    const float val = clampf(((s0 + s1 * 2.0f) + s2) * 0.25f, 0.0f, 1.0f);
    const float zv = val * (float)g.nb;
    const int zi = (int)zv;
    const float zf = zv - (float)zi;
    auto L = [&](int cx, int cy, int z, int c) { return line[(((long)(cy - g.cy0) * g.ncx + (cx - g.cx0)) * g.nz + z) * 12 + c]; };
    float m[12];
    for (int c = 0; c < 12; c++) {
        float mz[2];
        for (int dz = 0; dz < 2; dz++) {
            const float y0 = lerpf(L(xi, yi, zi + dz, c), L(xi, yi + 1, zi + dz, c), yf);
            const float y1 = lerpf(L(xi + 1, yi, zi + dz, c), L(xi + 1, yi + 1, zi + dz, c), yf);
            mz[dz] = lerpf(y0, y1, xf);
        }
        m[c] = lerpf(mz[0], mz[1], zf);
    }
    float *op = out + (long)yo * q.o_sy + xo;
    for (int c = 0; c < 3; c++) op[c * q.o_sc] = clampf(dev::mad(m[4 * c + 2], s2, dev::mad2(m[4 * c], s0, m[4 * c + 1], s1)) + m[4 * c + 3], 0.0f, 1.0f);
}

const int64_t e0 = 0, e3 = 3, e192 = 192, e320 = 320, e1536 = 1536, e2560 = 2560;
const int64_t *const est_lo[6] = {&e0, &e192, &e0, &e320, &e0, &e3};     // generator :676-687
const int64_t *const est_hi[6] = {&e0, &e1536, &e0, &e2560, &e0, &e3};
halide_scalar_value_t mk_i(int v) { halide_scalar_value_t s{}; s.u.i32 = v; return s; }
halide_scalar_value_t mk_f(float v) { halide_scalar_value_t s{}; s.u.f32 = v; return s; }
const halide_scalar_value_t est_r = mk_f(1.0f / 8.0f), est_s = mk_i(16);  // :674-675
const halide_type_t ty_i32 = {(decltype(halide_type_t::code))0, 32, 0};
const halide_type_t ty_f32 = {(decltype(halide_type_t::code))2, 32, 0};
const halide_filter_argument_t bgu_args[6] = {
    {"r_sigma", halide_argument_kind_input_scalar, 0, ty_f32, nullptr, nullptr, nullptr, &est_r, nullptr},
    {"s_sigma", halide_argument_kind_input_scalar, 0, ty_i32, nullptr, nullptr, nullptr, &est_s, nullptr},
    {"splat_loc", halide_argument_kind_input_buffer, 3, ty_f32, nullptr, nullptr, nullptr, nullptr, est_lo},
    {"values", halide_argument_kind_input_buffer, 3, ty_f32, nullptr, nullptr, nullptr, nullptr, est_lo},
    {"slice_loc", halide_argument_kind_input_buffer, 3, ty_f32, nullptr, nullptr, nullptr, nullptr, est_hi},
    {"output", halide_argument_kind_output_buffer, 3, ty_f32, nullptr, nullptr, nullptr, nullptr, est_hi},
};
const halide_filter_metadata_t bgu_md = {1, 6, bgu_args, kTargetString, "bgu"};

int ceil_div_f(int a, int b) { return (int)ceilf((float)a / (float)b); }   // i32(ceil(f32(a) / b)), :275-278

}  // namespace

extern "C" int bgu(float r_sigma, int32_t s_sigma, halide_buffer_t *splat_loc, halide_buffer_t *values, halide_buffer_t *slice_loc,
                   halide_buffer_t *output) {
    void *uc = nullptr;
    BufArg args[4] = {{"splat_loc", splat_loc, T_F32, 3, false}, {"values", values, T_F32, 3, false},
                      {"slice_loc", slice_loc, T_F32, 3, false}, {"output", output, T_F32, 3, true}};
    int r = check_not_null(uc, args, 4);
    if (r) return r;
    if ((r = check_type_and_dims(uc, args, 4))) return r;
    const int ox0 = output->dim[0].min, oy0 = output->dim[1].min, ow = output->dim[0].extent, oh = output->dim[1].extent;
    if (any_bounds_query(args, 4)) {
        // slice_loc is read exactly where the output is written (:286-289, :473-477); the low-res pair is read only
        // through repeat_edge, so any non-empty box will do: propose the eighth-size one apps/bgu/filter.cpp:37-38 uses
        int mins[3] = {ox0, oy0, 0}, ext[3] = {ow, oh, 3};
        answer_query(slice_loc, mins, ext);
        answer_query(output, mins, ext);
        int lmins[3] = {0, 0, 0}, lext[3] = {max(1, ow / 8), max(1, oh / 8), 3};
        answer_query(splat_loc, lmins, lext);
        answer_query(values, lmins, lext);
        return 0;
    }
    for (int i = 0; i < 4; i++)
        if ((r = check_shape(uc, args[i]))) return r;
    // the schedule bounds the output's channels to [0, 3) (:566, :648)
    if ((r = check_equal(uc, "output.min.2", output->dim[2].min, "0", 0)) || (r = check_equal(uc, "output.extent.2", output->dim[2].extent, "3", 3)))
        return r;
    const bool work = ow > 0 && oh > 0;
    if (work) {
        if ((r = check_covers(uc, args[2], 0, ox0, ow)) || (r = check_covers(uc, args[2], 1, oy0, oh)) || (r = check_covers(uc, args[2], 2, 0, 3)))
            return r;
    }
    if ((r = checks_done(uc))) return r;
    if (work) {
        for (int i = 0; i < 2; i++) {
            const halide_buffer_t *b = args[i].buf;
            if (b->dim[0].extent < 1 || b->dim[1].extent < 1 || b->dim[2].extent < 1) {
                return report(uc, halide_error_code_access_out_of_bounds, "Input buffer %s is empty but is accessed (clamped)", args[i].name);
            }
        }
        // arguments the generator gives no meaning to (an empty reduction domain and a division by zero, :292, :441)
        if (s_sigma < 1) return report(uc, halide_error_code_requirement_failed, "bgu: s_sigma is %d but must be at least 1", s_sigma);
        if (!(r_sigma > 0.0f) || !(1.0f / r_sigma <= 256.0f)) {   // a cell's histogram (22 floats per bin) stays in LDS
            return report(uc, halide_error_code_requirement_failed, "bgu: r_sigma is %g but must be in [1/256, inf)", (double)r_sigma);
        }
    }
    DeviceCtx ctx;
    if ((r = acquire_device(uc, &ctx))) return r;
    for (int i = 0; i < 3; i++)
        if ((r = input_to_device(uc, ctx, args[i]))) return r;
    if ((r = output_on_device(uc, ctx, args[3]))) return r;
    if (!work) {
        mark_output_written(output);
        return 0;
    }
    // ---- geometry (oracle/bgu_oracle.c)
    const int ufx = ceil_div_f(slice_loc->dim[0].extent, splat_loc->dim[0].extent), ufy = ceil_div_f(slice_loc->dim[1].extent, splat_loc->dim[1].extent);
    const long bigl = (long)s_sigma * max(ufx, ufy);
    // cell indices are floor(float(x) / float(big_sigma)) (:441-451): exact for coordinates below 2^20
    if (bigl < 1 || bigl >= (1 << 20) || abs(ox0) >= (1 << 20) || abs(oy0) >= (1 << 20) || ow >= (1 << 20) || oh >= (1 << 20)) {
        return report(uc, halide_error_code_buffer_extents_too_large, "bgu: coordinates or the cell size (%ld) exceed 2^20", bigl);
    }
    BGeom g;
    g.s = s_sigma, g.big = (int)bigl;
    g.inv_r = 1.0f / r_sigma;
    g.nb = (int)(1.0f / r_sigma);
    g.zmax = (int)rintf(g.inv_r);
    g.nz = g.nb + 2, g.nhz = g.zmax + 1;
    auto cell = [&](int v) { return (int)floorf((float)v / (float)g.big); };
    g.cx0 = cell(ox0), g.cy0 = cell(oy0);
    g.ncx = cell(ox0 + ow - 1) + 1 - g.cx0 + 1, g.ncy = cell(oy0 + oh - 1) + 1 - g.cy0 + 1;
    g.nhx = g.ncx + 6, g.nhy = g.ncy + 6;
    auto low = [](const halide_buffer_t *b) {
        LowRes L;
        L.p = dev_ptr<float>(b);
        L.x0 = b->dim[0].min, L.y0 = b->dim[1].min, L.c0 = b->dim[2].min;
        L.w = b->dim[0].extent, L.h = b->dim[1].extent, L.c = b->dim[2].extent;
        L.sy = b->dim[1].stride, L.sc = b->dim[2].stride;
        return L;
    };
    const LowRes S = low(splat_loc), V = low(values);
    auto al = [](size_t n) { return (n + 63) & ~(size_t)63; };
    const size_t n_hist = al((size_t)g.nhy * g.nhx * g.nz * NC), n_line = al((size_t)g.ncy * g.ncx * g.nz * 12);
    if ((n_hist + n_line) * sizeof(float) > ((size_t)1 << 32)) {
        return report(uc, halide_error_code_buffer_allocation_too_large, "bgu: the grid needs %zu bytes", (n_hist + n_line) * sizeof(float));
    }
    void *ws = nullptr;
    if ((r = get_workspace(uc, ctx, (n_hist + n_line) * sizeof(float), &ws))) return r;
    float *hist = (float *)ws, *line = hist + n_hist;
    hipStream_t st = ctx.stream;
    HLMI_LAUNCH(uc, "bgu_hist", st, bgu_hist, dim3(g.nhx, g.nhy), dim3(256), (size_t)g.nhz * NC * sizeof(float), S, V, g, hist);
    HLMI_LAUNCH(uc, "bgu_fit", st, bgu_fit, dim3((g.ncx + XT - 1) / XT, g.ncy, g.nz), dim3(512), 0, hist, g, line);
    SGeom q;
    q.ox0 = ox0, q.oy0 = oy0, q.ow = ow, q.oh = oh;
    q.nsub = (g.big + RB - 1) / RB;
    q.nlc = (TW - 1) / g.big + 3;
    q.s_sy = slice_loc->dim[1].stride, q.s_sc = slice_loc->dim[2].stride, q.o_sy = output->dim[1].stride, q.o_sc = output->dim[2].stride;
    const float *sl = dev_ptr<float>(slice_loc) + (long)(oy0 - slice_loc->dim[1].min) * q.s_sy + (ox0 - slice_loc->dim[0].min) +
                      (long)(0 - slice_loc->dim[2].min) * q.s_sc;
    const size_t lds = (size_t)4 * q.nlc * g.nz * 12 * sizeof(float);
    timing_note_bytes(24.0 * ow * oh);
    if (lds <= 64 * 1024) {   // (grids whose four cell rows do not fit: the direct kernel)
        if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void *)bgu_slice, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
        HLMI_LAUNCH(uc, "bgu_slice", st, bgu_slice, dim3((ow + TW - 1) / TW, (g.ncy - 1) * q.nsub), dim3(TW), lds, line, sl, g, q,
                    dev_ptr<float>(output));
    } else {
        HLMI_LAUNCH(uc, "bgu_slice_direct", st, bgu_slice_direct, dim3((ow + 255) / 256, oh), dim3(256), 0, line, sl, g, q, dev_ptr<float>(output));
    }
    mark_output_written(output);
    return 0;
}

extern "C" int bgu_argv(void **a) {
    return bgu(*(float *)a[0], *(int32_t *)a[1], (halide_buffer_t *)a[2], (halide_buffer_t *)a[3], (halide_buffer_t *)a[4], (halide_buffer_t *)a[5]);
}
extern "C" const halide_filter_metadata_t *bgu_metadata(void) { return &bgu_md; }
extern "C" int bgu_auto_schedule(float r_sigma, int32_t s_sigma, halide_buffer_t *splat_loc, halide_buffer_t *values, halide_buffer_t *slice_loc,
                                 halide_buffer_t *output) {
    return bgu(r_sigma, s_sigma, splat_loc, values, slice_loc, output);
}
