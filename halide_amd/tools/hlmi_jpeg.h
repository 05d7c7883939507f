// hlmi_jpeg.h — baseline JPEG for the command-line runner (hlmi_rungen): a decoder and (further down) an encoder.  The reference's image I/O reads JPEG through
// libjpeg with its default settings (tools/halide_image_io.h:1506-1548 load_jpg: jpeg_read_header, jpeg_start_decompress,
// 8-bit scanlines of output_components samples); this image has no libjpeg headers, so the decoder is written out here
// against ITU-T T.81 and does what libjpeg's defaults do, step for step, so that the samples are the ones libjpeg returns:
//   * sequential DCT, Huffman coding, 8-bit precision, one (gray) or three (YCbCr) components, restart intervals;
//   * the "slow but accurate" integer inverse DCT (13-bit constants, two passes with 2 guard bits between them);
//   * chroma sampled 2:1 horizontally or 2:1 in both directions comes back through the triangle filter of libjpeg's
//     "fancy upsampling" (3/4 nearer + 1/4 further sample, the two roundings alternating), edges by replication;
//   * YCbCr -> RGB with 16-bit fixed-point tables (1.402, 1.772, 0.71414, 0.34414) and the usual range limiting.
// tests/test_rungen.py compares the output with what libjpeg-turbo (through Pillow) returned for the files in
// tests/golden/jpeg/ — bit for bit.  Progressive, arithmetic-coded, 12-bit and CMYK files are refused.
#pragma once

#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include <vector>

namespace hlmi_jpeg {

struct Image {
    uint32_t width = 0, height = 0;
    int channels = 0;              // 1 gray, 3 RGB
    std::vector<uint8_t> bytes;    // row-major, channels interleaved
    unsigned at(uint32_t x, uint32_t y, int c) const { return bytes[((size_t)y * width + x) * channels + c]; }
};

namespace detail {

struct Huff {
    // canonical code tables of T.81 Annex C / F.2.2.3: for each code length the smallest code, the largest code and the
    // index of its first symbol
    int mincode[17], maxcode[18], valptr[17];
    uint8_t vals[256];
    bool present = false;
};

struct Component {
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
    int wblocks = 0, hblocks = 0;       // blocks stored per row / column (padded to whole MCUs)
    int dw = 0, dh = 0;                 // true downsampled size: ceil(image * h / hmax)
    std::vector<uint8_t> plane;         // wblocks * 8 x hblocks * 8 samples after the inverse DCT
    int pred = 0;
};

struct BitReader {
    const uint8_t *p, *end;
    uint32_t acc = 0;
    int n = 0;
    bool hit_marker = false, exhausted = false;
    BitReader(const uint8_t *b, const uint8_t *e) : p(b), end(e) {}
    void fill() {
        while (n <= 24) {
            unsigned byte = 0;
            if (!hit_marker && p >= end) exhausted = true;    // the file ends inside the scan
            if (!hit_marker && p < end) {
                byte = *p;
                if (byte == 0xff) {
                    if (p + 1 < end && p[1] == 0x00) p += 2;      // a stuffed zero follows every data byte ff
                    else { hit_marker = true; byte = 0; }         // a marker: the segment is over, feed zeros
                } else {
                    p++;
                }
            }
            acc |= byte << (24 - n);
            n += 8;
        }
    }
    int bits(int k) {   // k <= 16
        if (k == 0) return 0;
        fill();
        const int v = (int)(acc >> (32 - k));
        acc <<= k, n -= k;
        return v;
    }
    int peek16() { fill(); return (int)(acc >> 16); }
    void skip(int k) { acc <<= k, n -= k; }
    void reset() { acc = 0, n = 0, hit_marker = false; }
};

inline int extend(int v, int t) { return v < (1 << (t - 1)) ? v - (1 << t) + 1 : v; }   // T.81 F.2.2.1

inline bool decode_symbol(BitReader &br, const Huff &h, int &sym) {
    const int look = br.peek16();
    int code = 0;
    for (int len = 1; len <= 16; len++) {
        code = look >> (16 - len);
        if (h.maxcode[len] >= 0 && code <= h.maxcode[len] && code >= h.mincode[len]) {
            br.skip(len);
            sym = h.vals[h.valptr[len] + code - h.mincode[len]];
            return true;
        }
    }
    return false;
}

// libjpeg's range-limit table, as a function: index (x & 1023) of a table that holds x + 128 for -128 <= x < 128, 255 up to 511
// and 0 for what wrapped around from below
inline uint8_t range_limit(int x) {
    x &= 1023;
    if (x < 128) return (uint8_t)(x + 128);
    if (x < 512) return 255;
    if (x < 896) return 0;
    return (uint8_t)(x - 896);
}

// jidctint.c: the accurate integer inverse DCT (Loeffler, Ligtenberg and Moschytz), CONST_BITS = 13, PASS1_BITS = 2
inline void idct_islow(const int *coef /* dequantized, natural order */, uint8_t *out, int stride) {
    constexpr int CB = 13, P1 = 2;
    constexpr long F_0_298631336 = 2446, F_0_390180644 = 3196, F_0_541196100 = 4433, F_0_765366865 = 6270, F_0_899976223 = 7373,
                   F_1_175875602 = 9633, F_1_501321110 = 12299, F_1_847759065 = 15137, F_1_961570560 = 16069, F_2_053119869 = 16819,
                   F_2_562915447 = 20995, F_3_072711026 = 25172;
    auto descale = [](long x, int n) { return (x + (1L << (n - 1))) >> n; };
    long ws[64];
    for (int c = 0; c < 8; c++) {
        const int *in = coef + c;
        long z2 = in[16], z3 = in[48];
        long z1 = (z2 + z3) * F_0_541196100;
        long tmp2 = z1 + z3 * (-F_1_847759065), tmp3 = z1 + z2 * F_0_765366865;
        z2 = in[0], z3 = in[32];
        long tmp0 = (z2 + z3) * (1L << CB), tmp1 = (z2 - z3) * (1L << CB);
        const long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        tmp0 = in[56], tmp1 = in[40], tmp2 = in[24], tmp3 = in[8];
        z1 = tmp0 + tmp3, z2 = tmp1 + tmp2, z3 = tmp0 + tmp2;
        long z4 = tmp1 + tmp3;
        const long z5 = (z3 + z4) * F_1_175875602;
        tmp0 *= F_0_298631336, tmp1 *= F_2_053119869, tmp2 *= F_3_072711026, tmp3 *= F_1_501321110;
        z1 *= -F_0_899976223, z2 *= -F_2_562915447, z3 *= -F_1_961570560, z4 *= -F_0_390180644;
        z3 += z5, z4 += z5;
        tmp0 += z1 + z3, tmp1 += z2 + z4, tmp2 += z2 + z3, tmp3 += z1 + z4;
        ws[c] = descale(tmp10 + tmp3, CB - P1), ws[56 + c] = descale(tmp10 - tmp3, CB - P1);
        ws[8 + c] = descale(tmp11 + tmp2, CB - P1), ws[48 + c] = descale(tmp11 - tmp2, CB - P1);
        ws[16 + c] = descale(tmp12 + tmp1, CB - P1), ws[40 + c] = descale(tmp12 - tmp1, CB - P1);
        ws[24 + c] = descale(tmp13 + tmp0, CB - P1), ws[32 + c] = descale(tmp13 - tmp0, CB - P1);
    }
    for (int r = 0; r < 8; r++) {
        const long *w = ws + 8 * r;
        long z2 = w[2], z3 = w[6];
        long z1 = (z2 + z3) * F_0_541196100;
        long tmp2 = z1 + z3 * (-F_1_847759065), tmp3 = z1 + z2 * F_0_765366865;
        long tmp0 = (w[0] + w[4]) * (1L << CB), tmp1 = (w[0] - w[4]) * (1L << CB);
        const long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        tmp0 = w[7], tmp1 = w[5], tmp2 = w[3], tmp3 = w[1];
        z1 = tmp0 + tmp3, z2 = tmp1 + tmp2, z3 = tmp0 + tmp2;
        long z4 = tmp1 + tmp3;
        const long z5 = (z3 + z4) * F_1_175875602;
        tmp0 *= F_0_298631336, tmp1 *= F_2_053119869, tmp2 *= F_3_072711026, tmp3 *= F_1_501321110;
        z1 *= -F_0_899976223, z2 *= -F_2_562915447, z3 *= -F_1_961570560, z4 *= -F_0_390180644;
        z3 += z5, z4 += z5;
        tmp0 += z1 + z3, tmp1 += z2 + z4, tmp2 += z2 + z3, tmp3 += z1 + z4;
        uint8_t *o = out + (size_t)r * stride;
        constexpr int S = CB + P1 + 3;
        o[0] = range_limit((int)descale(tmp10 + tmp3, S)), o[7] = range_limit((int)descale(tmp10 - tmp3, S));
        o[1] = range_limit((int)descale(tmp11 + tmp2, S)), o[6] = range_limit((int)descale(tmp11 - tmp2, S));
        o[2] = range_limit((int)descale(tmp12 + tmp1, S)), o[5] = range_limit((int)descale(tmp12 - tmp1, S));
        o[3] = range_limit((int)descale(tmp13 + tmp0, S)), o[4] = range_limit((int)descale(tmp13 - tmp0, S));
    }
}

const uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                             41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                             30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// a component plane brought to full resolution, `w` x `h` samples (libjpeg's jdsample.c: fullsize, h2v1_fancy, h2v2_fancy)
inline std::string upsample(const Component &c, int hmax, int vmax, int w, int h, std::vector<uint8_t> &out) {
    out.assign((size_t)w * h, 0);
    const int pitch = c.wblocks * 8;
    const int hx = hmax / c.h, vx = vmax / c.v;
    if (hmax % c.h || vmax % c.v) return "fractional sampling ratios are not supported";
    auto row = [&](int y) { return c.plane.data() + (size_t)(y < 0 ? 0 : y >= c.dh ? c.dh - 1 : y) * pitch; };   // edge rows repeat
    if (hx == 1 && vx == 1) {
        for (int y = 0; y < h; y++) memcpy(&out[(size_t)y * w], row(y), (size_t)w);
        return "";
    }
    if ((hx == 2 && (vx == 1 || vx == 2)) && c.dw <= 2) {
        // libjpeg takes the triangle filter only for components more than two samples wide; narrower ones are replicated
        for (int y = 0; y < h; y++) {
            const uint8_t *in = row(y / vx);
            for (int x = 0; x < w; x++) out[(size_t)y * w + x] = in[x >> 1];
        }
        return "";
    }
    const int n = c.dw;   // input columns (> 2)
    std::vector<int> sum((size_t)n + 1);
    std::vector<uint8_t> line((size_t)2 * n + 2);
    if (hx == 2 && vx == 1) {
        for (int y = 0; y < h; y++) {
            const uint8_t *in = row(y);
            line[0] = in[0], line[1] = (uint8_t)((in[0] * 3 + in[1] + 2) >> 2);
            for (int i = 1; i < n - 1; i++) {
                const int v = in[i] * 3;
                line[2 * i] = (uint8_t)((v + in[i - 1] + 1) >> 2), line[2 * i + 1] = (uint8_t)((v + in[i + 1] + 2) >> 2);
            }
            line[2 * n - 2] = (uint8_t)((in[n - 1] * 3 + in[n - 2] + 1) >> 2), line[2 * n - 1] = in[n - 1];
            memcpy(&out[(size_t)y * w], line.data(), (size_t)w);
        }
        return "";
    }
    if (hx == 2 && vx == 2) {
        for (int y = 0; y < h; y++) {
            const int iy = y >> 1;
            const uint8_t *in0 = row(iy), *in1 = row((y & 1) ? iy + 1 : iy - 1);   // the nearer row, then the further one
            for (int i = 0; i < n; i++) sum[i] = in0[i] * 3 + in1[i];
            line[0] = (uint8_t)((sum[0] * 4 + 8) >> 4), line[1] = (uint8_t)((sum[0] * 3 + sum[1] + 7) >> 4);
            for (int i = 1; i < n - 1; i++) {
                line[2 * i] = (uint8_t)((sum[i] * 3 + sum[i - 1] + 8) >> 4);
                line[2 * i + 1] = (uint8_t)((sum[i] * 3 + sum[i + 1] + 7) >> 4);
            }
            line[2 * n - 2] = (uint8_t)((sum[n - 1] * 3 + sum[n - 2] + 8) >> 4), line[2 * n - 1] = (uint8_t)((sum[n - 1] * 4 + 7) >> 4);
            memcpy(&out[(size_t)y * w], line.data(), (size_t)w);
        }
        return "";
    }
    return "chroma sampling " + std::to_string(hx) + "x" + std::to_string(vx) + " is not supported (1x1, 2x1 and 2x2 are)";
}

}  // namespace detail

// returns "" on success, otherwise what is wrong with the file
inline std::string read(const std::string &path, Image &im) {
    using namespace detail;
    std::vector<uint8_t> b;
    {
        FILE *f = fopen(path.c_str(), "rb");
        if (!f) return "cannot open " + path;
        uint8_t buf[65536];
        size_t n;
        while ((n = fread(buf, 1, sizeof buf, f)) > 0) b.insert(b.end(), buf, buf + n);
        fclose(f);
    }
    if (b.size() < 4 || b[0] != 0xff || b[1] != 0xd8) return path + ": not a JPEG file";
    uint16_t qt[4][64];
    bool have_qt[4] = {false, false, false, false};
    Huff dc[4], ac[4];
    std::vector<Component> comps;
    int hmax = 1, vmax = 1, restart = 0;
    bool have_frame = false, adobe = false;
    int adobe_transform = -1;
    size_t pos = 2;
    auto u16 = [&](size_t o) { return (unsigned)b[o] << 8 | b[o + 1]; };
    for (;;) {
        if (pos + 4 > b.size()) return path + ": truncated (no scan)";
        if (b[pos] != 0xff) return path + ": marker expected";
        while (pos < b.size() && b[pos] == 0xff) pos++;        // fill bytes
        if (pos >= b.size()) return path + ": truncated";
        const int m = b[pos++];
        if (m == 0xd8 || (m >= 0xd0 && m <= 0xd7) || m == 0x01) continue;   // markers without a length
        if (m == 0xd9) return path + ": no image data";
        if (pos + 2 > b.size()) return path + ": truncated";
        const size_t len = u16(pos);
        if (len < 2 || pos + len > b.size()) return path + ": truncated segment";
        const uint8_t *s = &b[pos + 2];
        const size_t n = len - 2;
        if (m == 0xdb) {                                       // DQT
            size_t o = 0;
            while (o < n) {
                const int pq = s[o] >> 4, tq = s[o] & 15;
                o++;
                if (tq > 3 || pq > 1 || o + (pq ? 128 : 64) > n) return path + ": bad quantization table";
                for (int i = 0; i < 64; i++) qt[tq][i] = pq ? (uint16_t)(s[o + 2 * i] << 8 | s[o + 2 * i + 1]) : s[o + i];
                o += pq ? 128 : 64;
                have_qt[tq] = true;
            }
        } else if (m == 0xc4) {                                // DHT
            size_t o = 0;
            while (o < n) {
                if (o + 17 > n) return path + ": bad Huffman table";
                const int tc = s[o] >> 4, th = s[o] & 15;
                if (tc > 1 || th > 3) return path + ": bad Huffman table";
                Huff &h = tc ? ac[th] : dc[th];
                int total = 0, code = 0, k = 0;
                for (int l = 1; l <= 16; l++) total += s[o + l];
                if (total > 256 || o + 17 + total > n) return path + ": bad Huffman table";
                for (int l = 1; l <= 16; l++) {
                    const int cnt = s[o + l];
                    h.valptr[l] = k, h.mincode[l] = code;
                    code += cnt, k += cnt;
                    h.maxcode[l] = cnt ? code - 1 : -1;
                    if (code > (1 << l)) return path + ": bad Huffman table";
                    code <<= 1;
                }
                memcpy(h.vals, s + o + 17, (size_t)total);
                h.present = true;
                o += 17 + total;
            }
        } else if (m == 0xc0 || m == 0xc1) {                   // SOF0 / SOF1: sequential, Huffman
            if (n < 6) return path + ": bad frame header";
            if (s[0] != 8) return path + ": only 8-bit samples are supported";
            im.height = u16(pos + 3), im.width = u16(pos + 5);
            const int nc = s[5];
            if (!im.width || !im.height) return path + ": empty image";
            if (nc != 1 && nc != 3) return path + ": only gray and three-component files are supported";
            if (n < (size_t)6 + 3 * nc) return path + ": bad frame header";
            comps.resize(nc);
            for (int i = 0; i < nc; i++) {
                comps[i].id = s[6 + 3 * i], comps[i].h = s[7 + 3 * i] >> 4, comps[i].v = s[7 + 3 * i] & 15, comps[i].tq = s[8 + 3 * i];
                if (comps[i].h < 1 || comps[i].h > 4 || comps[i].v < 1 || comps[i].v > 4 || comps[i].tq > 3) return path + ": bad frame header";
                hmax = comps[i].h > hmax ? comps[i].h : hmax, vmax = comps[i].v > vmax ? comps[i].v : vmax;
            }
            have_frame = true;
        } else if (m == 0xc2) {
            return path + ": progressive JPEG files are not supported";
        } else if (m >= 0xc3 && m <= 0xcf && m != 0xc4 && m != 0xc8 && m != 0xcc) {
            return path + ": this JPEG process (lossless / hierarchical / arithmetic coding) is not supported";
        } else if (m == 0xdd) {                                // DRI
            if (n < 2) return path + ": bad restart interval";
            restart = (int)u16(pos + 2);
        } else if (m == 0xee && n >= 12 && !memcmp(s, "Adobe", 5)) {
            adobe = true, adobe_transform = s[11];
        } else if (m == 0xda) {                                // SOS: the one scan of a sequential file
            if (!have_frame) return path + ": scan before frame header";
            if (n < 1) return path + ": empty scan header";
            const int ns = s[0];
            if (ns != (int)comps.size() || n < (size_t)1 + 2 * ns + 3) return path + ": non-interleaved scans are not supported";
            for (int i = 0; i < ns; i++) {
                bool found = false;
                if ((s[2 + 2 * i] >> 4) > 3 || (s[2 + 2 * i] & 15) > 3) return path + ": bad table selector";   // dc[] / ac[] hold four tables
                for (auto &c : comps)
                    if (c.id == s[1 + 2 * i]) c.td = s[2 + 2 * i] >> 4, c.ta = s[2 + 2 * i] & 15, found = true;
                if (!found) return path + ": scan names an unknown component";
            }
            pos += len;
            break;
        }
        pos += len;
    }
    if (adobe && comps.size() == 3 && adobe_transform == 0) return path + ": Adobe RGB-coded JPEG files are not supported";
    // ---- geometry: MCUs of hmax * 8 x vmax * 8 pixels
    const int mcux = (int)((im.width + 8 * hmax - 1) / (8 * hmax)), mcuy = (int)((im.height + 8 * vmax - 1) / (8 * vmax));
    if ((double)mcux * mcuy * hmax * vmax * 64 * comps.size() > 4e9) return path + ": implausible dimensions";
    // the compressed data cannot describe more blocks than it has bits: refuse before allocating
    if ((double)mcux * mcuy > (double)(b.size() - pos) * 8 + 64) return path + ": truncated scan";
    for (auto &c : comps) {
        if (comps.size() == 1) c.h = c.v = 1;                  // a single-component scan is never interleaved: 1 x 1 blocks
        if (!have_qt[c.tq] || !dc[c.td].present || !ac[c.ta].present) return path + ": a table the scan refers to is missing";
    }
    if (comps.size() == 1) hmax = vmax = 1;
    const int mx = comps.size() == 1 ? (int)((im.width + 7) / 8) : mcux, my = comps.size() == 1 ? (int)((im.height + 7) / 8) : mcuy;
    for (auto &c : comps) {
        c.wblocks = mx * c.h, c.hblocks = my * c.v;
        c.dw = (int)((im.width * c.h + hmax - 1) / hmax), c.dh = (int)((im.height * c.v + vmax - 1) / vmax);
        c.plane.assign((size_t)c.wblocks * 8 * c.hblocks * 8, 0);
    }
    // ---- entropy-coded segment
    BitReader br(&b[pos], b.data() + b.size());
    int coef[64], left = restart, next_rst = 0;
    for (int my_ = 0; my_ < my; my_++) {
        for (int mx_ = 0; mx_ < mx; mx_++) {
            if (restart && left == 0) {                        // RSTn: byte-align, expect the marker, reset the predictions
                br.reset();
                const uint8_t *q = br.p;
                while (q + 1 < br.end && !(q[0] == 0xff && q[1] >= 0xd0 && q[1] <= 0xd7)) q++;
                if (q + 1 >= br.end || q[1] != 0xd0 + next_rst) return path + ": restart marker missing";
                br.p = q + 2, next_rst = (next_rst + 1) & 7, left = restart;
                for (auto &c : comps) c.pred = 0;
            }
            for (auto &c : comps) {
                for (int by = 0; by < c.v; by++) {
                    for (int bx = 0; bx < c.h; bx++) {
                        memset(coef, 0, sizeof coef);
                        int sym;
                        if (!decode_symbol(br, dc[c.td], sym) || sym > 11) return path + ": corrupt data (DC code)";
                        c.pred += sym ? extend(br.bits(sym), sym) : 0;
                        coef[0] = c.pred * qt[c.tq][0];
                        for (int k = 1; k < 64;) {
                            if (!decode_symbol(br, ac[c.ta], sym)) return path + ": corrupt data (AC code)";
                            const int r = sym >> 4, sz = sym & 15;
                            if (sz == 0) {
                                if (r != 15) break;             // end of block
                                k += 16;
                                continue;
                            }
                            k += r;
                            if (k > 63) return path + ": corrupt data (run past the block)";
                            coef[kZigzag[k]] = extend(br.bits(sz), sz) * qt[c.tq][k];
                            k++;
                        }
                        const size_t ox = ((size_t)mx_ * c.h + bx) * 8, oy = ((size_t)my_ * c.v + by) * 8;
                        idct_islow(coef, &c.plane[oy * ((size_t)c.wblocks * 8) + ox], c.wblocks * 8);
                    }
                }
            }
            if (restart) left--;
        }
    }
    if (br.exhausted) return path + ": truncated scan";
    // ---- to full resolution, then to the output colour space
    im.channels = (int)comps.size();
    im.bytes.assign((size_t)im.width * im.height * im.channels, 0);
    const int W = (int)im.width, H = (int)im.height;
    if (im.channels == 1) {
        const Component &c = comps[0];
        for (int y = 0; y < H; y++) memcpy(&im.bytes[(size_t)y * W], &c.plane[(size_t)y * c.wblocks * 8], (size_t)W);
        return "";
    }
    std::vector<uint8_t> full[3];
    for (int i = 0; i < 3; i++) {
        const std::string e = upsample(comps[i], hmax, vmax, W, H, full[i]);
        if (!e.empty()) return path + ": " + e;
    }
    // jdcolor.c: SCALEBITS = 16, FIX(x) = (int)(x * 65536 + 0.5)
    constexpr long F_1_40200 = 91881, F_1_77200 = 116130, F_0_71414 = 46802, F_0_34414 = 22554, HALF = 1L << 15;
    auto clamp8 = [](int v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); };
    for (size_t i = 0; i < (size_t)W * H; i++) {
        const int y = full[0][i], cb = full[1][i] - 128, cr = full[2][i] - 128;
        const int r = y + (int)((F_1_40200 * cr + HALF) >> 16);
        const int g = y + (int)((-F_0_34414 * cb + HALF - F_0_71414 * cr) >> 16);
        const int bl = y + (int)((F_1_77200 * cb + HALF) >> 16);
        im.bytes[3 * i] = clamp8(r), im.bytes[3 * i + 1] = clamp8(g), im.bytes[3 * i + 2] = clamp8(bl);
    }
    return "";
}

// ---- encoder: what libjpeg writes with jpeg_set_defaults + jpeg_set_quality(quality, TRUE) — the reference's save_jpg
// (tools/halide_image_io.h:1558-1610, quality 99): JFIF 1.01 header, the Annex K quantization tables scaled by the quality,
// YCbCr with the chroma averaged 2 x 2 (gray: one component), accurate integer forward DCT, the Annex K Huffman tables, one scan.
// Written step for step after libjpeg so that the FILE is the one libjpeg produces: tests/test_rungen.py compares with files
// libjpeg-turbo wrote (through Pillow) byte for byte.
namespace detail {

const uint8_t kStdLumQ[64] = {16, 11, 10, 16, 24,  40,  51,  61,  12, 12, 14, 19, 26,  58,  60,  55,  14, 13, 16, 24, 40,  57,
                              69, 56, 14, 17, 22,  29,  51,  87,  80, 62, 18, 22, 37,  56,  68,  109, 103, 77, 24, 35, 55,  64,
                              81, 104, 113, 92, 49, 64,  78,  87,  103, 121, 120, 101, 72, 92,  95,  98,  112, 100, 103, 99};
const uint8_t kStdChrQ[64] = {17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99,
                              99, 99, 47, 66, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
                              99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};
// T.81 Annex K.3: code-length counts and symbols of the four typical tables
const uint8_t kDcLumBits[16] = {0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0};
const uint8_t kDcChrBits[16] = {0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0};
const uint8_t kDcVals[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
const uint8_t kAcLumBits[16] = {0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d};
const uint8_t kAcLumVals[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14, 0x32, 0x81, 0x91, 0xa1,
    0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26,
    0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56,
    0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83, 0x84, 0x85,
    0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa,
    0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6,
    0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9,
    0xfa};
const uint8_t kAcChrBits[16] = {0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77};
const uint8_t kAcChrVals[162] = {
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22, 0x32, 0x81, 0x08, 0x14, 0x42,
    0x91, 0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18, 0x19,
    0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55,
    0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x82, 0x83,
    0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8,
    0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4,
    0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9,
    0xfa};

struct EncTable {
    uint16_t code[256];
    uint8_t size[256];
};
inline void build_enc(const uint8_t *bits, const uint8_t *vals, EncTable &t) {   // T.81 Annex C
    memset(&t, 0, sizeof t);
    int code = 0, k = 0;
    for (int l = 1; l <= 16; l++) {
        for (int i = 0; i < bits[l - 1]; i++, k++) t.code[vals[k]] = (uint16_t)code++, t.size[vals[k]] = (uint8_t)l;
        code <<= 1;
    }
}

// jfdctint.c: forward DCT, results scaled up by 8 (the quantizer divides by 8 q)
inline void fdct_islow(int *d) {
    constexpr int CB = 13, P1 = 2;
    constexpr long F_0_298631336 = 2446, F_0_390180644 = 3196, F_0_541196100 = 4433, F_0_765366865 = 6270, F_0_899976223 = 7373,
                   F_1_175875602 = 9633, F_1_501321110 = 12299, F_1_847759065 = 15137, F_1_961570560 = 16069, F_2_053119869 = 16819,
                   F_2_562915447 = 20995, F_3_072711026 = 25172;
    auto descale = [](long x, int n) { return (x + (1L << (n - 1))) >> n; };
    for (int pass = 0; pass < 2; pass++) {
        const int step = pass ? 8 : 1, next = pass ? 1 : 8;   // pass 0: along rows, pass 1: down columns
        for (int i = 0; i < 8; i++) {
            int *p = d + i * next;
            const long tmp0 = p[0] + p[7 * step], tmp7 = p[0] - p[7 * step], tmp1 = p[step] + p[6 * step], tmp6 = p[step] - p[6 * step];
            const long tmp2 = p[2 * step] + p[5 * step], tmp5 = p[2 * step] - p[5 * step], tmp3 = p[3 * step] + p[4 * step], tmp4 = p[3 * step] - p[4 * step];
            const long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
            if (pass == 0) p[0] = (int)((tmp10 + tmp11) * (1 << P1)), p[4 * step] = (int)((tmp10 - tmp11) * (1 << P1));
            else p[0] = (int)descale(tmp10 + tmp11, P1), p[4 * step] = (int)descale(tmp10 - tmp11, P1);
            const int S = pass == 0 ? CB - P1 : CB + P1;
            long z1 = (tmp12 + tmp13) * F_0_541196100;
            p[2 * step] = (int)descale(z1 + tmp13 * F_0_765366865, S), p[6 * step] = (int)descale(z1 + tmp12 * (-F_1_847759065), S);
            z1 = tmp4 + tmp7;
            long z2 = tmp5 + tmp6, z3 = tmp4 + tmp6, z4 = tmp5 + tmp7;
            const long z5 = (z3 + z4) * F_1_175875602;
            const long t4 = tmp4 * F_0_298631336, t5 = tmp5 * F_2_053119869, t6 = tmp6 * F_3_072711026, t7 = tmp7 * F_1_501321110;
            z1 *= -F_0_899976223, z2 *= -F_2_562915447, z3 *= -F_1_961570560, z4 *= -F_0_390180644;
            z3 += z5, z4 += z5;
            p[7 * step] = (int)descale(t4 + z1 + z3, S), p[5 * step] = (int)descale(t5 + z2 + z4, S);
            p[3 * step] = (int)descale(t6 + z2 + z3, S), p[step] = (int)descale(t7 + z1 + z4, S);
        }
    }
}

struct BitWriter {
    std::vector<uint8_t> &out;
    uint32_t acc = 0;
    int n = 0;
    explicit BitWriter(std::vector<uint8_t> &o) : out(o) {}
    void put(unsigned code, int size) {
        if (!size) return;
        acc |= (code & ((1u << size) - 1)) << (24 - n - size + 8);   // keep the pending bits left-aligned in a 32-bit window
        n += size;
        while (n >= 8) {
            const uint8_t b = (uint8_t)(acc >> 24);
            out.push_back(b);
            if (b == 0xff) out.push_back(0);
            acc <<= 8, n -= 8;
        }
    }
    void flush() { put(0x7f, 7); acc = 0, n = 0; }   // pad the last byte with ones
};

}  // namespace detail

// `im`: 1 (gray) or 3 (RGB) channels of 8-bit samples.  Returns "" on success.
inline std::string write(const std::string &path, const Image &im, int quality = 99) {
    using namespace detail;
    if (im.channels != 1 && im.channels != 3) return path + ": JPEG files hold 1 or 3 channels";
    if (!im.width || !im.height || im.width > 65535 || im.height > 65535) return path + ": JPEG dimensions are 1 .. 65535";
    const int W = (int)im.width, H = (int)im.height, nc = im.channels;
    quality = quality < 1 ? 1 : quality > 100 ? 100 : quality;
    const int scale = quality < 50 ? 5000 / quality : 200 - 2 * quality;
    uint8_t q[2][64];
    for (int t = 0; t < 2; t++)
        for (int i = 0; i < 64; i++) {
            long v = ((long)(t ? kStdChrQ[i] : kStdLumQ[i]) * scale + 50) / 100;
            q[t][i] = (uint8_t)(v < 1 ? 1 : v > 255 ? 255 : v);
        }
    // ---- component planes, padded as libjpeg pads them: columns by repeating the last one up to whole blocks (before the chroma
    // is averaged), rows by repeating the last row — of the full-size data up to an even count, of each plane up to whole MCUs
    const int hs = nc == 3 ? 2 : 1;                                     // luma sampling factor in both directions
    struct Plane {
        int wb, hb, wpad, hpad;                                         // blocks that exist; samples stored (whole MCUs down)
        std::vector<uint8_t> s;
    } pl[3];
    const int mcux = (W + 8 * hs - 1) / (8 * hs), mcuy = (H + 8 * hs - 1) / (8 * hs);
    const int Hfull = nc == 3 ? (H + 1) / 2 * 2 : H;                    // full-size rows after the bottom edge is repeated
    std::vector<uint8_t> ycc[3];
    for (int c = 0; c < nc; c++) ycc[c].assign((size_t)W * Hfull, 0);
    for (int y = 0; y < Hfull; y++) {
        const int sy = y < H ? y : H - 1;
        for (int x = 0; x < W; x++) {
            const uint8_t *p = &im.bytes[((size_t)sy * W + x) * nc];
            if (nc == 1) {
                ycc[0][(size_t)y * W + x] = p[0];
            } else {   // jccolor.c: SCALEBITS 16; the chroma rows carry ONE_HALF - 1 so that the maximum stays below 256
                const long r = p[0], g = p[1], b = p[2];
                ycc[0][(size_t)y * W + x] = (uint8_t)((19595 * r + 38470 * g + 7471 * b + 32768) >> 16);
                ycc[1][(size_t)y * W + x] = (uint8_t)((-11059 * r - 21709 * g + 32768 * b + (128L << 16) + 32767) >> 16);
                ycc[2][(size_t)y * W + x] = (uint8_t)((32768 * r - 27439 * g - 5329 * b + (128L << 16) + 32767) >> 16);
            }
        }
    }
    for (int c = 0; c < nc; c++) {
        Plane &P = pl[c];
        const int samp = c == 0 ? hs : 1, cw = (W * samp + hs - 1) / hs, chh = (H * samp + hs - 1) / hs;   // true component size
        P.wb = (cw + 7) / 8, P.hb = (chh + 7) / 8;
        P.wpad = P.wb * 8, P.hpad = mcuy * samp * 8;
        P.s.assign((size_t)P.wpad * P.hpad, 0);
        const int rows = c == 0 || nc == 1 ? Hfull : Hfull / 2;          // rows that come out of the (down)sampler
        for (int y = 0; y < P.hpad; y++) {
            const int sy = y < rows ? y : rows - 1;
            uint8_t *o = &P.s[(size_t)y * P.wpad];
            if (c == 0 || nc == 1) {
                const uint8_t *in = &ycc[c][(size_t)sy * W];
                for (int x = 0; x < P.wpad; x++) o[x] = in[x < W ? x : W - 1];
            } else {   // jcsample.c h2v2_downsample: the rounding bias alternates 1, 2 along a row
                const uint8_t *in0 = &ycc[c][(size_t)(2 * sy) * W], *in1 = &ycc[c][(size_t)(2 * sy + 1) * W];
                int bias = 1;
                for (int x = 0; x < P.wpad; x++, bias ^= 3) {
                    const int x0 = 2 * x < W ? 2 * x : W - 1, x1 = 2 * x + 1 < W ? 2 * x + 1 : W - 1;
                    o[x] = (uint8_t)((in0[x0] + in0[x1] + in1[x0] + in1[x1] + bias) >> 2);
                }
            }
        }
    }
    // ---- headers
    std::vector<uint8_t> f;
    auto put8 = [&](int v) { f.push_back((uint8_t)v); };
    auto put16 = [&](int v) { f.push_back((uint8_t)(v >> 8)), f.push_back((uint8_t)v); };
    put16(0xffd8);
    put16(0xffe0), put16(16), put8('J'), put8('F'), put8('I'), put8('F'), put8(0), put16(0x0101), put8(0), put16(1), put16(1), put8(0), put8(0);
    for (int t = 0; t < (nc == 3 ? 2 : 1); t++) {
        put16(0xffdb), put16(67), put8(t);
        for (int k = 0; k < 64; k++) put8(q[t][kZigzag[k]]);
    }
    put16(0xffc0), put16(8 + 3 * nc), put8(8), put16(H), put16(W), put8(nc);
    for (int c = 0; c < nc; c++) put8(c + 1), put8(c == 0 ? (hs << 4 | hs) : 0x11), put8(c ? 1 : 0);
    auto dht = [&](int tc_th, const uint8_t *bits, const uint8_t *vals, int nv) {
        put16(0xffc4), put16(19 + nv), put8(tc_th);
        for (int i = 0; i < 16; i++) put8(bits[i]);
        for (int i = 0; i < nv; i++) put8(vals[i]);
    };
    dht(0x00, kDcLumBits, kDcVals, 12), dht(0x10, kAcLumBits, kAcLumVals, 162);
    if (nc == 3) dht(0x01, kDcChrBits, kDcVals, 12), dht(0x11, kAcChrBits, kAcChrVals, 162);
    put16(0xffda), put16(6 + 2 * nc), put8(nc);
    for (int c = 0; c < nc; c++) put8(c + 1), put8(c ? 0x11 : 0x00);
    put8(0), put8(63), put8(0);
    // ---- the scan
    EncTable edc[2], eac[2];
    build_enc(kDcLumBits, kDcVals, edc[0]), build_enc(kAcLumBits, kAcLumVals, eac[0]);
    build_enc(kDcChrBits, kDcVals, edc[1]), build_enc(kAcChrBits, kAcChrVals, eac[1]);
    BitWriter bw(f);
    int pred[3] = {0, 0, 0};
    auto nbits_of = [](int v) { int n = 0; while (v) n++, v >>= 1; return n; };
    auto encode_block = [&](const int *zz /* quantized, natural order */, int c) {
        const EncTable &D = edc[c ? 1 : 0], &A = eac[c ? 1 : 0];
        int diff = zz[0] - pred[c];
        pred[c] = zz[0];
        int t = diff, t2 = diff;
        if (t < 0) t = -t, t2--;
        int nb = nbits_of(t);
        bw.put(D.code[nb], D.size[nb]);
        if (nb) bw.put((unsigned)t2, nb);
        int run = 0;
        for (int k = 1; k < 64; k++) {
            int v = zz[kZigzag[k]];
            if (v == 0) { run++; continue; }
            while (run > 15) bw.put(A.code[0xf0], A.size[0xf0]), run -= 16;
            t = v, t2 = v;
            if (t < 0) t = -t, t2--;
            nb = nbits_of(t);
            bw.put(A.code[(run << 4) + nb], A.size[(run << 4) + nb]);
            bw.put((unsigned)t2, nb);
            run = 0;
        }
        if (run > 0) bw.put(A.code[0], A.size[0]);
    };
    int blk[64], prev_dc[3] = {0, 0, 0};
    for (int my = 0; my < mcuy; my++) {
        for (int mx = 0; mx < mcux; mx++) {
            for (int c = 0; c < nc; c++) {
                const Plane &P = pl[c];
                const int samp = c == 0 ? hs : 1;
                for (int by = 0; by < samp; by++) {
                    for (int bx = 0; bx < samp; bx++) {
                        const int bxx = mx * samp + bx, byy = my * samp + by;
                        if (bxx < P.wb && byy < P.hb) {
                            const uint8_t *src = &P.s[(size_t)byy * 8 * P.wpad + (size_t)bxx * 8];
                            for (int i = 0; i < 64; i++) blk[i] = (int)src[(size_t)(i >> 3) * P.wpad + (i & 7)] - 128;
                            fdct_islow(blk);
                            const uint8_t *qt = q[c ? 1 : 0];
                            for (int i = 0; i < 64; i++) {   // jcdctmgr.c: divide by 8 q, rounding half away from zero
                                const int qv = qt[i] << 3;
                                int v = blk[i];
                                if (v < 0) v = -((-v + (qv >> 1)) / qv);
                                else v = (v + (qv >> 1)) / qv;
                                blk[i] = v;
                            }
                        } else {   // a block that only pads the MCU: no AC, the DC of the block coded before it in this MCU
                            memset(blk, 0, sizeof blk);
                            blk[0] = prev_dc[c];
                        }
                        prev_dc[c] = blk[0];
                        encode_block(blk, c);
                    }
                }
            }
        }
    }
    bw.flush();
    put16(0xffd9);
    FILE *fp = fopen(path.c_str(), "wb");
    if (!fp) return "cannot write " + path;
    const bool ok = fwrite(f.data(), 1, f.size(), fp) == f.size();
    fclose(fp);
    return ok ? "" : path + ": short write";
}

}  // namespace hlmi_jpeg
