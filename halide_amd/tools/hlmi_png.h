// hlmi_png.h — PNG files for the command-line runner (hlmi_rungen), over zlib: the formats the reference's image I/O
// reads and writes (tools/halide_image_io.h:856-1040 load_png / save_png: non-interlaced, 8 or 16 bits per sample, gray /
// gray+alpha / RGB / RGBA).  Decoder: every chunk's CRC is checked, IDAT chunks are concatenated and inflated, the five
// scanline filters (None, Sub, Up, Average, Paeth) are undone.  Encoder: filter 0, one IDAT.  Written against the PNG
// specification (ISO/IEC 15948); no libpng in this image.
#pragma once

#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include <vector>
#include <zlib.h>

namespace hlmi_png {

struct Image {
    uint32_t width = 0, height = 0;
    int channels = 0;      // 1 gray, 2 gray+alpha, 3 RGB, 4 RGBA
    int bit_depth = 0;     // 8 or 16
    // samples in file order: row-major, channels interleaved, 16-bit samples as host-order uint16_t pairs of bytes (see at())
    std::vector<uint8_t> bytes;
    unsigned at(uint32_t x, uint32_t y, int c) const {
        const size_t i = ((size_t)y * width + x) * channels + c;
        return bit_depth == 16 ? ((unsigned)bytes[2 * i] << 8) | bytes[2 * i + 1] : bytes[i];   // PNG is big-endian
    }
    void set(uint32_t x, uint32_t y, int c, unsigned v) {
        const size_t i = ((size_t)y * width + x) * channels + c;
        if (bit_depth == 16) bytes[2 * i] = (uint8_t)(v >> 8), bytes[2 * i + 1] = (uint8_t)(v & 255);
        else bytes[i] = (uint8_t)v;
    }
};

inline uint32_t be32(const uint8_t *b) { return ((uint32_t)b[0] << 24) | ((uint32_t)b[1] << 16) | ((uint32_t)b[2] << 8) | b[3]; }
inline void put_be32(uint8_t *b, uint32_t v) { b[0] = (uint8_t)(v >> 24), b[1] = (uint8_t)(v >> 16), b[2] = (uint8_t)(v >> 8), b[3] = (uint8_t)v; }

// returns "" on success, otherwise what is wrong with the file
inline std::string read(const std::string &path, Image &im) {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return "cannot open " + path;
    struct Close {
        FILE *f;
        ~Close() { fclose(f); }
    } closer{f};
    static const uint8_t sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
    uint8_t head[8];
    if (fread(head, 1, 8, f) != 8 || memcmp(head, sig, 8)) return path + ": not a PNG file";
    std::vector<uint8_t> z;
    bool have_ihdr = false, done = false;
    int color_type = 0;
    while (!done) {
        if (fread(head, 1, 8, f) != 8) return path + ": truncated (no IEND chunk)";
        const uint32_t len = be32(head);
        // nothing is allocated on a length field's word alone: a chunk cannot be longer than what is left of the file, and
        // only IDAT chunks may be large at all
        if (len > (1u << 31)) return path + ": chunk length out of range";
        {
            const long here = ftell(f);
            if (here < 0 || fseek(f, 0, SEEK_END) != 0) return path + ": cannot seek";
            const long end = ftell(f);
            if (fseek(f, here, SEEK_SET) != 0) return path + ": cannot seek";
            if ((long)len + 4 > end - here) return path + ": truncated chunk";
        }
        if (memcmp(head + 4, "IDAT", 4) && len > (1u << 20)) return path + ": oversized ancillary chunk";
        std::vector<uint8_t> body(len);
        uint8_t crc[4];
        if ((len && fread(body.data(), 1, len, f) != len) || fread(crc, 1, 4, f) != 4) return path + ": truncated chunk";
        uint32_t c = (uint32_t)crc32(0L, head + 4, 4);
        if (len) c = (uint32_t)crc32(c, body.data(), len);
        if (c != be32(crc)) return path + ": chunk CRC mismatch";
        if (!memcmp(head + 4, "IHDR", 4)) {
            if (len != 13) return path + ": bad IHDR";
            im.width = be32(body.data()), im.height = be32(body.data() + 4);
            im.bit_depth = body[8], color_type = body[9];
            im.channels = color_type == 0 ? 1 : color_type == 4 ? 2 : color_type == 2 ? 3 : color_type == 6 ? 4 : 0;
            if (!im.channels) return path + ": palette images are not supported (the reference's loader rejects them too)";
            if (im.bit_depth != 8 && im.bit_depth != 16) return path + ": only 8 and 16 bits per sample are supported";
            if (body[10] || body[11]) return path + ": unknown compression / filter method";
            if (body[12]) return path + ": interlaced PNG files are not supported";
            if (!im.width || !im.height) return path + ": empty image";
            // (size_t)height * (rowbytes + 1) below must not wrap, and the callers keep extents in int
            if (im.width > (1u << 20) || im.height > (1u << 20)) return path + ": dimensions beyond 2^20 are not supported";
            have_ihdr = true;
        } else if (!memcmp(head + 4, "IDAT", 4)) {
            z.insert(z.end(), body.begin(), body.end());
        } else if (!memcmp(head + 4, "IEND", 4)) {
            done = true;
        } else if (!(head[4] & 0x20)) {
            return path + ": unknown critical chunk";   // bit 5 of the first type byte clear = critical
        }
    }
    if (!have_ihdr || z.empty()) return path + ": no image data";
    const size_t bpp = (size_t)im.channels * (im.bit_depth / 8), rowbytes = (size_t)im.width * bpp;
    std::vector<uint8_t> raw((size_t)im.height * (rowbytes + 1));
    uLongf n = (uLongf)raw.size();
    if (uncompress(raw.data(), &n, z.data(), (uLong)z.size()) != Z_OK || n != raw.size()) return path + ": corrupt image data";
    im.bytes.assign((size_t)im.height * rowbytes, 0);
    for (uint32_t y = 0; y < im.height; y++) {
        const uint8_t *src = raw.data() + (size_t)y * (rowbytes + 1);
        uint8_t *cur = im.bytes.data() + (size_t)y * rowbytes;
        const uint8_t *up = y ? cur - rowbytes : nullptr;
        const int ft = src[0];
        if (ft > 4) return path + ": unknown scanline filter";
        for (size_t i = 0; i < rowbytes; i++) {
            const int a = i >= bpp ? cur[i - bpp] : 0, b = up ? up[i] : 0, c = (up && i >= bpp) ? up[i - bpp] : 0;
            int pred = 0;
            if (ft == 1) pred = a;
            else if (ft == 2) pred = b;
            else if (ft == 3) pred = (a + b) >> 1;
            else if (ft == 4) {
                const int p = a + b - c, pa = p > a ? p - a : a - p, pb = p > b ? p - b : b - p, pc = p > c ? p - c : c - p;
                pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
            }
            cur[i] = (uint8_t)(src[1 + i] + pred);
        }
    }
    return "";
}

inline std::string write(const std::string &path, const Image &im) {
    const int color_type = im.channels == 1 ? 0 : im.channels == 2 ? 4 : im.channels == 3 ? 2 : im.channels == 4 ? 6 : -1;
    if (color_type < 0 || (im.bit_depth != 8 && im.bit_depth != 16)) return path + ": PNG needs 1-4 channels of 8 or 16 bits";
    const size_t rowbytes = (size_t)im.width * im.channels * (im.bit_depth / 8);
    if (im.bytes.size() != rowbytes * im.height) return path + ": sample array does not match the image shape";
    std::vector<uint8_t> raw((size_t)im.height * (rowbytes + 1));
    for (uint32_t y = 0; y < im.height; y++) {
        raw[(size_t)y * (rowbytes + 1)] = 0;   // filter None
        memcpy(raw.data() + (size_t)y * (rowbytes + 1) + 1, im.bytes.data() + (size_t)y * rowbytes, rowbytes);
    }
    uLongf zn = compressBound((uLong)raw.size());
    std::vector<uint8_t> z(zn);
    if (compress2(z.data(), &zn, raw.data(), (uLong)raw.size(), 6) != Z_OK) return path + ": compression failed";
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) return "cannot write " + path;
    auto chunk = [&](const char *type, const uint8_t *data, uint32_t len) {
        uint8_t head[8], crc[4];
        put_be32(head, len);
        memcpy(head + 4, type, 4);
        uint32_t c = (uint32_t)crc32(0L, head + 4, 4);
        if (len) c = (uint32_t)crc32(c, data, len);
        put_be32(crc, c);
        fwrite(head, 1, 8, f);
        if (len) fwrite(data, 1, len, f);
        fwrite(crc, 1, 4, f);
    };
    static const uint8_t sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
    fwrite(sig, 1, 8, f);
    uint8_t ihdr[13];
    put_be32(ihdr, im.width), put_be32(ihdr + 4, im.height);
    ihdr[8] = (uint8_t)im.bit_depth, ihdr[9] = (uint8_t)color_type, ihdr[10] = ihdr[11] = ihdr[12] = 0;
    chunk("IHDR", ihdr, 13);
    chunk("IDAT", z.data(), (uint32_t)zn);
    chunk("IEND", nullptr, 0);
    const bool ok = !ferror(f);
    fclose(f);
    return ok ? "" : "write error on " + path;
}

}  // namespace hlmi_png
