/* png.h — TEST INFRASTRUCTURE ONLY.  The subset of the libpng API that the reference's tools/halide_image_io.h uses
 * (load_png :856-940, save_png :952-1040: create/destroy structs, init_io, read_info, get_*, read_row, set_IHDR,
 * write_info, write_row, write_end, setjmp error protocol), implemented over zlib.  This image has no libpng; with this
 * header on the include path the reference's drivers compile WITHOUT -DHALIDE_NO_PNG and read / write real PNG files
 * (non-interlaced, 8 / 16 bit, gray / gray+alpha / RGB / RGBA; all five scanline filters on input, filter 0 on output).
 * Own text, written against the PNG specification (ISO/IEC 15948), not against libpng's sources. */
#ifndef HLMI_TEST_PNG_SHIM_H
#define HLMI_TEST_PNG_SHIM_H

#include <setjmp.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <zlib.h>

#define PNG_LIBPNG_VER_STRING "hlmi-shim"
#define PNG_COLOR_TYPE_GRAY 0
#define PNG_COLOR_TYPE_RGB 2
#define PNG_COLOR_TYPE_PALETTE 3
#define PNG_COLOR_TYPE_GRAY_ALPHA 4
#define PNG_COLOR_TYPE_RGB_ALPHA 6
#define PNG_INTERLACE_NONE 0
#define PNG_COMPRESSION_TYPE_BASE 0
#define PNG_FILTER_TYPE_BASE 0

typedef unsigned char png_byte;
typedef png_byte *png_bytep;
typedef uint32_t png_uint_32;

struct png_struct_def {
    jmp_buf jb;
    FILE *f = nullptr;
    bool writing = false;
    uint32_t width = 0, height = 0;
    int bit_depth = 0, color_type = 0, channels = 0;
    size_t rowbytes = 0, next_row = 0;
    std::vector<uint8_t> data;      // reading: the inflated, still filtered scanlines; writing: filtered scanlines so far
    std::vector<uint8_t> prev;      // reading: the previous reconstructed row
};
struct png_info_def {
    int unused;
};
typedef png_struct_def png_struct;
typedef png_struct *png_structp;
typedef png_struct **png_structpp;
typedef png_info_def png_info;
typedef png_info *png_infop;
typedef png_info **png_infopp;

#define png_jmpbuf(png_ptr) ((png_ptr)->jb)

static inline void png_shim_fail(png_structp p) { longjmp(p->jb, 1); }

static inline int png_sig_cmp(const png_byte *sig, size_t start, size_t n) {
    static const png_byte want[8] = {137, 80, 78, 71, 13, 10, 26, 10};
    if (start + n > 8) n = 8 - start;
    return memcmp(sig + start, want + start, n);
}
static inline png_structp png_create_read_struct(const char *, void *, void *, void *) { return new png_struct(); }
static inline png_structp png_create_write_struct(const char *, void *, void *, void *) {
    png_structp p = new png_struct();
    p->writing = true;
    return p;
}
static inline png_infop png_create_info_struct(png_structp) { return new png_info(); }
static inline void png_init_io(png_structp p, FILE *f) { p->f = f; }
static inline void png_set_sig_bytes(png_structp, int) {}
static inline void png_destroy_read_struct(png_structpp p, png_infopp i, png_infopp) {
    if (p && *p) delete *p, *p = nullptr;
    if (i && *i) delete *i, *i = nullptr;
}
static inline void png_destroy_write_struct(png_structpp p, png_infopp i) { png_destroy_read_struct(p, i, nullptr); }

static inline uint32_t png_shim_be32(const uint8_t *b) { return ((uint32_t)b[0] << 24) | ((uint32_t)b[1] << 16) | ((uint32_t)b[2] << 8) | b[3]; }
static inline int png_shim_channels(int color_type) {
    switch (color_type) {
    case PNG_COLOR_TYPE_GRAY: return 1;
    case PNG_COLOR_TYPE_GRAY_ALPHA: return 2;
    case PNG_COLOR_TYPE_RGB: return 3;
    case PNG_COLOR_TYPE_RGB_ALPHA: return 4;
    default: return 0;
    }
}

/* reads every chunk after the signature: IHDR, the concatenated IDAT stream (inflated here), stops at IEND */
static inline void png_read_info(png_structp p, png_infop) {
    std::vector<uint8_t> z;
    bool have_ihdr = false;
    for (;;) {
        uint8_t head[8];
        if (fread(head, 1, 8, p->f) != 8) png_shim_fail(p);
        const uint32_t len = png_shim_be32(head);
        std::vector<uint8_t> body(len);
        if (len && fread(body.data(), 1, len, p->f) != len) png_shim_fail(p);
        uint8_t crc[4];
        if (fread(crc, 1, 4, p->f) != 4) png_shim_fail(p);
        uint32_t c = crc32(0L, head + 4, 4);
        if (len) c = crc32(c, body.data(), len);
        if (c != png_shim_be32(crc)) png_shim_fail(p);
        if (!memcmp(head + 4, "IHDR", 4)) {
            if (len != 13) png_shim_fail(p);
            p->width = png_shim_be32(body.data()), p->height = png_shim_be32(body.data() + 4);
            p->bit_depth = body[8], p->color_type = body[9];
            p->channels = png_shim_channels(p->color_type);
            if (!p->channels || (p->bit_depth != 8 && p->bit_depth != 16) || body[10] || body[11] || body[12] != PNG_INTERLACE_NONE) png_shim_fail(p);
            p->rowbytes = (size_t)p->width * p->channels * (p->bit_depth / 8);
            have_ihdr = true;
        } else if (!memcmp(head + 4, "IDAT", 4)) {
            z.insert(z.end(), body.begin(), body.end());
        } else if (!memcmp(head + 4, "IEND", 4)) {
            break;
        }
    }
    if (!have_ihdr) png_shim_fail(p);
    p->data.resize((size_t)p->height * (p->rowbytes + 1));
    uLongf n = (uLongf)p->data.size();
    if (uncompress(p->data.data(), &n, z.data(), (uLong)z.size()) != Z_OK || n != p->data.size()) png_shim_fail(p);
    p->prev.assign(p->rowbytes, 0);
    p->next_row = 0;
}
static inline png_uint_32 png_get_image_width(png_structp p, png_infop) { return p->width; }
static inline png_uint_32 png_get_image_height(png_structp p, png_infop) { return p->height; }
static inline png_byte png_get_channels(png_structp p, png_infop) { return (png_byte)p->channels; }
static inline png_byte png_get_bit_depth(png_structp p, png_infop) { return (png_byte)p->bit_depth; }
static inline size_t png_get_rowbytes(png_structp p, png_infop) { return p->rowbytes; }
static inline void png_read_update_info(png_structp, png_infop) {}

/* reconstructs the next scanline (filter types 0..4 of the PNG specification, clause 9) */
static inline void png_read_row(png_structp p, png_bytep row, png_bytep) {
    if (p->next_row >= p->height) png_shim_fail(p);
    const uint8_t *src = p->data.data() + p->next_row * (p->rowbytes + 1);
    const int ft = src[0];
    const size_t bpp = (size_t)p->channels * (p->bit_depth / 8);
    const uint8_t *up = p->prev.data();
    for (size_t i = 0; i < p->rowbytes; i++) {
        const int a = i >= bpp ? row[i - bpp] : 0, b = up[i], c = i >= bpp ? up[i - bpp] : 0;
        int pred;
        switch (ft) {
        case 0: pred = 0; break;
        case 1: pred = a; break;
        case 2: pred = b; break;
        case 3: pred = (a + b) >> 1; break;
        case 4: {
            const int pp = a + b - c, pa = abs(pp - a), pb = abs(pp - b), pc = abs(pp - c);
            pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
            break;
        }
        default: png_shim_fail(p); pred = 0;
        }
        row[i] = (uint8_t)(src[1 + i] + pred);
    }
    memcpy(p->prev.data(), row, p->rowbytes);
    p->next_row++;
}

static inline void png_set_IHDR(png_structp p, png_infop, png_uint_32 w, png_uint_32 h, int bit_depth, int color_type, int interlace, int, int) {
    p->width = w, p->height = h, p->bit_depth = bit_depth, p->color_type = color_type;
    p->channels = png_shim_channels(color_type);
    if (!p->channels || (bit_depth != 8 && bit_depth != 16) || interlace != PNG_INTERLACE_NONE) png_shim_fail(p);
    p->rowbytes = (size_t)w * p->channels * (bit_depth / 8);
}
static inline void png_shim_chunk(png_structp p, const char *type, const uint8_t *body, uint32_t len) {
    uint8_t head[8] = {(uint8_t)(len >> 24), (uint8_t)(len >> 16), (uint8_t)(len >> 8), (uint8_t)len, (uint8_t)type[0], (uint8_t)type[1], (uint8_t)type[2], (uint8_t)type[3]};
    uint32_t c = crc32(0L, head + 4, 4);
    if (len) c = crc32(c, body, len);
    const uint8_t crc[4] = {(uint8_t)(c >> 24), (uint8_t)(c >> 16), (uint8_t)(c >> 8), (uint8_t)c};
    if (fwrite(head, 1, 8, p->f) != 8 || (len && fwrite(body, 1, len, p->f) != len) || fwrite(crc, 1, 4, p->f) != 4) png_shim_fail(p);
}
static inline void png_write_info(png_structp p, png_infop) {
    static const uint8_t sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
    if (fwrite(sig, 1, 8, p->f) != 8) png_shim_fail(p);
    uint8_t ihdr[13] = {(uint8_t)(p->width >> 24), (uint8_t)(p->width >> 16), (uint8_t)(p->width >> 8), (uint8_t)p->width,
                        (uint8_t)(p->height >> 24), (uint8_t)(p->height >> 16), (uint8_t)(p->height >> 8), (uint8_t)p->height,
                        (uint8_t)p->bit_depth, (uint8_t)p->color_type, 0, 0, 0};
    png_shim_chunk(p, "IHDR", ihdr, 13);
    p->data.clear();
}
static inline void png_write_row(png_structp p, const png_byte *row) {
    p->data.push_back(0);   // filter type 0 (None)
    p->data.insert(p->data.end(), row, row + p->rowbytes);
}
static inline void png_write_end(png_structp p, png_infop) {
    uLongf n = compressBound((uLong)p->data.size());
    std::vector<uint8_t> z(n);
    if (compress2(z.data(), &n, p->data.data(), (uLong)p->data.size(), 6) != Z_OK) png_shim_fail(p);
    png_shim_chunk(p, "IDAT", z.data(), (uint32_t)n);
    png_shim_chunk(p, "IEND", nullptr, 0);
}
#endif
