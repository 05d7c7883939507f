// hlmi_rungen — command-line runner for the pipelines of libhlmi.so, with the command line and the output format of
// the reference's RunGen (/root/reference/tools/RunGenMain.cpp:41-190 usage text, tools/RunGen.h:1212-1300), so that
// the commands its build files issue for `benchmark_apps` keep working:
//
//     hlmi_rungen --name=local_laplacian --estimate_all --benchmarks=all --parsable_output
//     local_laplacian.rungen input=random:0:[3840,2160,3] levels=8 alpha=0.142857 beta=1 output=out.ppm
//
// (the reference links one RunGen binary per generator; here the pipeline is `--name=` or the basename of argv[0]
// up to the first '.').  It drives a pipeline ONLY through what the reference's RunGen uses: `<name>_argv`
// (src/CodeGen_C.cpp:688-694), `<name>_metadata` (HalideRuntime.h:1937-1975), the bounds-query protocol
// (HalideRuntime.h:1851-1853: buffers with host == device == 0) and `buf->device_interface` for sync / copy_to_host.
//
// Image files: PNG (8 / 16 bit, gray / gray+alpha / RGB / RGBA, non-interlaced — the formats tools/halide_image_io.h:856-1040
// reads and writes; own codec over zlib in hlmi_png.h, this image has no libpng), binary PGM / PPM, and the reference's three raw
// array formats: .npy, .mat (MATLAB level 5) and .tmp (ImageStack), plus uncompressed TIFF (which the reference only writes) and baseline
// JPEG (hlmi_jpeg.h: reads to the samples libjpeg's defaults return, writes the file libjpeg writes at quality 99) — printed by --help.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <unistd.h>
#include <fstream>
#include <functional>
#include <iostream>
#include <map>
#include <random>
#include <sstream>
#include <string>

#include "hlmi_jpeg.h"
#include "hlmi_png.h"
#include <vector>

#include "hlmi_abi.h"
#include "hlmi_runtime.h"

namespace {

using Shape = std::vector<halide_dimension_t>;

[[noreturn]] void fail(const std::string &msg) {
    std::cerr << "hlmi_rungen: " << msg << "\n";
    exit(1);
}

std::vector<std::string> split(const std::string &s, char sep) {
    std::vector<std::string> out;
    std::stringstream ss(s);
    std::string item;
    while (std::getline(ss, item, sep)) out.push_back(item);
    return out;
}

bool parse_extents(const std::string &s, std::vector<int> *out) {  // "[a,b,c]"
    if (s.size() < 2 || s.front() != '[' || s.back() != ']') return false;
    out->clear();
    for (const auto &t : split(s.substr(1, s.size() - 2), ',')) {
        char *end = nullptr;
        long v = strtol(t.c_str(), &end, 10);
        if (end == t.c_str() || *end) return false;
        out->push_back((int)v);
    }
    return true;
}

size_t elem_bytes(halide_type_t t) { return (t.bits + 7) / 8; }

std::string type_name(halide_type_t t) {
    const char *base = t.code == halide_type_int ? "int" : t.code == halide_type_uint ? "uint" : t.code == halide_type_float ? "float" : "handle";
    return std::string(base) + std::to_string((int)t.bits);
}

// RunGen's warnings (tools/RunGenMain.cpp:297-300: "Warning: " + text on stderr), same wording
void warn(const std::string &msg) { std::cerr << "Warning: " << msg << "\n"; }

// one argument of the pipeline
struct Arg {
    const halide_filter_argument_t *md = nullptr;
    std::string spec;                 // raw command-line value ("" if none)
    halide_scalar_value_t scalar{};   // scalars
    halide_buffer_t buf{};            // buffers
    Shape dims;
    std::vector<uint8_t> storage;
    std::string out_path;
};

Shape dense_shape(const std::vector<int> &mins, const std::vector<int> &extents) {
    Shape s(extents.size());
    int stride = 1;
    for (size_t i = 0; i < extents.size(); i++) {
        s[i].min = mins.empty() ? 0 : mins[i];
        s[i].extent = extents[i];
        s[i].stride = stride;
        s[i].flags = 0;
        stride *= extents[i];
    }
    return s;
}

void allocate(Arg &a) {
    size_t n = 1;
    for (auto &d : a.dims) n *= (size_t)std::max(0, d.extent);
    a.storage.assign(n * elem_bytes(a.md->type) + 64, 0);
    a.buf = halide_buffer_t{};
    a.buf.host = a.storage.data();
    a.buf.type = a.md->type;
    a.buf.dimensions = (int)a.dims.size();
    a.buf.dim = a.dims.data();
}

size_t count(const Arg &a) {
    size_t n = 1;
    for (auto &d : a.dims) n *= (size_t)std::max(0, d.extent);
    return n;
}

template<typename F>
void for_each_element(Arg &a, F f) {  // f(index, coords)
    const size_t n = count(a);
    std::vector<int> c(a.dims.size(), 0);
    for (size_t i = 0; i < n; i++) {
        f(i, c);
        for (size_t d = 0; d < c.size(); d++) {
            if (++c[d] < a.dims[d].extent) break;
            c[d] = 0;
        }
    }
}

void store_value(Arg &a, size_t i, double v) {
    uint8_t *p = a.storage.data() + i * elem_bytes(a.md->type);
    const halide_type_t t = a.md->type;
    if (t.code == halide_type_float && t.bits == 32) { float x = (float)v; memcpy(p, &x, 4); }
    else if (t.code == halide_type_float && t.bits == 64) { memcpy(p, &v, 8); }
    else if (t.bits == 8) { uint8_t x = (uint8_t)(int64_t)v; memcpy(p, &x, 1); }
    else if (t.bits == 16) { uint16_t x = (uint16_t)(int64_t)v; memcpy(p, &x, 2); }
    else if (t.bits == 32) { uint32_t x = (uint32_t)(int64_t)v; memcpy(p, &x, 4); }
    else if (t.bits == 64) { uint64_t x = (uint64_t)(int64_t)v; memcpy(p, &x, 8); }
    else fail("unsupported element type " + type_name(t));
}

double load_value(const Arg &a, size_t i) {
    const uint8_t *p = a.storage.data() + i * elem_bytes(a.md->type);
    const halide_type_t t = a.md->type;
    if (t.code == halide_type_float && t.bits == 32) { float x; memcpy(&x, p, 4); return x; }
    if (t.code == halide_type_float && t.bits == 64) { double x; memcpy(&x, p, 8); return x; }
    if (t.bits == 8) return t.code == halide_type_int ? (double)*(const int8_t *)p : (double)*p;
    if (t.bits == 16) { uint16_t x; memcpy(&x, p, 2); return t.code == halide_type_int ? (double)(int16_t)x : (double)x; }
    if (t.bits == 32) { uint32_t x; memcpy(&x, p, 4); return t.code == halide_type_int ? (double)(int32_t)x : (double)x; }
    fail("unsupported element type " + type_name(t));
}

// random fill: mt19937_64 seeded as given; floats uniform in [0, 1), integers uniform over the whole range of the type
// (the convention the reference's RunGen documents, tools/RunGenMain.cpp:98-104)
void fill_random(Arg &a, uint64_t seed) {
    std::mt19937_64 rng(seed);
    const halide_type_t t = a.md->type;
    const size_t n = count(a);
    for (size_t i = 0; i < n; i++) {
        if (t.code == halide_type_float) store_value(a, i, std::uniform_real_distribution<double>(0.0, 1.0)(rng));
        else store_value(a, i, (double)(rng() & ((t.bits >= 64) ? ~0ull : ((1ull << t.bits) - 1))) - (t.code == halide_type_int ? std::ldexp(1.0, t.bits - 1) : 0.0));
    }
}

// ---- image files: binary PGM / PPM (maxval 255 or 65535, big-endian samples) and .npy (the reference's layout, little endian)
bool ends_with(const std::string &s, const std::string &e) { return s.size() >= e.size() && s.compare(s.size() - e.size(), e.size(), e) == 0; }

// conversion rule of the reference's image I/O when the file's sample type differs from the buffer's
// (tools/halide_image_io.h:79-240): integers rescale by the ratio of the maxima (u8 -> u16 is x257), integer <-> float
// maps the full range to [0, 1]
double convert_sample(double v, double file_max, halide_type_t t) {
    if (t.code == halide_type_float) return v / file_max;
    const double tmax = std::ldexp(1.0, t.bits) - 1;
    if (tmax == file_max) return v;
    return std::floor(v * (tmax / file_max) + 0.5);
}

// what RunGen says when a file does not have the argument's shape or type (tools/RunGen.h:433-477)
void note_loaded(const Arg &a, const std::vector<int> &file_extents, halide_type_t file_type) {
    const int have = (int)file_extents.size(), need = a.md->dimensions;
    if (have > need) {
        bool trivial = true;
        for (int d = need; d < have; d++) trivial = trivial && file_extents[d] == 1;
        if (!trivial) {
            warn(std::string("Image for Input \"") + a.md->name + "\" has " + std::to_string(have) + " dimensions, but only the first " +
                 std::to_string(need) + " were used; data loss may have occurred.");
        }
    } else if (have < need) {
        warn(std::string("Image for Input \"") + a.md->name + "\" has " + std::to_string(have) + " dimensions, but this argument requires at least " +
             std::to_string(need) + " dimensions: adding dummy dimensions of extent 1.");
    }
    if (file_type.code != a.md->type.code || file_type.bits != a.md->type.bits) {
        warn(std::string("Image loaded for argument \"") + a.md->name + "\" is type " + type_name(file_type) + " but this argument expects type " +
             type_name(a.md->type) + "; data loss may have occurred.");
    }
}
// ... and when an output is not stored as what it is (tools/RunGen.h:1116-1121)
void note_saved(const Arg &a, halide_type_t saved) {
    if (saved.code != a.md->type.code || saved.bits != a.md->type.bits) {
        warn(std::string("Image for argument \"") + a.md->name + "\" is of type " + type_name(a.md->type) + " but is being saved as type " +
             type_name(saved) + "; data loss may have occurred.");
    }
}
const halide_type_t kU8 = {halide_type_uint, 8, 1}, kU16 = {halide_type_uint, 16, 1};

void load_pnm(const std::string &path, Arg &a) {
    std::ifstream f(path, std::ios::binary);
    if (!f) fail("cannot open " + path);
    std::string magic;
    int w = 0, h = 0, maxv = 0;
    f >> magic >> w >> h >> maxv;
    f.get();
    const int ch = magic == "P6" ? 3 : magic == "P5" ? 1 : 0;
    if (!ch) fail(path + ": only binary PGM (P5) / PPM (P6) are supported");
    if (!f || w <= 0 || h <= 0 || maxv <= 0 || maxv > 65535) fail(path + ": damaged PGM / PPM header");
    const int bps = maxv > 255 ? 2 : 1;
    {   // the samples must be in the file before anything of their size is allocated
        const std::streampos here = f.tellg();
        f.seekg(0, std::ios::end);
        const double left = (double)(f.tellg() - here);
        f.seekg(here);
        if ((double)w * h * ch * bps > left) fail(path + ": file is truncated");
    }
    std::vector<uint8_t> raw((size_t)w * h * ch * bps);
    f.read((char *)raw.data(), raw.size());
    note_loaded(a, ch > 1 ? std::vector<int>{w, h, ch} : std::vector<int>{w, h}, bps == 2 ? kU16 : kU8);
    std::vector<int> ext = {w, h};
    if ((int)a.md->dimensions >= 3) ext.push_back(ch);
    while ((int)ext.size() < a.md->dimensions) ext.push_back(1);
    a.dims = dense_shape({}, ext);
    allocate(a);
    for_each_element(a, [&](size_t i, const std::vector<int> &c) {
        const int cc = c.size() >= 3 ? std::min(c[2], ch - 1) : 0;
        const size_t o = (((size_t)c[1] * w + c[0]) * ch + cc) * bps;
        const double v = bps == 2 ? (double)((raw[o] << 8) | raw[o + 1]) : (double)raw[o];
        store_value(a, i, convert_sample(v, maxv, a.md->type));
    });
}

// PNG: samples convert like the reference's image I/O (convert_sample); a 2-D buffer takes channel 0, a buffer with more
// channels than the file repeats the last one (RunGen's "adapt the image to the argument" rule, tools/RunGen.h:640-700)
void load_png(const std::string &path, Arg &a) {
    hlmi_png::Image im;
    const std::string err = hlmi_png::read(path, im);
    if (!err.empty()) fail(err);
    const int w = (int)im.width, h = (int)im.height, ch = im.channels;
    note_loaded(a, ch > 1 ? std::vector<int>{w, h, ch} : std::vector<int>{w, h}, im.bit_depth == 16 ? kU16 : kU8);
    std::vector<int> ext = {w, h};
    if ((int)a.md->dimensions >= 3) ext.push_back(ch);
    while ((int)ext.size() < a.md->dimensions) ext.push_back(1);
    a.dims = dense_shape({}, ext);
    allocate(a);
    const double file_max = im.bit_depth == 16 ? 65535.0 : 255.0;
    for_each_element(a, [&](size_t i, const std::vector<int> &c) {
        const int cc = c.size() >= 3 ? std::min(c[2], ch - 1) : 0;
        store_value(a, i, convert_sample((double)im.at((uint32_t)c[0], (uint32_t)c[1], cc), file_max, a.md->type));
    });
}

// JPEG (input only): hlmi_jpeg.h decodes baseline files to the samples libjpeg's default settings return — what the reference's
// load_jpg (tools/halide_image_io.h:1506-1548) hands on; conversion to the argument's type as for PNG
void load_jpg(const std::string &path, Arg &a) {
    hlmi_jpeg::Image im;
    const std::string err = hlmi_jpeg::read(path, im);
    if (!err.empty()) fail(err);
    const int w = (int)im.width, h = (int)im.height, ch = im.channels;
    note_loaded(a, ch > 1 ? std::vector<int>{w, h, ch} : std::vector<int>{w, h}, kU8);
    std::vector<int> ext = {w, h};
    if ((int)a.md->dimensions >= 3) ext.push_back(ch);
    while ((int)ext.size() < a.md->dimensions) ext.push_back(1);
    a.dims = dense_shape({}, ext);
    allocate(a);
    for_each_element(a, [&](size_t i, const std::vector<int> &c) {
        const int cc = c.size() >= 3 ? std::min(c[2], ch - 1) : 0;
        store_value(a, i, convert_sample((double)im.at((uint32_t)c[0], (uint32_t)c[1], cc), 255.0, a.md->type));
    });
}

// JPEG out: what the reference's save_jpg writes (tools/halide_image_io.h:1558-1610: libjpeg's defaults at quality 99) — the
// encoder in hlmi_jpeg.h reproduces libjpeg's file byte for byte.  8-bit samples; wider or float buffers are narrowed the way
// the reference narrows them before it saves (:143-186: u16 by (x + 128) * 255 + 255 >> 16, floats by lround(x * 255)).
void save_jpg(const std::string &path, const Arg &a) {
    const int w = a.dims.size() > 0 ? a.dims[0].extent : 1, h = a.dims.size() > 1 ? a.dims[1].extent : 1;
    const int ch = a.dims.size() > 2 ? a.dims[2].extent : 1;
    if (ch != 1 && ch != 3) fail(path + ": JPEG needs 1 or 3 channels, the buffer has " + std::to_string(ch));
    const halide_type_t t = a.md->type;
    note_saved(a, kU8);
    hlmi_jpeg::Image im;
    im.width = (uint32_t)w, im.height = (uint32_t)h, im.channels = ch;
    im.bytes.assign((size_t)w * h * ch, 0);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            for (int c = 0; c < ch; c++) {
                const size_t i = (size_t)x * a.dims[0].stride + (a.dims.size() > 1 ? (size_t)y * a.dims[1].stride : 0) +
                                 (a.dims.size() > 2 ? (size_t)c * a.dims[2].stride : 0);
                const double v = load_value(a, i);
                unsigned o;
                if (t.code == halide_type_float) o = (unsigned)std::lround(std::min(1.0, std::max(0.0, v)) * 255.0);
                else if (t.bits == 8) o = (unsigned)std::min(255.0, std::max(0.0, v));
                else if (t.bits == 16) o = (((unsigned)std::min(65535.0, std::max(0.0, v)) + 0x80u) * 255u + 255u) >> 16;
                else o = (unsigned)(((uint64_t)std::min(4294967295.0, std::max(0.0, v)) + 0x00808080u) / 0x01010101u);
                im.bytes[((size_t)y * w + x) * ch + c] = (uint8_t)(o > 255 ? 255 : o);
            }
    const std::string err = hlmi_jpeg::write(path, im, 99);
    if (!err.empty()) fail(err);
}

void save_png(const std::string &path, const Arg &a) {
    const int w = a.dims.size() > 0 ? a.dims[0].extent : 1, h = a.dims.size() > 1 ? a.dims[1].extent : 1;
    const int ch = a.dims.size() > 2 ? a.dims[2].extent : 1;
    if (ch < 1 || ch > 4) fail(path + ": PNG needs 1 to 4 channels, the buffer has " + std::to_string(ch));
    const halide_type_t t = a.md->type;
    hlmi_png::Image im;
    im.width = (uint32_t)w, im.height = (uint32_t)h, im.channels = ch;
    im.bit_depth = (t.code == halide_type_float || t.bits > 8) ? 16 : 8;
    note_saved(a, im.bit_depth == 16 ? kU16 : kU8);
    const double maxv = im.bit_depth == 16 ? 65535.0 : 255.0;
    im.bytes.assign((size_t)w * h * ch * (im.bit_depth / 8), 0);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            for (int c = 0; c < ch; c++) {
                const size_t i = (size_t)x * a.dims[0].stride + (a.dims.size() > 1 ? (size_t)y * a.dims[1].stride : 0) +
                                 (a.dims.size() > 2 ? (size_t)c * a.dims[2].stride : 0);
                double v = load_value(a, i);
                if (t.code == halide_type_float) v = std::floor(std::min(1.0, std::max(0.0, v)) * maxv + 0.5);
                else v = std::min(maxv, std::max(0.0, v));
                im.set((uint32_t)x, (uint32_t)y, c, (unsigned)v);
            }
    const std::string err = hlmi_png::write(path, im);
    if (!err.empty()) fail(err);
}

void save_pnm(const std::string &path, const Arg &a) {
    const int w = a.dims.size() > 0 ? a.dims[0].extent : 1, h = a.dims.size() > 1 ? a.dims[1].extent : 1;
    const int ch = a.dims.size() > 2 ? a.dims[2].extent : 1;
    if (ch != 1 && ch != 3) fail(path + ": PGM/PPM need 1 or 3 channels, the buffer has " + std::to_string(ch));
    const halide_type_t t = a.md->type;
    const bool wide = t.code == halide_type_float || t.bits > 8;
    note_saved(a, wide ? kU16 : kU8);
    const int maxv = wide ? 65535 : 255;
    std::ofstream f(path, std::ios::binary);
    f << (ch == 3 ? "P6" : "P5") << "\n" << w << " " << h << "\n" << maxv << "\n";
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            for (int c = 0; c < ch; c++) {
                const size_t i = (size_t)x * a.dims[0].stride + (a.dims.size() > 1 ? (size_t)y * a.dims[1].stride : 0) +
                                 (a.dims.size() > 2 ? (size_t)c * a.dims[2].stride : 0);
                double v = load_value(a, i);
                if (t.code == halide_type_float) v = std::floor(std::min(1.0, std::max(0.0, v)) * maxv + 0.5);
                else if (t.bits > 16) v = std::min(65535.0, std::max(0.0, v));
                const unsigned u = (unsigned)v;
                if (wide) f.put((char)(u >> 8));
                f.put((char)(u & 255));
            }
}

std::string npy_descr(halide_type_t t) {
    const char k = t.code == halide_type_float ? 'f' : t.code == halide_type_int ? 'i' : 'u';
    return std::string(t.bits == 8 ? "|" : "<") + k + std::to_string(t.bits / 8);
}

// The reference's convention (tools/halide_image_io.h:1313-1385 load_npy, :1414-1470 save_npy): the header's shape tuple lists the
// HALIDE extents, dimension 0 (x) first, 'fortran_order': False, and the payload is the dense buffer with x innermost — so
// numpy reads a W x H x C image as an array of shape (W, H, C) whose memory is really [c][y][x].  Kept bit for bit: files
// move between this runner and the reference's RunGen unchanged (tests/test_reference_consumers.py).
void save_npy(const std::string &path, const Arg &a) {
    std::string shape = "(";
    for (int d = 0; d < (int)a.dims.size(); d++) shape += std::to_string(a.dims[d].extent) + ",";
    shape += ")";
    std::string hdr = "{'descr': '" + npy_descr(a.md->type) + "', 'fortran_order': False, 'shape': " + shape + ", }";
    while ((10 + hdr.size() + 1) % 64) hdr += ' ';
    hdr += '\n';
    std::ofstream f(path, std::ios::binary);
    f.write("\x93NUMPY\x01\x00", 8);
    const uint16_t hl = (uint16_t)hdr.size();
    f.write((const char *)&hl, 2);
    f << hdr;
    f.write((const char *)a.storage.data(), count(a) * elem_bytes(a.md->type));
}

void load_npy(const std::string &path, Arg &a) {
    std::ifstream f(path, std::ios::binary);
    if (!f) fail("cannot open " + path);
    char magic[8] = {0};
    f.read(magic, 8);
    if (f.gcount() != 8 || memcmp(magic, "\x93NUMPY", 6) != 0 || magic[6] != 1) fail(path + ": not a version-1 .npy file");
    uint16_t hl = 0;
    f.read((char *)&hl, 2);
    std::string hdr(hl, ' ');
    f.read(&hdr[0], hl);
    if ((size_t)f.gcount() != hl || hdr.find("shape") == std::string::npos || hdr.find('(') == std::string::npos) fail(path + ": damaged .npy header");
    if (hdr.find(npy_descr(a.md->type)) == std::string::npos) fail(path + ": dtype must be " + npy_descr(a.md->type) + " for this argument");
    if (hdr.find("'fortran_order': False") == std::string::npos) fail(path + ": fortran_order arrays are not supported");
    const size_t p0 = hdr.find('(', hdr.find("shape")), p1 = hdr.find(')', p0);
    std::vector<int> np;
    for (const auto &t : split(hdr.substr(p0 + 1, p1 - p0 - 1), ','))
        if (t.find_first_of("0123456789") != std::string::npos) np.push_back(atoi(t.c_str()));
    if ((int)np.size() != a.md->dimensions) fail(path + ": expected " + std::to_string(a.md->dimensions) + " dimensions");
    double want = elem_bytes(a.md->type);
    for (int e : np) {
        if (e < 0) fail(path + ": negative extent");
        want *= e;
    }
    // (bytes_left is defined with the other raw formats below; the samples must be in the file before anything is allocated)
    {
        const std::streampos here = f.tellg();
        f.seekg(0, std::ios::end);
        const double left = (double)(f.tellg() - here);
        f.seekg(here);
        if (want > left) fail(path + ": file is truncated");
    }
    a.dims = dense_shape({}, np);   // dimension 0 first, as the reference writes them
    allocate(a);
    f.read((char *)a.storage.data(), count(a) * elem_bytes(a.md->type));
}

// ---- .tmp (ImageStack) and .mat (MATLAB level 5, one uncompressed numeric array): the two binary array formats of the
// reference's image I/O besides .npy (tools/halide_image_io.h:1632-1722 and :1760-2100; apps/interpolate and apps/camera_pipe feed
// .mat matrices).  Both hold a dense array with dimension 0 innermost; samples of another type than the argument's convert like
// every other file (convert_sample), extents of 1 are added or dropped to reach the argument's dimensionality.
struct RawArray {
    halide_type_t type;
    std::vector<int> extents;
    std::vector<uint8_t> bytes;
};
const halide_type_t kTmpTypes[10] = {{halide_type_float, 32, 1}, {halide_type_float, 64, 1}, {halide_type_uint, 8, 1}, {halide_type_int, 8, 1},
                                     {halide_type_uint, 16, 1}, {halide_type_int, 16, 1}, {halide_type_uint, 32, 1}, {halide_type_int, 32, 1},
                                     {halide_type_uint, 64, 1}, {halide_type_int, 64, 1}};
// MAT-file data type codes (miINT8 ...) <-> element types; class codes (mxDOUBLE_CLASS ...) for the array flags
struct MatType { uint32_t mi, mx; halide_type_t t; };
const MatType kMatTypes[10] = {{1, 8, {halide_type_int, 8, 1}},   {2, 9, {halide_type_uint, 8, 1}},   {3, 10, {halide_type_int, 16, 1}}, {4, 11, {halide_type_uint, 16, 1}},
                               {5, 12, {halide_type_int, 32, 1}}, {6, 13, {halide_type_uint, 32, 1}}, {7, 7, {halide_type_float, 32, 1}}, {9, 6, {halide_type_float, 64, 1}},
                               {12, 14, {halide_type_int, 64, 1}}, {13, 15, {halide_type_uint, 64, 1}}};
bool same_type(halide_type_t a, halide_type_t b) { return a.code == b.code && a.bits == b.bits; }

double raw_value(const RawArray &r, size_t i) {
    const uint8_t *p = r.bytes.data() + i * (r.type.bits / 8);
    const halide_type_t t = r.type;
    if (t.code == halide_type_float && t.bits == 32) { float x; memcpy(&x, p, 4); return x; }
    if (t.code == halide_type_float && t.bits == 64) { double x; memcpy(&x, p, 8); return x; }
    if (t.bits == 8) return t.code == halide_type_int ? (double)*(const int8_t *)p : (double)*p;
    if (t.bits == 16) { uint16_t x; memcpy(&x, p, 2); return t.code == halide_type_int ? (double)(int16_t)x : (double)x; }
    if (t.bits == 32) { uint32_t x; memcpy(&x, p, 4); return t.code == halide_type_int ? (double)(int32_t)x : (double)x; }
    uint64_t x; memcpy(&x, p, 8);
    return t.code == halide_type_int ? (double)(int64_t)x : (double)x;
}

void assign_raw(const std::string &path, Arg &a, const RawArray &r) {
    note_loaded(a, r.extents, r.type);
    std::vector<int> ext = r.extents;
    while ((int)ext.size() > a.md->dimensions && ext.back() == 1) ext.pop_back();
    while ((int)ext.size() < a.md->dimensions) ext.push_back(1);
    if ((int)ext.size() != a.md->dimensions) fail(path + ": the file has more non-trivial dimensions than the argument (" + std::to_string(a.md->dimensions) + ")");
    a.dims = dense_shape({}, ext);
    allocate(a);
    const size_t n = count(a);
    if (n * elem_bytes(r.type) > r.bytes.size()) fail(path + ": the file holds fewer samples than its header promises");
    if (same_type(r.type, a.md->type)) {
        memcpy(a.storage.data(), r.bytes.data(), n * elem_bytes(a.md->type));
        return;
    }
    const bool file_float = r.type.code == halide_type_float;
    const double file_max = file_float ? 1.0 : std::ldexp(1.0, r.type.bits) - 1;
    for (size_t i = 0; i < n; i++) {
        double v = raw_value(r, i);
        if (file_float && a.md->type.code != halide_type_float) {   // [0, 1] -> the integer range, rounded (the reference's rule)
            v = std::floor(std::min(1.0, std::max(0.0, v)) * (std::ldexp(1.0, a.md->type.bits) - 1) + 0.5);
            store_value(a, i, v);
        } else if (file_float) {
            store_value(a, i, v);
        } else {
            store_value(a, i, convert_sample(v, file_max, a.md->type));
        }
    }
}

// bytes between the read position and the end of the file: a header that promises more than that is rejected BEFORE anything
// of that size is allocated
uint64_t bytes_left(std::ifstream &f) {
    const std::streampos here = f.tellg();
    f.seekg(0, std::ios::end);
    const std::streampos end = f.tellg();
    f.seekg(here);
    return here < 0 || end < here ? 0 : (uint64_t)(end - here);
}

void read_exact(std::ifstream &f, void *dst, size_t n, const std::string &path) {
    f.read((char *)dst, (std::streamsize)n);
    if ((size_t)f.gcount() != n) fail(path + ": file is truncated");
}

void load_tmp(const std::string &path, Arg &a) {
    std::ifstream f(path, std::ios::binary);
    if (!f) fail("cannot open " + path);
    int32_t h[5];
    read_exact(f, h, sizeof h, path);
    if (h[0] <= 0 || h[1] <= 0 || h[2] <= 0 || h[3] <= 0 || h[4] < 0 || h[4] >= 10) fail(path + ": bad .tmp header");
    if ((double)h[0] * h[1] * h[2] * h[3] > 1e10) fail(path + ": .tmp extents are implausible");
    RawArray r;
    r.type = kTmpTypes[h[4]];
    r.extents = {h[0], h[1], h[2], h[3]};
    const uint64_t payload = (uint64_t)h[0] * h[1] * h[2] * h[3] * (r.type.bits / 8);
    if (payload > bytes_left(f)) fail(path + ": file is truncated");
    r.bytes.resize((size_t)payload);
    read_exact(f, r.bytes.data(), r.bytes.size(), path);
    assign_raw(path, a, r);
}

void save_tmp(const std::string &path, const Arg &a) {
    if (a.dims.size() > 4) fail(path + ": .tmp holds at most 4 dimensions");
    int32_t h[5] = {1, 1, 1, 1, -1};
    for (size_t d = 0; d < a.dims.size(); d++) h[d] = a.dims[d].extent;
    for (int i = 0; i < 10; i++) if (same_type(kTmpTypes[i], a.md->type)) h[4] = i;
    if (h[4] < 0) fail(path + ": element type " + type_name(a.md->type) + " cannot be stored in a .tmp file");
    std::ofstream f(path, std::ios::binary);
    f.write((const char *)h, sizeof h);
    f.write((const char *)a.storage.data(), (std::streamsize)(count(a) * elem_bytes(a.md->type)));
}

// ---- .tiff: uncompressed baseline TIFF, one strip per sample plane.  The reference only WRITES TIFF (tools/halide_image_io.h:2109-2113
// declines to read, :2230-2384 writes); the writer here produces the same kind of file -- little endian, planar when there is more
// than one sample per pixel, SampleFormat from the element type, a third dimension beyond 4 samples stored as tag 32997 (ImageDepth)
// -- and the reader accepts that family (either byte order, chunky or planar, any strip split), so a runner's output can be fed back.
struct TiffField { uint16_t tag, type; uint32_t count, value; };
void put16(std::vector<uint8_t> &o, uint16_t v) { o.push_back(v & 255), o.push_back(v >> 8); }
void put32(std::vector<uint8_t> &o, uint32_t v) { for (int i = 0; i < 4; i++) o.push_back((v >> (8 * i)) & 255); }

void save_tiff(const std::string &path, const Arg &a) {
    if (a.dims.size() > 4) fail(path + ": TIFF holds at most 4 dimensions");
    const halide_type_t t = a.md->type;
    if (t.code > halide_type_float || t.bits < 8) fail(path + ": element type " + type_name(t) + " cannot be stored in a TIFF file");
    uint32_t e[4] = {1, 1, 1, 1};
    for (size_t d = 0; d < a.dims.size(); d++) e[d] = (uint32_t)a.dims[d].extent;
    uint32_t depth = e[2], samples = e[3];
    if (samples <= 1 && depth < 5) samples = depth, depth = 1;     // x, y, c with few channels: the third dimension is the sample
    const uint64_t plane = (uint64_t)e[0] * e[1] * depth * (t.bits / 8);
    if (plane * samples >> 31) fail(path + ": too large for a TIFF file");
    const uint16_t sample_format = t.code == halide_type_uint ? 1 : t.code == halide_type_int ? 2 : 3;
    // layout: header (8) | IFD (2 + 12 n + 4) | two rationals | strip offsets | strip byte counts | samples
    const uint32_t n_fields = 15, ifd = 8, rationals = ifd + 2 + 12 * n_fields + 4, arrays = rationals + 16;
    const uint32_t data = arrays + (samples > 1 ? 8 * samples : 0);
    const TiffField fields[n_fields] = {
        {256, 4, 1, e[0]}, {257, 4, 1, e[1]}, {258, 3, 1, (uint32_t)t.bits}, {259, 3, 1, 1},
        {262, 3, 1, samples >= 3 ? 2u : 1u},                                       // RGB, or black is zero
        {273, 4, samples, samples > 1 ? arrays : data}, {277, 3, 1, samples}, {278, 4, 1, e[1]},
        {279, 4, samples, samples > 1 ? arrays + 4 * samples : (uint32_t)plane},
        {282, 5, 1, rationals}, {283, 5, 1, rationals + 8}, {284, 3, 1, samples > 1 ? 2u : 1u}, {296, 3, 1, 1},
        {339, 3, 1, sample_format}, {32997, 4, 1, depth}};
    std::vector<uint8_t> o;
    o.push_back('I'), o.push_back('I'), put16(o, 42), put32(o, ifd);
    put16(o, n_fields);
    for (const TiffField &f : fields) put16(o, f.tag), put16(o, f.type), put32(o, f.count), put32(o, f.value);
    put32(o, 0);                                                                   // no further IFD
    for (int i = 0; i < 4; i++) put32(o, 1);                                       // resolution 1/1 twice
    if (samples > 1) {
        for (uint32_t c = 0; c < samples; c++) put32(o, data + (uint32_t)(c * plane));
        for (uint32_t c = 0; c < samples; c++) put32(o, (uint32_t)plane);
    }
    std::ofstream f(path, std::ios::binary);
    f.write((const char *)o.data(), (std::streamsize)o.size());
    f.write((const char *)a.storage.data(), (std::streamsize)(plane * samples));   // dense, dimension 0 innermost: already planar
}

void load_tiff(const std::string &path, Arg &a) {
    std::ifstream f(path, std::ios::binary);
    if (!f) fail("cannot open " + path);
    std::vector<uint8_t> b((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    if (b.size() < 8 || (b[0] != 'I' && b[0] != 'M') || b[1] != b[0]) fail(path + ": not a TIFF file");
    const bool be = b[0] == 'M';
    auto need = [&](uint64_t off, uint64_t n) { if (off + n > b.size()) fail(path + ": file is truncated"); };
    auto rd16 = [&](uint64_t off) { need(off, 2); return (uint32_t)(be ? b[off] << 8 | b[off + 1] : b[off + 1] << 8 | b[off]); };
    auto rd32 = [&](uint64_t off) { need(off, 4); return be ? (uint32_t)b[off] << 24 | b[off + 1] << 16 | b[off + 2] << 8 | b[off + 3]
                                                             : (uint32_t)b[off + 3] << 24 | b[off + 2] << 16 | b[off + 1] << 8 | b[off]; };
    if (rd16(2) != 42) fail(path + ": not a TIFF file (BigTIFF is not supported)");
    const uint64_t ifd = rd32(4);
    const uint32_t n = rd16(ifd);
    uint32_t width = 0, height = 0, bits = 1, compression = 1, samples = 1, rows_per_strip = 0xffffffffu, planar = 1, format = 1, depth = 1;
    std::vector<uint32_t> offsets, counts;
    // a field's values: inline in the entry when they fit in 4 bytes, else at the offset the entry holds
    auto values = [&](uint64_t entry) {
        const uint32_t type = rd16(entry + 2), cnt = rd32(entry + 4);
        const uint32_t size = type == 3 ? 2 : type == 4 ? 4 : type == 1 ? 1 : 0;
        if (!size || cnt > (1u << 24)) fail(path + ": unsupported TIFF field type");
        const uint64_t at = (uint64_t)size * cnt <= 4 ? entry + 8 : rd32(entry + 8);
        std::vector<uint32_t> v(cnt);
        for (uint32_t i = 0; i < cnt; i++) v[i] = size == 2 ? rd16(at + 2 * i) : size == 4 ? rd32(at + 4 * i) : (need(at + i, 1), b[at + i]);
        return v;
    };
    for (uint32_t i = 0; i < n; i++) {
        const uint64_t entry = ifd + 2 + 12 * (uint64_t)i;
        const uint32_t tag = rd16(entry);
        if (tag == 282 || tag == 283) continue;                                    // rationals: not needed
        auto first = [&]() { auto v = values(entry); if (v.empty()) fail(path + ": empty TIFF field"); return v[0]; };
        switch (tag) {
        case 256: width = first(); break;
        case 257: height = first(); break;
        case 258: { auto v = values(entry); bits = v.at(0); for (uint32_t x : v) if (x != bits) fail(path + ": mixed sample sizes"); break; }
        case 259: compression = first(); break;
        case 273: offsets = values(entry); break;
        case 277: samples = first(); break;
        case 278: rows_per_strip = first(); break;
        case 279: counts = values(entry); break;
        case 284: planar = first(); break;
        case 339: format = first(); break;
        case 32997: depth = first(); break;
        default: break;
        }
    }
    if (compression != 1) fail(path + ": compressed TIFF files are not supported");
    if (!width || !height || !samples || !depth || width > (1u << 20) || height > (1u << 20) || depth > (1u << 20) || samples > 65535) fail(path + ": bad TIFF dimensions");
    if ((bits != 8 && bits != 16 && bits != 32 && bits != 64) || format < 1 || format > 3 || (format == 3 && bits < 32)) fail(path + ": unsupported TIFF sample type");
    if (offsets.empty() || offsets.size() != counts.size()) fail(path + ": bad TIFF strip tables");
    RawArray r;
    r.type = {format == 1 ? halide_type_uint : format == 2 ? halide_type_int : halide_type_float, (uint8_t)bits, 1};
    const size_t eb = bits / 8;
    // width, height, depth <= 2^20, samples < 2^16, eb <= 8: the product can pass 2^64 — compared in double before it is formed
    if ((double)width * height * depth * samples * (double)eb > (double)b.size()) fail(path + ": the strips hold fewer samples than the image");   // uncompressed: the samples are in the file
    const uint64_t total = (uint64_t)width * height * depth * samples * eb;
    std::vector<uint8_t> flat;                                                     // the strips, concatenated in file order
    flat.reserve(total);
    for (size_t s = 0; s < offsets.size(); s++) {
        need(offsets[s], counts[s]);
        flat.insert(flat.end(), b.begin() + offsets[s], b.begin() + offsets[s] + counts[s]);
    }
    if (flat.size() < total) fail(path + ": the strips hold fewer samples than the image");
    (void)rows_per_strip;
    if (be && eb > 1)
        for (uint64_t i = 0; i < total; i += eb) std::reverse(flat.begin() + i, flat.begin() + i + eb);
    r.bytes.resize(total);
    const uint64_t px = (uint64_t)width * height * depth;
    if (planar == 2 || samples == 1) {
        memcpy(r.bytes.data(), flat.data(), total);
    } else {                                                                       // chunky: sample index innermost in the file
        for (uint64_t p = 0; p < px; p++)
            for (uint32_t c = 0; c < samples; c++) memcpy(&r.bytes[(c * px + p) * eb], &flat[(p * samples + c) * eb], eb);
    }
    r.extents = {(int)width, (int)height, (int)depth, (int)samples};
    if (depth == 1) r.extents = {(int)width, (int)height, (int)samples};
    assign_raw(path, a, r);
}

void load_mat(const std::string &path, Arg &a) {
    std::ifstream f(path, std::ios::binary);
    if (!f) fail("cannot open " + path);
    uint8_t text[128];
    uint32_t w[4];
    read_exact(f, text, sizeof text, path);
    read_exact(f, w, 8, path);                    // data element: type, byte count
    if (w[0] != 14) fail(path + ": the first data element is not a numeric array (compressed .mat files are not supported)");
    read_exact(f, w, 16, path);                   // array flags sub-element
    if (w[0] != 6 || w[1] != 8) fail(path + ": bad array-flags element");
    read_exact(f, w, 8, path);                    // dimensions sub-element
    if (w[0] != 5 || w[1] % 4 != 0 || w[1] / 4 < 1 || w[1] / 4 > 16) fail(path + ": bad dimensions element");
    const int nd = (int)(w[1] / 4);
    RawArray r;
    r.extents.resize(nd);
    read_exact(f, r.extents.data(), 4 * (size_t)nd, path);
    if (nd & 1) read_exact(f, w, 4, path);        // sub-elements are padded to 8 bytes
    double total = 1;
    for (int e : r.extents) {
        if (e <= 0) fail(path + ": bad extent");
        total *= e;
    }
    if (total > 1e10) fail(path + ": extents are implausible");
    read_exact(f, w, 8, path);                    // array name: either packed into these 8 bytes (small element) or a sub-element
    if ((w[0] >> 16) == 0) {
        if (w[0] != 1) fail(path + ": bad array-name element");
        std::vector<uint8_t> name(((size_t)w[1] + 7) / 8 * 8);
        if (name.size() > (1u << 20)) fail(path + ": bad array-name length");
        read_exact(f, name.data(), name.size(), path);
    }
    read_exact(f, w, 8, path);                    // the real part
    bool known = false;
    for (const MatType &m : kMatTypes) if (m.mi == w[0]) r.type = m.t, known = true;
    if (!known) fail(path + ": unsupported sample type " + std::to_string(w[0]));
    if (total * (r.type.bits / 8) > (double)bytes_left(f)) fail(path + ": file is truncated");
    r.bytes.resize((size_t)total * (r.type.bits / 8));
    read_exact(f, r.bytes.data(), r.bytes.size(), path);
    assign_raw(path, a, r);
}

void save_mat(const std::string &path, const Arg &a) {
    const MatType *mt = nullptr;
    for (const MatType &m : kMatTypes) if (same_type(m.t, a.md->type)) mt = &m;
    if (!mt) fail(path + ": element type " + type_name(a.md->type) + " cannot be stored in a .mat file");
    // variable name = the file's stem, made a C identifier
    std::string name = path.substr(0, path.rfind('.'));
    if (name.rfind('/') != std::string::npos) name = name.substr(name.rfind('/') + 1);
    if (name.empty() || !isalpha((unsigned char)name[0])) name = "v" + name;
    for (char &c : name) if (!isalnum((unsigned char)c)) c = '_';
    const uint32_t name_len = (uint32_t)name.size();
    while (name.size() & 7) name += '\0';
    const uint64_t payload = (uint64_t)count(a) * elem_bytes(a.md->type);
    if (payload >> 32) fail(path + ": too large for a .mat file");
    const uint32_t pad = (uint32_t)(7 - ((payload - 1) & 7));
    const int nd_file = std::max(2, (int)a.dims.size()), nd_padded = nd_file + (nd_file & 1);
    char text[128];
    memset(text, ' ', sizeof text);
    const char *banner = "MATLAB 5.0 MAT-file, produced by hlmi_rungen";
    memcpy(text, banner, strlen(banner));
    text[124] = 0, text[125] = 1, text[126] = 'I', text[127] = 'M';   // version 0x0100, little endian
    std::ofstream f(path, std::ios::binary);
    f.write(text, sizeof text);
    const uint32_t head[2] = {14, 40u + 4u * (uint32_t)nd_padded + (uint32_t)name.size() + (uint32_t)payload + pad};
    const uint32_t flags[4] = {6, 8, mt->mx, 1};
    const uint32_t shape_h[2] = {5, 4u * (uint32_t)a.dims.size()};   // as the reference writes it: the argument's own dimension count
    std::vector<int32_t> ext;
    for (auto &d : a.dims) ext.push_back(d.extent);
    while ((int)ext.size() < nd_file) ext.push_back(1);
    while ((int)ext.size() < nd_padded) ext.push_back(0);
    const uint32_t name_h[2] = {1, name_len}, data_h[2] = {mt->mi, (uint32_t)payload};
    f.write((const char *)head, 8), f.write((const char *)flags, 16), f.write((const char *)shape_h, 8);
    f.write((const char *)ext.data(), (std::streamsize)(4 * ext.size()));
    f.write((const char *)name_h, 8), f.write(name.data(), (std::streamsize)name.size()), f.write((const char *)data_h, 8);
    f.write((const char *)a.storage.data(), (std::streamsize)payload);
    const uint64_t zero = 0;
    f.write((const char *)&zero, pad);
}

// ---- benchmark: protocol of tools/halide_benchmark.h:165-241 (>= 3 samples, iterations per sample grown until a
// sample set lasts min_time, more samples until best and third best agree within 3 % or max_time is spent)
struct BenchResult { double wall_time; uint64_t samples, iterations; double accuracy; };

double time_iters(uint64_t iters, const std::function<void()> &op) {
    const auto t0 = std::chrono::high_resolution_clock::now();
    for (uint64_t i = 0; i < iters; i++) op();
    const auto t1 = std::chrono::high_resolution_clock::now();
    return std::chrono::duration<double>(t1 - t0).count() / (double)iters;
}

BenchResult run_benchmark(const std::function<void()> &op, double min_time_cfg) {
    const double min_time = std::max(10e-6, min_time_cfg), max_time = std::max(min_time, 4 * min_time_cfg), accuracy = 1.03;
    BenchResult r{0, 0, 0, 0};
    double times[4] = {0, 0, 0, 0}, total = 0;
    uint64_t iters = 1;
    for (;;) {
        r.samples = r.iterations = 0, total = 0;
        for (int i = 0; i < 3; i++) {
            times[i] = time_iters(iters, op);
            r.samples++, r.iterations += iters, total += times[i] * iters;
        }
        std::sort(times, times + 3);
        if (times[0] < 1e-9) { iters *= 2; }
        else {
            const double f = times[0] * 3;
            if (f * iters >= min_time) break;
            iters = (uint64_t)std::llround(std::max(min_time / f, iters * 2.0));
        }
        if (iters >= 1000000) { iters = 1000000; break; }
    }
    while ((times[0] * accuracy < times[2] || total < min_time) && total < max_time) {
        times[3] = time_iters(iters, op);
        r.samples++, r.iterations += iters, total += times[3] * iters;
        std::sort(times, times + 4);
    }
    r.wall_time = times[0], r.accuracy = times[2] / times[0] - 1.0;
    return r;
}

void usage() {
    std::cout <<
        "Usage: hlmi_rungen --name=PIPELINE argument=value [argument=value ...] [flags]\n"
        "   or: PIPELINE.rungen argument=value ... (pipeline = basename of argv[0] up to the first '.')\n\n"
        "Arguments follow the reference's RunGen (tools/RunGenMain.cpp): scalars as literals or `default` / `estimate`;\n"
        "buffers as a file (.png .jpg .pgm .ppm .npy .mat .tmp .tiff) or a pseudo-file: zero:[e0,e1,..]  constant:V:[..]  identity:[..]\n"
        "random:SEED:[..]; `auto` or `estimate` may stand for the extents.\n\n"
        "Flags: --help --describe --output_extents=[..]|estimate --benchmarks=all --benchmark_min_time=SEC\n"
        "       --parsable_output --estimate_all --default_input_buffers[=V] --default_input_scalars[=V]\n"
        "       --success --verbose --quiet\n\n"
        "PNG: 8 / 16 bit, gray / gray+alpha / RGB / RGBA, non-interlaced.  JPG: baseline (sequential, Huffman, 8 bit, gray or YCbCr); written as libjpeg writes it at quality 99.\n";
}

}  // namespace

int main(int argc, char **argv) {
    std::string name, default_bufs, default_scalars, output_extents;
    std::map<std::string, std::string> given;
    bool describe = false, benchmarks = false, parsable = false, success = false, verbose = false, track_memory = false;
    double min_time = 0.1;
    {
        std::string base = argv[0];
        base = base.substr(base.find_last_of('/') + 1);
        if (base.find('.') != std::string::npos && base.rfind("hlmi_rungen", 0) != 0) name = base.substr(0, base.find('.'));
    }
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        auto val = [&](const std::string &flag) { return a.size() > flag.size() + 1 ? a.substr(flag.size() + 1) : std::string(); };
        // boolean flags as RunGen takes them (tools/RunGenMain.cpp:421-500): --flag, --flag=true, --flag=false
        auto bool_flag = [&](const std::string &flag, bool *dst) {
            if (a != flag && a.rfind(flag + "=", 0) != 0) return false;
            const std::string v = val(flag);
            if (v.empty() || v == "true") *dst = true;
            else if (v == "false") *dst = false;
            else fail("Invalid value for flag: " + flag.substr(2));
            return true;
        };
        bool ignored = false;
        if (a == "--help") { usage(); return 0; }
        else if (a.rfind("--name=", 0) == 0) name = val("--name");
        else if (bool_flag("--describe", &describe) || bool_flag("--parsable_output", &parsable) || bool_flag("--success", &success) ||
                 bool_flag("--verbose", &verbose) || bool_flag("--quiet", &ignored) || bool_flag("--track_memory", &track_memory) ||
                 bool_flag("--skip_bad_environment", &ignored) ||   // what RunGen parses (tools/RunGenMain.cpp:494) ...
                 bool_flag("--skip_bad_environement", &ignored)) {}  // ... and what its usage text prints (:179)
        else if (a.rfind("--output_extents", 0) == 0) output_extents = val("--output_extents");
        else if (a.rfind("--benchmarks", 0) == 0) { if (val("--benchmarks") != "all") fail("--benchmarks only supports 'all'"); benchmarks = true; }
        else if (a.rfind("--benchmark_min_time", 0) == 0) min_time = atof(val("--benchmark_min_time").c_str());
        else if (a == "--estimate_all") { default_bufs = "estimate_then_auto", default_scalars = "estimate", output_extents = "estimate"; }
        else if (a.rfind("--default_input_buffers", 0) == 0) { default_bufs = val("--default_input_buffers"); if (default_bufs.empty()) default_bufs = "zero:auto"; }
        else if (a.rfind("--default_input_scalars", 0) == 0) { default_scalars = val("--default_input_scalars"); if (default_scalars.empty()) default_scalars = "estimate,default"; }
        else if (a.rfind("--", 0) == 0) fail("unknown flag " + a);
        else {
            const size_t eq = a.find('=');
            if (eq == std::string::npos) fail("expected argument=value, got " + a);
            given[a.substr(0, eq)] = a.substr(eq + 1);
        }
    }
    if (name.empty()) { usage(); fail("no pipeline: use --name=PIPELINE"); }
    using argv_fn = int (*)(void **);
    using md_fn = const halide_filter_metadata_t *(*)();
    // the library: $HLMI_LIB, else ../lib/libhlmi.so next to this binary, else the loader's search path
    void *lib = nullptr;
    if (const char *e = getenv("HLMI_LIB")) lib = dlopen(e, RTLD_NOW | RTLD_GLOBAL);
    if (!lib) {
        char self[4096];
        const ssize_t n = readlink("/proc/self/exe", self, sizeof self - 1);
        if (n > 0) {
            self[n] = 0;
            std::string dir = self;
            dir = dir.substr(0, dir.find_last_of('/'));
            lib = dlopen((dir + "/../lib/libhlmi.so").c_str(), RTLD_NOW | RTLD_GLOBAL);
        }
    }
    if (!lib) lib = dlopen("libhlmi.so", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) fail(std::string("cannot load libhlmi.so: ") + dlerror());
    auto f_argv = (argv_fn)dlsym(lib, (name + "_argv").c_str());
    auto f_md = (md_fn)dlsym(lib, (name + "_metadata").c_str());
    if (!f_argv || !f_md) fail("libhlmi.so exports no pipeline named '" + name + "'");
    const halide_filter_metadata_t *md = f_md();

    std::vector<Arg> args(md->num_arguments);
    for (int i = 0; i < md->num_arguments; i++) {
        args[i].md = &md->arguments[i];
        auto it = given.find(md->arguments[i].name);
        if (it != given.end()) { args[i].spec = it->second; given.erase(it); }
    }
    if (!given.empty()) fail("unknown argument '" + given.begin()->first + "' (see --describe)");
    if (describe) {
        std::cout << "Filter name: \"" << md->name << "\"\n";
        for (auto &a : args) {
            const char *kind = a.md->kind == halide_argument_kind_input_scalar ? "Input" : a.md->kind == halide_argument_kind_input_buffer ? "Input" : "Output";
            std::cout << "  " << kind << " \"" << a.md->name << "\" is of type ";
            if (a.md->kind == halide_argument_kind_input_scalar) std::cout << type_name(a.md->type) << "\n";
            else std::cout << "Buffer<" << type_name(a.md->type) << "> with " << a.md->dimensions << " dimensions\n";
        }
        return 0;
    }

    auto estimate_extents = [&](const Arg &a, std::vector<int> *mins, std::vector<int> *ext) {
        if (!a.md->buffer_estimates) return false;
        mins->clear(), ext->clear();
        for (int d = 0; d < a.md->dimensions; d++) {
            const int64_t *m = a.md->buffer_estimates[2 * d], *e = a.md->buffer_estimates[2 * d + 1];
            if (!m || !e) return false;
            mins->push_back((int)*m), ext->push_back((int)*e);
        }
        return true;
    };

    // ---- scalars
    for (auto &a : args) {
        if (a.md->kind != halide_argument_kind_input_scalar) continue;
        std::string spec = a.spec.empty() ? default_scalars : a.spec;
        if (spec.empty()) fail(std::string("no value for scalar '") + a.md->name + "' (or use --default_input_scalars / --estimate_all)");
        bool done = false;
        for (const auto &tok : split(spec, ',')) {
            const halide_scalar_value_t *src = tok == "default" ? a.md->scalar_def : tok == "estimate" ? a.md->scalar_estimate : nullptr;
            if (tok == "default" || tok == "estimate") {
                if (src) { a.scalar = *src; done = true; break; }
                continue;
            }
            const halide_type_t t = a.md->type;
            if (t.code == halide_type_float && t.bits == 32) a.scalar.u.f32 = (float)atof(tok.c_str());
            else if (t.code == halide_type_float) a.scalar.u.f64 = atof(tok.c_str());
            else if (t.bits == 1 || tok == "true" || tok == "false") a.scalar.u.b = (tok == "true" || tok == "1");
            else a.scalar.u.i64 = atoll(tok.c_str());
            done = true;
            break;
        }
        if (!done) fail(std::string("scalar '") + a.md->name + "' has no " + spec + " value in the metadata");
    }

    // ---- output shapes requested on the command line
    std::vector<int> out_ext;
    const bool out_estimate = output_extents == "estimate";
    if (!output_extents.empty() && !out_estimate && !parse_extents(output_extents, &out_ext)) fail("bad --output_extents " + output_extents);

    // ---- input buffers (pass 1: everything that does not need a bounds query)
    std::vector<Arg *> auto_inputs;
    for (auto &a : args) {
        if (a.md->kind != halide_argument_kind_input_buffer) continue;
        std::string spec = a.spec.empty() ? default_bufs : a.spec;
        if (spec.empty()) fail(std::string("no value for buffer '") + a.md->name + "' (or use --default_input_buffers / --estimate_all)");
        if (spec == "estimate_then_auto") spec = a.md->buffer_estimates ? "random:0:estimate" : "random:0:auto";
        auto parts = split(spec, ':');
        const std::string kind = parts[0];
        if (kind == "zero" || kind == "constant" || kind == "identity" || kind == "random") {
            std::string ext_s = parts.back();
            std::vector<int> mins, ext;
            // estimate_then_auto (tools/RunGenMain.cpp:90-100): the estimate when the generator declared one, else the bounds query
            if (ext_s == "estimate_then_auto") ext_s = a.md->buffer_estimates && estimate_extents(a, &mins, &ext) ? "estimate" : "auto";
            if (ext_s == "auto") {
                a.spec = spec;
                auto_inputs.push_back(&a);
                continue;
            } else if (ext_s == "estimate") {
                if (!estimate_extents(a, &mins, &ext)) fail(std::string("no estimate for '") + a.md->name + "'");
            } else if (!parse_extents(ext_s, &ext)) {
                fail("bad extents in " + spec);
            }
            if ((int)ext.size() != a.md->dimensions) fail(std::string("'") + a.md->name + "' has " + std::to_string(a.md->dimensions) + " dimensions");
            a.dims = dense_shape(mins, ext);
            allocate(a);
            a.spec = spec;
        } else if (ends_with(spec, ".png")) {
            load_png(spec, a);
            a.spec.clear();
        } else if (ends_with(spec, ".jpg") || ends_with(spec, ".jpeg")) {
            load_jpg(spec, a);
            a.spec.clear();
        } else if (ends_with(spec, ".pgm") || ends_with(spec, ".ppm")) {
            load_pnm(spec, a);
            a.spec.clear();
        } else if (ends_with(spec, ".npy")) {
            load_npy(spec, a);
            a.spec.clear();
        } else if (ends_with(spec, ".mat")) {
            load_mat(spec, a);
            a.spec.clear();
        } else if (ends_with(spec, ".tiff") || ends_with(spec, ".tif")) {
            load_tiff(spec, a);
            a.spec.clear();
        } else if (ends_with(spec, ".tmp")) {
            load_tmp(spec, a);
            a.spec.clear();
        } else {
            fail("cannot read '" + spec + "': supported are .png .jpg .pgm .ppm .npy .mat .tmp .tiff and the pseudo-files of --help");
        }
    }
    // ---- outputs: shape from --output_extents, the estimates, or a bounds query constrained by the inputs
    for (auto &a : args) {
        if (a.md->kind != halide_argument_kind_output_buffer) continue;
        a.out_path = a.spec;
        std::vector<int> mins, ext;
        if (out_estimate) {
            if (!estimate_extents(a, &mins, &ext)) fail(std::string("no estimate for output '") + a.md->name + "'");
        } else if (!out_ext.empty()) {
            ext = out_ext;
            if ((int)ext.size() != a.md->dimensions) fail("--output_extents has the wrong number of dimensions");
        }
        if (ext.empty() && output_extents.empty()) {
            // no --output_extents: RunGen lets every output assume the shape of the first input buffer — first by argument name —
            // cut or extended to the output's dimension count, missing extents guessed as 1000 (x, y) or 4 (tools/RunGen.h:378-389,
            // :1077-1090); the bounds query below then has something to constrain
            const Arg *first = nullptr;
            for (auto &in : args)
                if (in.md->kind == halide_argument_kind_input_buffer && !in.dims.empty() && !in.storage.empty() &&
                    (!first || strcmp(in.md->name, first->md->name) < 0)) first = &in;
            if (first) {
                for (int d = 0; d < a.md->dimensions; d++) ext.push_back(d < (int)first->dims.size() ? first->dims[d].extent : (d < 2 ? 1000 : 4));
            }
        }
        if (!ext.empty()) {
            a.dims = dense_shape(mins, ext);
            if (out_estimate || !out_ext.empty()) allocate(a);   // an assumed shape still goes through the bounds query
        }
    }
    auto call = [&](std::vector<Arg> &as) {
        std::vector<void *> av;
        for (auto &a : as) av.push_back(a.md->kind == halide_argument_kind_input_scalar ? (void *)&a.scalar : (void *)&a.buf);
        return f_argv(av.data());
    };
    // bounds query (tools/RunGen.h:1212-1250): EVERY buffer goes in with host == device == 0 and the shape it has so far (zeros
    // where nothing is known yet) and comes back with the region the pipeline wants of it given the others.  Buffers without
    // a shape take what comes back; an input that was loaded but does not cover its region is re-allocated on the region and
    // the loaded samples copied in (adapt_input_buffer, tools/RunGen.h:774-817).
    {
        std::vector<Shape> q(args.size());
        std::vector<halide_buffer_t> qb(args.size());
        std::vector<void *> av;
        for (size_t i = 0; i < args.size(); i++) {
            Arg &a = args[i];
            if (a.md->kind == halide_argument_kind_input_scalar) { av.push_back(&a.scalar); continue; }
            q[i] = a.dims.empty() ? dense_shape({}, std::vector<int>(a.md->dimensions, 0)) : a.dims;
            qb[i] = halide_buffer_t{};
            qb[i].type = a.md->type, qb[i].dimensions = a.md->dimensions, qb[i].dim = q[i].data();
            av.push_back(&qb[i]);
        }
        if (int r = f_argv(av.data())) fail("bounds query failed with error " + std::to_string(r));
        for (size_t i = 0; i < args.size(); i++) {
            Arg &a = args[i];
            if (a.md->kind == halide_argument_kind_input_scalar) continue;
            if (a.storage.empty()) {                       // no shape of its own (auto input, output): the pipeline's proposal
                a.dims = q[i];
                allocate(a);
                continue;
            }
            if (a.md->kind != halide_argument_kind_input_buffer) continue;
            Shape want = a.dims;
            bool grow = false;
            for (size_t d = 0; d < want.size(); d++) {
                const int cur0 = a.dims[d].min, cur1 = cur0 + a.dims[d].extent - 1, c0 = q[i][d].min, c1 = c0 + q[i][d].extent - 1;
                if (q[i][d].extent > 0 && (c0 < cur0 || c1 > cur1)) want[d].min = c0, want[d].extent = q[i][d].extent, grow = true;
            }
            if (!grow) continue;
            Arg old = a;                                   // (its storage is copied with it)
            std::vector<int> mins, ext;
            for (auto &d : want) mins.push_back(d.min), ext.push_back(d.extent);
            a.dims = dense_shape(mins, ext);
            allocate(a);
            const size_t eb = elem_bytes(a.md->type);
            for_each_element(a, [&](size_t idx, const std::vector<int> &c) {
                size_t o = 0;
                for (size_t d = 0; d < c.size(); d++) {
                    const int rel = a.dims[d].min + c[d] - old.dims[d].min;
                    if (rel < 0 || rel >= old.dims[d].extent) return;
                    o += (size_t)rel * old.dims[d].stride;
                }
                memcpy(a.storage.data() + idx * eb, old.storage.data() + o * eb, eb);
            });
            if (verbose) std::cout << "Input " << a.md->name << ": grown to the region the bounds query asks for\n";
        }
    }
    // ---- fill the pseudo-file inputs
    for (auto &a : args) {
        if (a.md->kind != halide_argument_kind_input_buffer || a.spec.empty()) continue;
        auto parts = split(a.spec, ':');
        if (parts[0] == "constant") { const double v = atof(parts[1].c_str()); for (size_t i = 0, n = count(a); i < n; i++) store_value(a, i, v); }
        else if (parts[0] == "identity") for_each_element(a, [&](size_t i, const std::vector<int> &c) { store_value(a, i, (c.size() >= 2 && c[0] == c[1]) ? 1.0 : 0.0); });
        else if (parts[0] == "random") fill_random(a, strtoull(parts[1].c_str(), nullptr, 10));
        a.buf.flags |= 1;  // host_dirty: the pipeline uploads it
    }
    if (verbose) {
        for (auto &a : args) {
            if (a.md->kind == halide_argument_kind_input_scalar) continue;
            std::cout << "Argument " << a.md->name << ": [";
            for (auto &d : a.dims) std::cout << " (" << d.min << "," << d.extent << "," << d.stride << ")";
            std::cout << " ]\n";
        }
        std::cout.flush();   // (an error in the run aborts: what was said so far should still be seen)
    }

    // ---- run
    auto sync_outputs = [&]() {
        for (auto &a : args)
            if (a.md->kind == halide_argument_kind_output_buffer && a.buf.device_interface) a.buf.device_interface->device_sync(nullptr, &a.buf);
    };
    if (int r = call(args)) fail(std::string(md->name) + " returned error " + std::to_string(r));
    sync_outputs();
    double pixels = 0;
    for (auto &a : args) {
        if (a.md->kind != halide_argument_kind_output_buffer) continue;
        pixels += a.dims.size() >= 2 ? (double)a.dims[0].extent * a.dims[1].extent : a.dims.size() == 1 ? a.dims[0].extent : 1;
    }
    if (benchmarks) {
        const BenchResult r = run_benchmark([&]() { (void)call(args); sync_outputs(); }, min_time);
        const double mpix = pixels / (1024.0 * 1024.0);
        if (!parsable) {
            std::cout << "Benchmark for " << md->name << " produces best case of " << r.wall_time << " sec/iter (over " << r.samples
                      << " samples, " << r.iterations << " iterations, accuracy " << (r.accuracy * 100.0) << "%).\n"
                      << "Best output throughput is " << (mpix / r.wall_time) << " mpix/sec.\n";
        } else {
            std::cout << md->name << "  BEST_TIME_MSEC_PER_ITER  " << r.wall_time * 1000.0 << "\n"
                      << md->name << "  SAMPLES                  " << r.samples << "\n"
                      << md->name << "  ITERATIONS               " << r.iterations << "\n"
                      << md->name << "  TIMING_ACCURACY          " << r.accuracy << "\n"
                      << md->name << "  THROUGHPUT_MPIX_PER_SEC  " << (mpix / r.wall_time) << "\n"
                      << md->name << "  HALIDE_TARGET            " << md->target << "\n";
        }
    }
    if (track_memory) {
        // RunGen tracks what the pipeline takes through halide_malloc (tools/RunGenMain.cpp:198-260, :624-632) — host memory; device
        // allocations are not part of its figure either.  These pipelines allocate nothing on the host, the line says so in RunGen's words.
        if (benchmarks) warn("Using --track_memory with --benchmarks will produce inaccurate benchmark results.");
        std::cout << "Maximum Halide memory: 0 bytes for output of " << pixels / (1024.0 * 1024.0) << " mpix.\n";
    }
    // ---- save outputs
    for (auto &a : args) {
        if (a.md->kind != halide_argument_kind_output_buffer) continue;
        if (a.buf.device_interface && (a.buf.flags & 2)) {
            if (int r = a.buf.device_interface->copy_to_host(nullptr, &a.buf)) fail("copy_to_host failed with " + std::to_string(r));
        }
        if (a.out_path.empty()) continue;
        if (ends_with(a.out_path, ".npy")) save_npy(a.out_path, a);
        else if (ends_with(a.out_path, ".png")) save_png(a.out_path, a);
        else if (ends_with(a.out_path, ".jpg") || ends_with(a.out_path, ".jpeg")) save_jpg(a.out_path, a);
        else if (ends_with(a.out_path, ".pgm") || ends_with(a.out_path, ".ppm")) save_pnm(a.out_path, a);
        else if (ends_with(a.out_path, ".mat")) save_mat(a.out_path, a);
        else if (ends_with(a.out_path, ".tmp")) save_tmp(a.out_path, a);
        else if (ends_with(a.out_path, ".tiff") || ends_with(a.out_path, ".tif")) save_tiff(a.out_path, a);
        else fail("cannot write '" + a.out_path + "': supported are .png .jpg .pgm .ppm .npy .mat .tmp .tiff");
    }
    for (auto &a : args)
        if (a.md->kind != halide_argument_kind_input_scalar && a.buf.device_interface) a.buf.device_interface->device_free(nullptr, &a.buf);
    if (success) std::cout << "Success!\n";
    return 0;
}
