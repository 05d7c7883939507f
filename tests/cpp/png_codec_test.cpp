// png_codec_test <in.png> <out.png> — reads a PNG with the runner's codec (halide_amd/tools/hlmi_png.h), prints its header and
// writes the decoded samples back as a new PNG.  tests/test_rungen.py compares both files sample by sample.
#include <stdio.h>

#include "hlmi_png.h"

int main(int argc, char **argv) {
    if (argc != 3) return 2;
    hlmi_png::Image im;
    std::string err = hlmi_png::read(argv[1], im);
    if (!err.empty()) {
        fprintf(stderr, "%s\n", err.c_str());
        return 1;
    }
    printf("%u %u %d %d\n", im.width, im.height, im.channels, im.bit_depth);
    err = hlmi_png::write(argv[2], im);
    if (!err.empty()) {
        fprintf(stderr, "%s\n", err.c_str());
        return 1;
    }
    return 0;
}
