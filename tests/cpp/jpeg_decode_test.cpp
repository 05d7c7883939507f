// jpeg_decode_test.cpp — test harness for halide_amd/tools/hlmi_jpeg.h: decodes argv[1], prints "width height channels" and
// writes the samples (row-major, channels interleaved) to argv[2].  Exit code 1 + the reason on stderr for a file it refuses.
#include "hlmi_jpeg.h"

int main(int argc, char **argv) {
    if (argc != 3) return 2;
    hlmi_jpeg::Image im;
    const std::string err = hlmi_jpeg::read(argv[1], im);
    if (!err.empty()) {
        fprintf(stderr, "%s\n", err.c_str());
        return 1;
    }
    printf("%u %u %d\n", im.width, im.height, im.channels);
    FILE *f = fopen(argv[2], "wb");
    if (!f) return 2;
    fwrite(im.bytes.data(), 1, im.bytes.size(), f);
    fclose(f);
    return 0;
}
