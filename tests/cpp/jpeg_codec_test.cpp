// jpeg_codec_test.cpp — test harness for halide_amd/tools/hlmi_jpeg.h.
//   jpeg_codec_test decode in.jpg out.bin             prints "width height channels", writes the samples (row-major, interleaved)
//   jpeg_codec_test encode in.bin w h c out.jpg q     writes the JPEG file of those samples at quality q
// Exit code 1 + the reason on stderr for a file it refuses.
#include "hlmi_jpeg.h"

#include <stdlib.h>

int main(int argc, char **argv) {
    if (argc == 4 && !strcmp(argv[1], "decode")) {
        hlmi_jpeg::Image im;
        const std::string err = hlmi_jpeg::read(argv[2], im);
        if (!err.empty()) {
            fprintf(stderr, "%s\n", err.c_str());
            return 1;
        }
        printf("%u %u %d\n", im.width, im.height, im.channels);
        FILE *f = fopen(argv[3], "wb");
        if (!f) return 2;
        fwrite(im.bytes.data(), 1, im.bytes.size(), f);
        fclose(f);
        return 0;
    }
    if (argc == 8 && !strcmp(argv[1], "encode")) {
        hlmi_jpeg::Image im;
        im.width = (uint32_t)atoi(argv[3]), im.height = (uint32_t)atoi(argv[4]), im.channels = atoi(argv[5]);
        im.bytes.resize((size_t)im.width * im.height * im.channels);
        FILE *f = fopen(argv[2], "rb");
        if (!f || fread(im.bytes.data(), 1, im.bytes.size(), f) != im.bytes.size()) return 2;
        fclose(f);
        const std::string err = hlmi_jpeg::write(argv[6], im, atoi(argv[7]));
        if (!err.empty()) {
            fprintf(stderr, "%s\n", err.c_str());
            return 1;
        }
        return 0;
    }
    return 2;
}
