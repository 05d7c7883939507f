// runner_fixture_lib.cpp — TEST INFRASTRUCTURE: two CPU "pipelines" behind the <name>_argv / <name>_metadata protocol, so that
// tests/test_rungen.py can take hlmi_rungen through load -> bounds query -> run -> save without a GPU (HLMI_LIB points the runner
// at this library instead of libhlmi.so).  Nothing of the product is in here.
//   fixture_copy   f32 [w, h] -> f32 [w, h], output = input
//   fixture_shift  u8  [w, h] -> u8  [w, h], output(x, y) = input(x + 1, y + 1): needs the input over the output's box grown by 2
#include "hlmi_abi.h"

namespace {
const halide_type_t f32 = {halide_type_float, 32, 1}, u8 = {halide_type_uint, 8, 1};
const halide_filter_argument_t copy_args[2] = {{"input", halide_argument_kind_input_buffer, 2, f32, nullptr, nullptr, nullptr, nullptr, nullptr},
                                               {"output", halide_argument_kind_output_buffer, 2, f32, nullptr, nullptr, nullptr, nullptr, nullptr}};
const halide_filter_argument_t shift_args[2] = {{"input", halide_argument_kind_input_buffer, 2, u8, nullptr, nullptr, nullptr, nullptr, nullptr},
                                                {"output", halide_argument_kind_output_buffer, 2, u8, nullptr, nullptr, nullptr, nullptr, nullptr}};
const halide_filter_metadata_t copy_md = {1, 2, copy_args, "host", "fixture_copy"}, shift_md = {1, 2, shift_args, "host", "fixture_shift"};
bool query(const halide_buffer_t *b) { return b->host == nullptr && b->device == 0; }
}  // namespace

extern "C" const halide_filter_metadata_t *fixture_copy_metadata() { return &copy_md; }
extern "C" int fixture_copy_argv(void **a) {
    halide_buffer_t *in = (halide_buffer_t *)a[0], *out = (halide_buffer_t *)a[1];
    if (query(in) || query(out)) {   // the output's shape is the request; a host-less input is told the same box
        if (query(in)) {
            int stride = 1;
            for (int d = 0; d < 2; d++) in->dim[d].min = out->dim[d].min, in->dim[d].extent = out->dim[d].extent, in->dim[d].stride = stride, stride *= out->dim[d].extent;
        }
        return 0;
    }
    for (int y = 0; y < out->dim[1].extent; y++)
        for (int x = 0; x < out->dim[0].extent; x++)
            ((float *)out->host)[y * out->dim[1].stride + x] = ((const float *)in->host)[y * in->dim[1].stride + x];
    return 0;
}

extern "C" const halide_filter_metadata_t *fixture_shift_metadata() { return &shift_md; }
extern "C" int fixture_shift_argv(void **a) {
    halide_buffer_t *in = (halide_buffer_t *)a[0], *out = (halide_buffer_t *)a[1];
    if (query(in) || query(out)) {
        if (query(in)) {
            int stride = 1;
            for (int d = 0; d < 2; d++) in->dim[d].min = out->dim[d].min, in->dim[d].extent = out->dim[d].extent + 2, in->dim[d].stride = stride, stride *= in->dim[d].extent;
        }
        return 0;
    }
    for (int d = 0; d < 2; d++)
        if (in->dim[d].min > out->dim[d].min || in->dim[d].min + in->dim[d].extent < out->dim[d].min + out->dim[d].extent + 2) return -4;   // access_out_of_bounds
    for (int y = 0; y < out->dim[1].extent; y++)
        for (int x = 0; x < out->dim[0].extent; x++)
            out->host[y * out->dim[1].stride + x] =
                in->host[(y + out->dim[1].min + 1 - in->dim[1].min) * in->dim[1].stride + (x + out->dim[0].min + 1 - in->dim[0].min)];
    return 0;
}
