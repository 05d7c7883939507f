// entry_protocol_ref.cpp — TEST INFRASTRUCTURE (own text; built by oracle/ref.mk against the REFERENCE's
// src/runtime/HalideRuntime.h, so every struct, enum value and halide_type_of<>() used here is the reference's).
//
// Walks the cases of the reference's test/generator/error_codes_aottest.cpp:27-120 against a pipeline of THIS library
// with the same argument shape (one 2-D input, one 2-D output of the same extents: stencil_chain, uint16) and against
// local_laplacian for the scalar-parameter cases.  Codes are compared with the reference's halide_error_code_t
// enumerators.  `errors` (default) needs no GPU: every case fails in the prologue; `all` adds the successful run.
#include "HalideRuntime.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "local_laplacian.h"
#include "stencil_chain.h"

namespace {

int failures = 0;
char last_message[1024];

void quiet_handler(void *, const char *msg) {
    strncpy(last_message, msg, sizeof last_message - 1);
}

void expect(const char *what, int got, int want) {
    if (got != want) {
        printf("FAIL %s: returned %d, expected %d (%s)\n", what, got, want, last_message);
        failures++;
    }
}

}  // namespace

int main(int argc, char **argv) {
    const bool with_device = argc > 1 && strcmp(argv[1], "all") == 0;
    halide_error_handler_t previous = halide_set_error_handler(&quiet_handler);
    if (previous == nullptr || previous == &quiet_handler) {
        printf("FAIL halide_set_error_handler did not return the default handler\n");
        failures++;
    }

    const int W = 64, H = 123;
    halide_buffer_t in = {0}, out = {0};
    halide_dimension_t shape[] = {{0, W, 1}, {0, H, W}};
    in.host = (uint8_t *)calloc(W * H, 2);
    in.type = halide_type_of<uint16_t>();
    in.dim = shape;
    in.dimensions = 2;
    in.set_host_dirty();
    out.host = (uint8_t *)calloc(W * H, 2);
    out.type = halide_type_of<uint16_t>();
    out.dim = shape;
    out.dimensions = 2;

    if (with_device) {
        expect("valid call", stencil_chain(&in, &out), halide_error_code_success);
        expect("copy_to_host", halide_copy_to_host(nullptr, &out), halide_error_code_success);
    }

    {   // would read out of bounds: blur_y-style unclamped reads exist in local_laplacian (:84), not in stencil_chain,
        // so the out-of-bounds case uses local_laplacian with an input one column short
        halide_dimension_t full[] = {{0, 64, 1}, {0, 48, 64}, {0, 3, 64 * 48}}, small[] = {{0, 63, 1}, {0, 48, 64}, {0, 3, 64 * 48}};
        halide_buffer_t i3 = in, o3 = out;
        i3.dimensions = o3.dimensions = 3;
        i3.host = (uint8_t *)calloc(64 * 48 * 3, 2), o3.host = (uint8_t *)calloc(64 * 48 * 3, 2);
        i3.dim = small, o3.dim = full;
        expect("input too small", local_laplacian(&i3, 8, 1.0f / 7, 1.0f, &o3), halide_error_code_access_out_of_bounds);
        i3.dim = full;
        expect("levels too small", local_laplacian(&i3, -23, 1.0f / 7, 1.0f, &o3), halide_error_code_param_too_small);
        expect("levels too large", local_laplacian(&i3, (1 << 20) + 1, 1.0f / 7, 1.0f, &o3), halide_error_code_param_too_large);   // the reference declares no bound; the library's is 2^20
        free(i3.host), free(o3.host);
    }

    {   // negative extents that do not trigger the out-of-bounds check
        halide_dimension_t bad_shape[] = {{0, W, 1}, {0, -H, W}};
        halide_buffer_t i = in, o = out;
        i.dim = bad_shape, o.dim = bad_shape;
        expect("negative extent", stencil_chain(&i, &o), halide_error_code_buffer_extents_negative);
    }
    {   // more than 2^31 - 1 elements
        halide_dimension_t huge[] = {{0, 10000000, 1}, {0, 10000000, 64}};
        in.dim = huge;
        expect("extents too large", stencil_chain(&in, &out), halide_error_code_buffer_extents_too_large);
        in.dim = shape;
    }
    {   // addressing that would overflow 32 bits
        halide_dimension_t huge_stride[] = {{0, W, 1}, {0, H, 0x7fffffff}};
        in.dim = huge_stride;
        expect("allocation too large", stencil_chain(&in, &out), halide_error_code_buffer_allocation_too_large);
        in.dim = shape;
    }
    {   // stride[0] is constrained to be 1
        halide_dimension_t wrong_stride[] = {{0, W, 2}, {0, H, W}};
        in.dim = wrong_stride;
        expect("stride 0", stencil_chain(&in, &out), halide_error_code_constraint_violated);
        in.dim = shape;
    }
    expect("null input", stencil_chain(nullptr, &out), halide_error_code_buffer_argument_is_null);
    expect("null output", stencil_chain(&in, nullptr), halide_error_code_buffer_argument_is_null);
    {   // type and dimensionality
        halide_buffer_t i = in;
        i.type = halide_type_of<int>();
        expect("bad type", stencil_chain(&i, &out), halide_error_code_bad_type);
        i = in;
        i.dimensions = 1;
        expect("bad dimensions", stencil_chain(&i, &out), halide_error_code_bad_dimensions);
    }
    {   // the argv form reports the same codes (test/generator/argvcall_aottest.cpp:41-49)
        void *args[2] = {nullptr, &out};
        expect("argv null input", stencil_chain_argv(args), halide_error_code_buffer_argument_is_null);
    }
    {   // bounds query (test/correctness/bounds_query.cpp:24-30): nothing runs, the query buffer is rewritten
        halide_dimension_t qshape[2] = {{0, 0, 0}, {0, 0, 0}};
        halide_buffer_t q = {0};
        q.type = halide_type_of<uint16_t>();
        q.dim = qshape, q.dimensions = 2;
        expect("bounds query", stencil_chain(&q, &out), halide_error_code_success);
        if (!(qshape[0].extent == W && qshape[1].extent == H && qshape[0].stride == 1 && qshape[1].stride == W && q.host == nullptr)) {
            printf("FAIL bounds query: input proposed as [%d,%d] x [%d,%d]\n", qshape[0].min, qshape[0].extent, qshape[1].min,
                   qshape[1].extent);
            failures++;
        }
    }
    if (with_device) {
        if (out.device) halide_device_free(nullptr, &out);
        if (in.device) halide_device_free(nullptr, &in);
    }
    free(in.host);
    free(out.host);
    if (failures) return 1;
    printf("Success!\n");
    return 0;
}
