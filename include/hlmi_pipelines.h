/* hlmi_pipelines.h — the AOT entry points libhlmi.so exports.
 *
 * Each pipeline keeps the exact C signature the reference's code generator emits for it
 * (src/CodeGen_C.cpp:1085-1112: buffers as `struct halide_buffer_t *`, scalars by value, inputs in
 * declaration order then outputs — src/AbstractGenerator.cpp:41-52), plus the argv-call variant
 * (src/CodeGen_C.cpp:675-705) and the metadata getter (src/CodeGen_C.cpp:707-720), so that the
 * object is a drop-in for `<name>.a` + `<name>.h` of the reference.  Return value: 0 or a negative
 * halide_error_code_t; on error `halide_error()` is called first (default handler aborts).
 *
 * Entry protocol, identical for all pipelines, in the order the reference emits its checks
 * (src/UnpackBuffers.cpp:148; src/AddParameterChecks.cpp; src/AddImageChecks.cpp:315-347, 393-470, 591-671,
 * 716-760; buffers are visited in NAME order within each step; pinned by tests/test_entry_protocol.py):
 *   1. a NULL buffer argument                       -> -12 (buffer_argument_is_null)
 *   2. a scalar parameter outside its range         -> -9 / -10 (param_too_small / param_too_large)
 *   3. any buffer with host==NULL && device==0      -> BOUNDS QUERY: dim[] (and type) of every such buffer is
 *      rewritten to the region the pipeline needs / produces, nothing is computed, return 0
 *   4. per buffer: element type mismatch -> -3, then wrong dimensionality -> -43
 *   5. per buffer and dimension: dim[0].stride != 1 or another pinned stride / min / extent -> -8
 *   6. per buffer and dimension: region required > region supplied -> -4, then a negative extent -> -28
 *   7. per buffer and dimension: |extent * stride| > 2^31-1 -> -5, product of extents > 2^31-1 -> -6
 *   8. with the device: both dirty bits -> -37; device handle without interface -> -19 (and vice versa -36); a
 *      device allocation of another API -> -42; a host-dirty input without host pointer -> -34.  (-44,
 *      device_dirty_with_no_device_support, is what a HOST-only target reports; a GPU target copies instead.)
 * Device protocol (what the reference emits for a GPU target,
 * src/InjectHostDevBufferCopies.cpp:197-217, 285-304): inputs are brought to the device with
 * halide_copy_to_device (allocation is attached to the caller's buffer and stays there), the
 * output gets a device allocation, kernels are ENQUEUED on the HIP stream, the output is marked
 * device_dirty and the call returns; the caller uses halide_device_sync / halide_copy_to_host
 * (Halide::Runtime::Buffer::device_sync()/copy_to_host()) exactly as with the reference's GPU
 * targets.  There is no CPU fallback: without a usable gfx950 device the call fails with -29.
 */
#ifndef HLMI_PIPELINES_H
#define HLMI_PIPELINES_H

#include "hlmi_abi.h"

#ifdef __cplusplus
extern "C" {
#endif

#define HLMI_DECLARE_AUX(name)                                   \
    int name##_argv(void **args);                                \
    const struct halide_filter_metadata_t *name##_metadata(void);

/* apps/local_laplacian/local_laplacian_generator.cpp:12-16,287 — u16 [W,H,3] planar in/out,
 * pyramid_levels J=8 (compile-time GeneratorParam :10), `levels` K >= 2 at run time (no upper bound, as in the reference).
 * Drivers pass alpha/(levels-1) (apps/local_laplacian/process.cpp:31). */
int local_laplacian(struct halide_buffer_t *input, int32_t levels, float alpha, float beta,
                    struct halide_buffer_t *output);
HLMI_DECLARE_AUX(local_laplacian)

/* apps/bilateral_grid/bilateral_grid_generator.cpp:10-12,203 — f32 [W,H]; s_sigma=8 compile-time (:8). */
int bilateral_grid(struct halide_buffer_t *input, float r_sigma, struct halide_buffer_t *bilateral_grid);
HLMI_DECLARE_AUX(bilateral_grid)

/* apps/blur/halide_blur_generator.cpp:31-32,117 — u16 [W+2,H+2] -> u16 [W,H]; no boundary condition. */
int halide_blur(struct halide_buffer_t *input, struct halide_buffer_t *blur_y);
HLMI_DECLARE_AUX(halide_blur)

/* apps/nl_means/nl_means_generator.cpp:9-14,162 — f32 [W,H,3] in/out. */
int nl_means(struct halide_buffer_t *input, int32_t patch_size, int32_t search_area, float sigma,
             struct halide_buffer_t *non_local_means);
HLMI_DECLARE_AUX(nl_means)

/* apps/stencil_chain/stencil_chain_generator.cpp:9-10,150 — u16 [W,H], stencils=32 compile-time (:7). */
int stencil_chain(struct halide_buffer_t *input, struct halide_buffer_t *output);
HLMI_DECLARE_AUX(stencil_chain)

/* apps/conv_layer/conv_layer_generator.cpp:9-12,207 — f32 input [CI,W+2,H+2,N], filter [CO,3,3,CI],
 * bias [CO], relu [CO,W,H,N] (c fastest).  The reference pins N=5, CI=CO=128, W=100, H=80 (:15,35-50);
 * this entry point accepts any N, W, H with CI a multiple of 32 and CO a multiple of 128 (superset), dense
 * strides and zero mins as the generator pins them.  Exact f32: a k-ordered fma chain on the f32 matrix cores. */
int conv_layer(struct halide_buffer_t *input, struct halide_buffer_t *filter, struct halide_buffer_t *bias,
               struct halide_buffer_t *relu);
HLMI_DECLARE_AUX(conv_layer)

/* Same algorithm, buffers and layouts as conv_layer (apps/conv_layer/conv_layer_generator.cpp:9-12, :21-27, :35-50),
 * evaluated on the bf16 matrix cores (BASELINE.json configs[4]): input and filter are rounded to bfloat16
 * (nearest even), products accumulate in f32 from the f32 bias.  CI must be a multiple of 64, CO of 128.
 * Not bit-exact against the f32 reference by construction: tolerance parity (tests/test_conv_layer.py). */
int conv_layer_bf16(struct halide_buffer_t *input, struct halide_buffer_t *filter, struct halide_buffer_t *bias,
                    struct halide_buffer_t *relu);
HLMI_DECLARE_AUX(conv_layer_bf16)

/* apps/depthwise_separable_conv/depthwise_separable_conv_generator.cpp:11-23 — f32 input [CI,W,H,N], depthwise_filter
 * [CM,IC,FW,FH] (stride(1) == CM, :283), pointwise_filter [CO,IC], bias [CO], output [CO,W,H,N]; zero padding,
 * depthwise FWxFH convolution, pointwise 1x1 convolution, bias, ReLU.  Bit-exact fma chains in RDom order. */
int depthwise_separable_conv(struct halide_buffer_t *input, struct halide_buffer_t *depthwise_filter,
                             struct halide_buffer_t *pointwise_filter, struct halide_buffer_t *bias,
                             struct halide_buffer_t *output);
HLMI_DECLARE_AUX(depthwise_separable_conv)

/* apps/unsharp/unsharp_generator.cpp:9-10,113 — f32 [W,H,3] planar in and out, sigma = 1.5 (GeneratorParam :7): gray,
 * separable 7-tap Gaussian, sharpen, ratio, recolour.  An adjacent app with the same boundary. */
int unsharp(struct halide_buffer_t *input, struct halide_buffer_t *output);
HLMI_DECLARE_AUX(unsharp)

/* apps/max_filter/max_filter_generator.cpp:11-12,121 — f32 [W,H,3] planar in and out, radius = 26 (GeneratorParam :10):
 * max over a disc-like footprint of the edge-clamped input.  An adjacent app with the same boundary. */
int max_filter(struct halide_buffer_t *input, struct halide_buffer_t *output);
HLMI_DECLARE_AUX(max_filter)

/* apps/hist/hist_generator.cpp:9-10,215 — u8 [W,H,3] planar in and out: histogram equalisation of the luma (integer
 * histogram over the whole input, cdf, pointwise recolouring).  An adjacent app with the same boundary. */
int hist(struct halide_buffer_t *input, struct halide_buffer_t *output);
HLMI_DECLARE_AUX(hist)

/* apps/harris/harris_generator.cpp:13-15,130 — f32 [W,H,3] planar in, f32 [.,.] out: Harris corner response; no
 * boundary condition (the input must cover the output grown by 2).  Adjacent app, same boundary. */
int harris(struct halide_buffer_t *input, struct halide_buffer_t *output);
HLMI_DECLARE_AUX(harris)

/* apps/interpolate/interpolate_generator.cpp:17-18,215 — f32 [W,H,4] planar in (r, g, b, alpha), f32 [W,H,3] out over
 * the input's extent: alpha-weighted pull-push pyramid, 10 levels.  Adjacent app, same boundary. */
int interpolate(struct halide_buffer_t *input, struct halide_buffer_t *output);
HLMI_DECLARE_AUX(interpolate)

/* apps/iir_blur/iir_blur_generator.cpp:136-144,181 — f32 [W,H,C] planar in and out, `alpha` = weight of the input:
 * first-order IIR low pass down and up the columns, then along the rows.  Adjacent app; the reference
 * pins 1536 x 2560 x 3 (:158-163), this entry point accepts any extents. */
int iir_blur(struct halide_buffer_t *input, float alpha, struct halide_buffer_t *output);
HLMI_DECLARE_AUX(iir_blur)

/* apps/lens_blur/lens_blur_generator.cpp:12-21,301 — u8 [W,H,3] planar stereo pair in, f32 [W,H,3] out: depth from
 * stereo (cost volume of `slices` disparities, confidence-weighted 8-level push-pull), depth-dependent bokeh from
 * `aperture_samples` pseudo-random samples per pixel.  Adjacent app, same boundary.  The sample
 * positions are Halide's random_float(): a fixed hash whose "definition tag" is a counter of the reference's COMPILER that
 * cannot be observed without it — see hlmi_lens_blur_set_random_tag below and oracle/lens_blur_oracle.c.
 * Limit of this implementation: the output plus its blur radius on either side must be narrower than 2^23 columns
 * (halide_error_code_buffer_extents_too_large otherwise). */
int lens_blur(struct halide_buffer_t *left_im, struct halide_buffer_t *right_im, int32_t slices, int32_t focus_depth,
              float blur_radius_scale, int32_t aperture_samples, struct halide_buffer_t *final);
HLMI_DECLARE_AUX(lens_blur)
/* The tag the random_float() calls of lens_blur's sample_locations were lowered with (src/Function.cpp:640-648: the number
 * of pure Func definitions the generator process made before it).  Default 71 = the count derived in
 * oracle/lens_blur_oracle.c; a maintainer who can run the reference's compiler sets the observed value. */
void hlmi_lens_blur_set_random_tag(int tag);
int hlmi_lens_blur_get_random_tag(void);

/* apps/bgu/bgu_generator.cpp:252-266,698 — bilateral-guided upsampling: fits a 3x4 affine colour transform per cell of
 * a bilateral grid (cells of s_sigma x s_sigma low-res pixels x r_sigma of luma) from the low-res pair splat_loc ->
 * values, and applies the trilinearly sliced transforms to the full-res slice_loc.  f32 [W,H,3] planar everywhere; the
 * low-res pair is edge-clamped (:270-271).  Adjacent app, same boundary.  fast_inverse (:170) is the
 * correctly rounded 1/x of the reference's CUDA path (src/runtime/ptx_dev.ll:61-66), not x86's rcpss estimate. */
int bgu(float r_sigma, int32_t s_sigma, struct halide_buffer_t *splat_loc, struct halide_buffer_t *values,
        struct halide_buffer_t *slice_loc, struct halide_buffer_t *output);
HLMI_DECLARE_AUX(bgu)

/* apps/camera_pipe/camera_pipe_generator.cpp:219-228,622 — raw u16 Bayer -> u8 [W,H,3]. */
int camera_pipe(struct halide_buffer_t *input, struct halide_buffer_t *matrix_3200,
                struct halide_buffer_t *matrix_7000, float color_temp, float gamma, float contrast,
                float sharpen_strength, int32_t blackLevel, int32_t whiteLevel,
                struct halide_buffer_t *processed);
HLMI_DECLARE_AUX(camera_pipe)

/* `<name>_auto_schedule` variants: the reference's drivers link both objects
 * (apps/local_laplacian/process.cpp:5-7,42-49); same algorithm, so they alias the entry above. */
int local_laplacian_auto_schedule(struct halide_buffer_t *input, int32_t levels, float alpha, float beta,
                                  struct halide_buffer_t *output);
int bilateral_grid_auto_schedule(struct halide_buffer_t *input, float r_sigma, struct halide_buffer_t *out);
int halide_blur_auto_schedule(struct halide_buffer_t *input, struct halide_buffer_t *blur_y);
int nl_means_auto_schedule(struct halide_buffer_t *input, int32_t patch_size, int32_t search_area, float sigma,
                           struct halide_buffer_t *non_local_means);
int stencil_chain_auto_schedule(struct halide_buffer_t *input, struct halide_buffer_t *output);
int conv_layer_auto_schedule(struct halide_buffer_t *input, struct halide_buffer_t *filter,
                             struct halide_buffer_t *bias, struct halide_buffer_t *relu);
int depthwise_separable_conv_auto_schedule(struct halide_buffer_t *input, struct halide_buffer_t *depthwise_filter,
                                           struct halide_buffer_t *pointwise_filter, struct halide_buffer_t *bias,
                                           struct halide_buffer_t *output);
int unsharp_auto_schedule(struct halide_buffer_t *input, struct halide_buffer_t *output);
int max_filter_auto_schedule(struct halide_buffer_t *input, struct halide_buffer_t *output);
int hist_auto_schedule(struct halide_buffer_t *input, struct halide_buffer_t *output);
int harris_auto_schedule(struct halide_buffer_t *input, struct halide_buffer_t *output);
int interpolate_auto_schedule(struct halide_buffer_t *input, struct halide_buffer_t *output);
int iir_blur_auto_schedule(struct halide_buffer_t *input, float alpha, struct halide_buffer_t *output);
int lens_blur_auto_schedule(struct halide_buffer_t *left_im, struct halide_buffer_t *right_im, int32_t slices,
                            int32_t focus_depth, float blur_radius_scale, int32_t aperture_samples,
                            struct halide_buffer_t *final);
int bgu_auto_schedule(float r_sigma, int32_t s_sigma, struct halide_buffer_t *splat_loc, struct halide_buffer_t *values,
                      struct halide_buffer_t *slice_loc, struct halide_buffer_t *output);
int camera_pipe_auto_schedule(struct halide_buffer_t *input, struct halide_buffer_t *matrix_3200,
                              struct halide_buffer_t *matrix_7000, float color_temp, float gamma, float contrast,
                              float sharpen_strength, int32_t blackLevel, int32_t whiteLevel,
                              struct halide_buffer_t *processed);

#undef HLMI_DECLARE_AUX

#ifdef __cplusplus
}
#endif
#endif /* HLMI_PIPELINES_H */
