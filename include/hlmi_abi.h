/* hlmi_abi.h — the buffer/argument ABI of the drop-in boundary.
 *
 * This header RE-DECLARES (own text, identical memory layout) the plain-C types that every
 * AOT pipeline entry point of the reference takes, so that libhlmi.so can be built and called
 * without the reference tree.  A caller that already includes the reference's runtime header
 * (include guard HALIDE_HALIDERUNTIME_H) gets the reference's own declarations instead; the two
 * are layout-identical, which tests/test_abi_layout.py proves with static_asserts against the
 * reference header where that tree is present.
 *
 * Reference interfaces replaced / mirrored (paths relative to /root/reference):
 *   halide_type_t             src/runtime/HalideRuntime.h:521-580   (4 bytes: code u8, bits u8, reserved u16)
 *   halide_dimension_t        src/runtime/HalideRuntime.h:1657-1684 (16 bytes: min, extent, stride, flags)
 *   halide_buffer_t           src/runtime/HalideRuntime.h:1710-1856 (56 bytes)
 *   halide_buffer_flags       src/runtime/HalideRuntime.h:1699-1702
 *   halide_device_interface_t src/runtime/HalideRuntime.h:875-897
 *   halide_error_code_t       src/runtime/HalideRuntime.h:1152-1357
 *   halide_scalar_value_t, halide_filter_argument_t, halide_filter_metadata_t
 *                             src/runtime/HalideRuntime.h:1880-1975
 */
#ifndef HLMI_ABI_H
#define HLMI_ABI_H

#include <stddef.h>
#include <stdint.h>

#ifndef HALIDE_HALIDERUNTIME_H /* the reference header, when present, wins */

#ifdef __cplusplus
extern "C" {
#endif

/* ---- element type ------------------------------------------------------------------------- */
enum hlmi_type_code {
    halide_type_int = 0,
    halide_type_uint = 1,
    halide_type_float = 2,
    halide_type_handle = 3,
    halide_type_bfloat = 4
};

struct halide_type_t {
    uint8_t code;      /* hlmi_type_code */
    uint8_t bits;      /* bits per element */
    uint16_t reserved; /* must be 0 */
};

/* ---- one dimension of a buffer ------------------------------------------------------------ */
typedef struct halide_dimension_t {
    int32_t min;    /* coordinate of the first element along this dimension */
    int32_t extent; /* number of elements */
    int32_t stride; /* distance between consecutive elements, in ELEMENTS */
    uint32_t flags; /* reserved */
} halide_dimension_t;

/* ---- buffer flags ------------------------------------------------------------------------- */
enum { halide_buffer_flag_host_dirty = 1,
       halide_buffer_flag_device_dirty = 2 };

struct halide_device_interface_t;

/* ---- the buffer descriptor ---------------------------------------------------------------- */
typedef struct halide_buffer_t {
    uint64_t device;                                         /* device allocation handle, 0 = none        */
    const struct halide_device_interface_t *device_interface;/* how to interpret `device`                 */
    uint8_t *host;                                           /* address of element (min0, min1, ...)      */
    uint64_t flags;                                          /* halide_buffer_flag_*                      */
    struct halide_type_t type;
    int32_t dimensions;
    halide_dimension_t *dim;                                 /* caller-owned array of `dimensions` entries */
    void *padding;
} halide_buffer_t;

/* ---- device interface: table of function pointers a Buffer uses to manage `device` -------- */
struct halide_device_interface_impl_t;
struct halide_device_interface_t {
    int (*device_malloc)(void *user_context, struct halide_buffer_t *buf,
                         const struct halide_device_interface_t *device_interface);
    int (*device_free)(void *user_context, struct halide_buffer_t *buf);
    int (*device_sync)(void *user_context, struct halide_buffer_t *buf);
    void (*device_release)(void *user_context, const struct halide_device_interface_t *device_interface);
    int (*copy_to_host)(void *user_context, struct halide_buffer_t *buf);
    int (*copy_to_device)(void *user_context, struct halide_buffer_t *buf,
                          const struct halide_device_interface_t *device_interface);
    int (*device_and_host_malloc)(void *user_context, struct halide_buffer_t *buf,
                                  const struct halide_device_interface_t *device_interface);
    int (*device_and_host_free)(void *user_context, struct halide_buffer_t *buf);
    int (*buffer_copy)(void *user_context, struct halide_buffer_t *src,
                       const struct halide_device_interface_t *dst_device_interface, struct halide_buffer_t *dst);
    int (*device_crop)(void *user_context, const struct halide_buffer_t *src, struct halide_buffer_t *dst);
    int (*device_slice)(void *user_context, const struct halide_buffer_t *src, int slice_dim, int slice_pos,
                        struct halide_buffer_t *dst);
    int (*device_release_crop)(void *user_context, struct halide_buffer_t *buf);
    int (*wrap_native)(void *user_context, struct halide_buffer_t *buf, uint64_t handle,
                       const struct halide_device_interface_t *device_interface);
    int (*detach_native)(void *user_context, struct halide_buffer_t *buf);
    int (*compute_capability)(void *user_context, int *major, int *minor);
    const struct halide_device_interface_impl_t *impl;
};

/* ---- error codes returned by every entry point (0 = success) ------------------------------- */
enum hlmi_error_code {
    halide_error_code_success = 0,
    halide_error_code_generic_error = -1,
    halide_error_code_explicit_bounds_too_small = -2,
    halide_error_code_bad_type = -3,
    halide_error_code_access_out_of_bounds = -4,
    halide_error_code_buffer_allocation_too_large = -5,
    halide_error_code_buffer_extents_too_large = -6,
    halide_error_code_constraints_make_required_region_smaller = -7,
    halide_error_code_constraint_violated = -8,
    halide_error_code_param_too_small = -9,
    halide_error_code_param_too_large = -10,
    halide_error_code_out_of_memory = -11,
    halide_error_code_buffer_argument_is_null = -12,
    halide_error_code_copy_to_host_failed = -14,
    halide_error_code_copy_to_device_failed = -15,
    halide_error_code_device_malloc_failed = -16,
    halide_error_code_device_sync_failed = -17,
    halide_error_code_device_free_failed = -18,
    halide_error_code_no_device_interface = -19,
    halide_error_code_unimplemented = -20,
    halide_error_code_internal_error = -22,
    halide_error_code_device_run_failed = -23,
    halide_error_code_requirement_failed = -27,
    halide_error_code_buffer_extents_negative = -28,
    halide_error_code_gpu_device_error = -29,
    halide_error_code_device_wrap_native_failed = -32,
    halide_error_code_device_detach_native_failed = -33,
    halide_error_code_host_is_null = -34,
    halide_error_code_device_interface_no_device = -36,
    halide_error_code_host_and_device_dirty = -37,
    halide_error_code_buffer_is_null = -38,
    halide_error_code_device_buffer_copy_failed = -39,
    halide_error_code_device_crop_unsupported = -40,
    halide_error_code_device_crop_failed = -41,
    halide_error_code_incompatible_device_interface = -42,
    halide_error_code_bad_dimensions = -43,
    halide_error_code_device_dirty_with_no_device_support = -44
};

/* ---- argv-call metadata ------------------------------------------------------------------- */
struct halide_scalar_value_t {
    union {
        uint8_t b; /* bool */
        int8_t i8;
        int16_t i16;
        int32_t i32;
        int64_t i64;
        uint8_t u8;
        uint16_t u16;
        uint32_t u32;
        uint64_t u64;
        float f32;
        double f64;
        void *handle;
    } u;
};

enum { halide_argument_kind_input_scalar = 0,
       halide_argument_kind_input_buffer = 1,
       halide_argument_kind_output_buffer = 2 };

struct halide_filter_argument_t {
    const char *name;
    int32_t kind;       /* halide_argument_kind_* */
    int32_t dimensions; /* 0 for scalars */
    struct halide_type_t type;
    const struct halide_scalar_value_t *scalar_def, *scalar_min, *scalar_max, *scalar_estimate;
    int64_t const *const *buffer_estimates; /* 2*dimensions pointers: &min0,&extent0,&min1,... */
};

struct halide_filter_metadata_t {
    int32_t version; /* 1 */
    int32_t num_arguments;
    const struct halide_filter_argument_t *arguments;
    const char *target;
    const char *name;
};

#ifdef __cplusplus
} /* extern "C" */
#endif

#endif /* !HALIDE_HALIDERUNTIME_H */
#endif /* HLMI_ABI_H */
