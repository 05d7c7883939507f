/* hlmi_runtime.h — the slice of the runtime C API that callers of the AOT entry points use.
 *
 * Every symbol here keeps the NAME, argument meaning and error behaviour of the reference symbol
 * it replaces (paths relative to /root/reference); the HIP-specific ones mirror the reference's
 * per-API runtime header (src/runtime/HalideRuntimeCuda.h) with "hip" in place of the API name.
 * All functions are exported from libhlmi.so with C linkage.
 */
#ifndef HLMI_RUNTIME_H
#define HLMI_RUNTIME_H

#include "hlmi_abi.h"
#ifndef __cplusplus
#include <stdbool.h>
#endif

#ifdef __cplusplus
extern "C" {
#endif

#ifndef HALIDE_HALIDERUNTIME_H /* the reference header declares this block itself (same signatures) */

/* ---- error / print hooks -------------------------------------------------------------------
 * halide_error / halide_set_error_handler: src/runtime/HalideRuntime.h:192-195;
 * default handler prints "Error: <msg>" and abort()s (src/runtime/posix_error_handler.cpp:9-21);
 * halide_set_error_handler returns the previous handler (posix_error_handler.cpp:40). */
typedef void (*halide_error_handler_t)(void *user_context, const char *msg);
void halide_error(void *user_context, const char *msg);
halide_error_handler_t halide_set_error_handler(halide_error_handler_t handler);

/* halide_print / halide_set_custom_print: src/runtime/HalideRuntime.h:178-181 */
typedef void (*halide_print_t)(void *user_context, const char *msg);
void halide_print(void *user_context, const char *msg);
halide_print_t halide_set_custom_print(halide_print_t print);

/* halide_malloc / halide_free (+ custom hooks): src/runtime/HalideRuntime.h:444-451;
 * default returns memory aligned to 128 bytes (src/runtime/posix_allocator.cpp). */
typedef void *(*halide_malloc_t)(void *user_context, size_t x);
typedef void (*halide_free_t)(void *user_context, void *ptr);
void *halide_malloc(void *user_context, size_t x);
void halide_free(void *user_context, void *ptr);
halide_malloc_t halide_set_custom_malloc(halide_malloc_t user_malloc);
halide_free_t halide_set_custom_free(halide_free_t user_free);
/* the built-in allocator, for hooks that wrap it (src/runtime/HalideRuntime.h:446-447; tools/RunGenMain.cpp:220,243) */
void *halide_default_malloc(void *user_context, size_t x);
void halide_default_free(void *user_context, void *ptr);

/* dlsym / dlopen hooks: src/runtime/HalideRuntime.h:465-476, defaults as src/runtime/posix_get_symbol.cpp
 * (halide_get_symbol(name) = dlsym(RTLD_DEFAULT, name)); tools/RunGenMain.cpp:531 probes the runtime through them. */
typedef void *(*halide_get_symbol_t)(const char *name);
typedef void *(*halide_load_library_t)(const char *name);
typedef void *(*halide_get_library_symbol_t)(void *lib, const char *name);
void *halide_get_symbol(const char *name);
void *halide_load_library(const char *name);
void *halide_get_library_symbol(void *lib, const char *name);
void *halide_default_get_symbol(const char *name);
void *halide_default_load_library(const char *name);
void *halide_default_get_library_symbol(void *lib, const char *name);
halide_get_symbol_t halide_set_custom_get_symbol(halide_get_symbol_t user_get_symbol);
halide_load_library_t halide_set_custom_load_library(halide_load_library_t user_load_library);
halide_get_library_symbol_t halide_set_custom_get_library_symbol(halide_get_library_symbol_t user_get_library_symbol);

/* ---- device bookkeeping (dirty-flag protocol) ------------------------------------------------
 * Declared at src/runtime/HalideRuntime.h:908-1011; semantics follow src/runtime/device_interface.cpp:141-330:
 *   halide_copy_to_device : device_malloc if buf->device==0, then H2D iff host_dirty (clears it);
 *   halide_copy_to_host   : D2H iff device_dirty (clears it); host must be non-null;
 *   halide_device_sync    : wait until work queued for this buffer's device has finished;
 *   halide_device_malloc / halide_device_free : attach / release the device allocation;
 *   halide_device_release : free cached allocations + per-device state of an interface. */
int halide_device_malloc(void *user_context, struct halide_buffer_t *buf,
                         const struct halide_device_interface_t *device_interface);
int halide_device_free(void *user_context, struct halide_buffer_t *buf);
int halide_device_sync(void *user_context, struct halide_buffer_t *buf);
int halide_device_sync_global(void *user_context, const struct halide_device_interface_t *device_interface);
int halide_copy_to_host(void *user_context, struct halide_buffer_t *buf);
int halide_copy_to_device(void *user_context, struct halide_buffer_t *buf,
                          const struct halide_device_interface_t *device_interface);
int halide_device_and_host_malloc(void *user_context, struct halide_buffer_t *buf,
                                  const struct halide_device_interface_t *device_interface);
int halide_device_and_host_free(void *user_context, struct halide_buffer_t *buf);
int halide_buffer_copy(void *user_context, struct halide_buffer_t *src,
                       const struct halide_device_interface_t *dst_device_interface, struct halide_buffer_t *dst);
int halide_device_wrap_native(void *user_context, struct halide_buffer_t *buf, uint64_t handle,
                              const struct halide_device_interface_t *device_interface);
int halide_device_detach_native(void *user_context, struct halide_buffer_t *buf);
void halide_device_release(void *user_context, const struct halide_device_interface_t *device_interface);
/* src/runtime/HalideRuntime.h:2343; tools/RunGenMain.cpp:608 calls it with (nullptr, true). */
int halide_reuse_device_allocations(void *user_context, bool flag);

#endif /* !HALIDE_HALIDERUNTIME_H */

/* ---- the HIP device interface (mirrors HalideRuntimeCuda.h:21-82) ----------------------------- */
const struct halide_device_interface_t *halide_hip_device_interface(void);
/* Wrap an existing device pointer (e.g. a torch tensor's data_ptr()) without taking ownership. */
int halide_hip_wrap_device_ptr(void *user_context, struct halide_buffer_t *buf, uint64_t device_ptr);
int halide_hip_detach_device_ptr(void *user_context, struct halide_buffer_t *buf);
uintptr_t halide_hip_get_device_ptr(void *user_context, struct halide_buffer_t *buf);
int halide_hip_release_unused_device_allocations(void *user_context);
/* Device selection: src/runtime/HalideRuntime.h:1019-1026 (halide_set_gpu_device / HL_GPU_DEVICE);
 * -1 = "use HL_GPU_DEVICE or device 0".  The setting is per calling thread, so that one host
 * thread per GPU can drive its own device (the multi-GPU frame sharder does exactly that). */
#ifndef HALIDE_HALIDERUNTIME_H
void halide_set_gpu_device(int n);
int halide_get_gpu_device(void *user_context);
#endif
/* Stream override (mirrors halide_set_cuda_get_stream, HalideRuntimeCuda.h:76-82, simplified to a
 * per-thread value): all work of subsequent calls on this thread is enqueued on `stream`
 * (a hipStream_t); NULL restores the library's own per-device stream.  To enqueue on HIP's null stream (the
 * default stream of torch-ROCm) pass its explicit handle hipStreamLegacy, not 0. */
void halide_hip_set_stream(void *stream);
void *halide_hip_get_stream(void *user_context);
/* One of `nparts` library-owned streams for independent frames in flight ("frame queues"): each has a hardware queue of its
 * own (hipExtStreamCreateWithCUMask; caller-made hipStreamCreate streams share GPU_MAX_HW_QUEUES = 4 queues with every other
 * stream of the process) and launches on it are sized for a 1 / nparts share of the device, i.e. for throughput with
 * nparts frames in flight rather than for the latency of one call.  The queue's CU mask is the full device by default:
 * real CU partitions (HLMI_PART_MASK=1 or 2) measured 3 % slower, and the every-nparts-th-bit mask of rounds 3-5
 * (HLMI_PART_MASK=0) never confined anything on an 8-XCD device — halide_amd/csrc/runtime.cpp, profiles/NOTES.md round 6.
 * The name is kept for callers.  NULL if the device refuses.  No reference counterpart. */
void *halide_hip_partition_stream(int part, int nparts);
/* another queue with the same mask and share as `part` (replica 0 is halide_hip_partition_stream's) */
void *halide_hip_partition_stream_replica(int part, int nparts, int replica);

/* ---- in-process frame sharder (no reference counterpart: the reference has no multi-device layer,
 * its building block is the per-thread halide_set_gpu_device above) ------------------------------------------
 * Runs `fn(frame_args[i])` for every frame i in [0, n_frames), where `fn` is the `<name>_argv` entry point of a
 * pipeline (src/CodeGen_C.cpp:688-694: buffers as halide_buffer_t*, scalars as pointers to their values) and
 * frame_args[i] its argument vector for frame i.  Frames are dealt round-robin to n_devices * streams_per_device
 * workers — one host thread and one HIP stream each, worker w on devices[w % n_devices]; a device listed twice gets
 * two workers (that is also how a one-GPU box exercises the path).  With streams_per_device > 1 the workers of a
 * device each get a frame queue (halide_hip_partition_stream).  Buffers with host data are uploaded to the worker's device; outputs are left
 * device-dirty THERE (halide_copy_to_host / device_sync find the device and stream that produced them).  There is
 * no data-path collective: frames are independent.  Returns 0, or the first error code any frame returned; returns
 * only after every enqueued frame has completed. */
typedef int (*hlmi_argv_fn)(void **args);
int hlmi_run_batch(hlmi_argv_fn fn, void ***frame_args, int n_frames, const int *devices, int n_devices,
                   int streams_per_device);
/* Number of usable gfx950 devices visible to the process (0 if none). */
int hlmi_device_count(void);

/* ---- measurement hooks (no reference counterpart; used by bench.py) ---------------------------
 * When enabled, every kernel launch is bracketed by hipEvents on its stream; the report is a
 * JSON array [{"name":..,"calls":..,"total_ms":..,"avg_ms":..}] written into `out` (returns the
 * number of bytes needed).  Off by default; never enabled inside a timed throughput region. */
void hlmi_kernel_timing_enable(int on);
void hlmi_kernel_timing_reset(void);
/* Measurement only: while `name` is non-empty, every kernel launch of every pipeline whose timing name differs is SKIPPED, so that
 * a caller can enqueue one kernel of a launch chain back to back and time it with nothing in between (bench.py's roofline leg;
 * outputs are meaningless meanwhile).  NULL or "" restores normal operation. */
void hlmi_kernel_timing_only(const char *name);
size_t hlmi_kernel_timing_report(char *out, size_t cap);
/* ---- environment variables -------------------------------------------------------------------------------------------
 * The INTERFACE (what a deployment may set):
 *   HL_GPU_DEVICE            device ordinal, as in the reference's GPU runtimes (src/runtime/HalideRuntime.h:1019-1026)
 *   HLMI_ALLOC_CACHE_MB      cap of the device allocation cache (halide_reuse_device_allocations)
 *   HLMI_LL_GRAPH=1          local_laplacian: replay the launch chain as a HIP graph from the second identical call (opt-in)
 *   HLMI_LL_NO_LUT_CACHE=1, HLMI_CONV_NO_FILTER_CACHE=1, HLMI_CP_NO_SETUP_CACHE=1
 *                            recompute the remap table / bf16 filter image / camera_pipe set-up in every call instead of
 *                            memoising them per (device, parameters)
 * Everything else named HLMI_* in halide_amd/csrc/ is a MEASUREMENT switch, not interface: it forces a slower or more general
 * kernel of the same pipeline (every alternative is bit-identical and is what the parity tests use to cover the general
 * paths on small inputs) or changes a launch geometry; DESIGN.md §7 lists them with what each one measured.  Switches whose
 * experiment is closed are deleted together with their kernels (round 4: ten of local_laplacian's and conv_layer_bf16's). */
/* Library identification: returns e.g. "hlmi 0.1 gfx950". */
const char *hlmi_version(void);
/* Which of the two canonical float forms the kernels were built for (halide_amd/csrc/hlmi_device_math.h): 1 = a multiply with
 * one use that feeds an add / subtract is fused with it, as LLVM contracts the reference's float operations
 * (/root/reference/src/CodeGen_LLVM.cpp:483-500); 0 = one rounding per operator (a `-DHLMI_CANON_FMA=0` build). */
int hlmi_canon_fma(void);

#ifdef __cplusplus
}
#endif
#endif /* HLMI_RUNTIME_H */
