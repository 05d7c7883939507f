// hlmi_internal.h — shared plumbing between the runtime slice and the per-pipeline host shims.
// Nothing here is exported; the exported C ABI is declared in include/hlmi_runtime.h and
// include/hlmi_pipelines.h.
#pragma once

#include <hip/hip_runtime.h>
#include <mutex>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "hlmi_pipelines.h"
#include "hlmi_runtime.h"

namespace hlmi {

// ---------------------------------------------------------------------------------------------
// errors: format a message, hand it to halide_error() (default handler aborts, like the reference's
// posix_error_handler.cpp:9-21) and return the code so callers can `return report(...)`.
int report(void *uc, int code, const char *fmt, ...) __attribute__((format(printf, 3, 4)));
int hip_failed(void *uc, hipError_t e, const char *what);  // -> halide_error_code_gpu_device_error

#define HLMI_HIP(uc, call)                                           \
    do {                                                             \
        hipError_t e__ = (call);                                     \
        if (e__ != hipSuccess) return ::hlmi::hip_failed(uc, e__, #call); \
    } while (0)

// ---------------------------------------------------------------------------------------------
// ABI helpers
constexpr uint32_t type_abi(int code, int bits) { return (uint32_t)code | ((uint32_t)bits << 8); }
constexpr uint32_t T_U8 = type_abi(1, 8), T_U16 = type_abi(1, 16), T_I16 = type_abi(0, 16),
                   T_I32 = type_abi(0, 32), T_F32 = type_abi(2, 32);
inline uint32_t buf_type_abi(const halide_buffer_t *b) {
    uint32_t v;
    memcpy(&v, &b->type, 4);
    return v;
}
const char *type_name(uint32_t abi, char tmp[16]);

struct BufArg {
    const char *name;
    halide_buffer_t *buf;
    uint32_t type;  // required element type (type_abi)
    int dims;       // required dimensionality
    bool is_output;
};

// ---------------------------------------------------------------------------------------------
// The entry prologue.  The reference emits its argument checks in a fixed order (src/AddImageChecks.cpp:716-760,
// reverse of the prepend order; buffers in the order of a std::map keyed by buffer name, i.e. alphabetical):
//   0. null buffer arguments, in signature order                       (src/UnpackBuffers.cpp:148)          -12
//   1. scalar parameter ranges                                           (src/AddParameterChecks.cpp)         -9 / -10
//   2. bounds-query mode: rewrite the query buffers, return 0            (:709-713)
//   3. per buffer: type, then dimensionality                             (asserts_type_checks, :329-347)      -3 / -43
//   4. per buffer, per dimension: stride / min / extent constraints      (asserts_constrained, :621-645)      -8
//   5. per buffer, per dimension: required region, then extent >= 0      (asserts_required, :414-418, :466-470) -4 / -28
//   6. per buffer, per dimension: |extent * stride| and the running product of extents <= 2^31 - 1
//                                                                        (dims_no_overflow_asserts, :436-462) -5 / -6
//   7. host pointers                                                     (asserts_host_non_null, :648-655)    -34
// The pipelines call the check_* helpers below in whatever order is convenient for them; failures of phases 4-6 are
// not reported on the spot but RECORDED with their (phase, buffer rank, dimension, kind) key, and the one the
// reference would have hit first is reported by checks_done() — which acquire_device() calls, so no kernel is ever
// enqueued with unchecked arguments.  tests/test_entry_protocol.py pins the codes and the order.
// step 0; also (re)starts the recording for this call and ranks the buffers by name
int check_not_null(void *uc, const BufArg *args, int n);
// step 2 (src/AddImageChecks.cpp:315-318, HalideRuntime.h:1851-1853)
bool any_bounds_query(const BufArg *args, int n);
// step 3, reported immediately (everything after it indexes dim[]).  In bounds-query mode the reference skips the
// type check and rewrites type and dimensions of the query buffers (BufferBuilder, :478-494): here a query buffer gets
// its type rewritten too, but a wrong dimensionality stays an error (-43) — writing dim[] entries the caller did not
// provide is not something a drop-in should copy.
int check_type_and_dims(void *uc, const BufArg *args, int n);
// steps 4-6 for one buffer: dim[0].stride == 1 (src/Parameter.cpp:30-35), extents >= 0, sizes < 2^31.  Always
// returns 0 (recorded).
int check_shape(void *uc, const BufArg &a);
// step 5: [min, min+extent) of dimension d must cover [req_min, req_min+req_extent)
// (src/AddImageChecks.cpp:393-418 -> halide_error_access_out_of_bounds).  Always returns 0 (recorded).
int check_covers(void *uc, const BufArg &a, int d, int req_min, int req_extent);
// step 4, a pinned constraint "<buffer>.<field>.<dim> == expect" (halide_error_constraint_violated,
// src/runtime/errors.cpp:104); `what` must start with the buffer's name.  Always returns 0 (recorded).
int check_equal(void *uc, const char *what, int val, const char *expect_what, int expect);
// report the recorded failure the reference would have hit first (0 if none) and forget the rest
int checks_done(void *uc);
// bounds-query answer: rewrite dim[] to a dense planar shape (stride[0]=1) — only if `buf` is itself
// a query buffer (host==NULL && device==0), as the reference does (AddImageChecks.cpp:480-497).
void answer_query(halide_buffer_t *buf, const int *mins, const int *extents);

// ---------------------------------------------------------------------------------------------
// device context
struct DeviceCtx {
    int device = -1;
    hipStream_t stream = nullptr;
    // Held from acquire_device() until the context goes out of scope at the end of the entry point: calls that share
    // a (device, stream) — and therefore its scratch arena — enqueue their launches one call at a time, so that stream
    // order alone makes the shared arena safe when several host threads call pipelines concurrently (the reference's
    // generated code is re-entrant, test/generator/gpu_multi_context_threaded_aottest.cpp).  Threads that want their
    // calls to overlap on the GPU use distinct streams (halide_hip_set_stream).
    std::unique_lock<std::recursive_mutex> call_lock;
};
// choose device (halide_set_gpu_device / HL_GPU_DEVICE / 0), hipSetDevice, choose stream.
// Fails with -29 when no gfx950 device is usable: there is NO CPU fallback.
// `lock` = take the stream's call lock (pipeline entry points do; copies, syncs and allocations do not: they never
// touch the scratch arena, and holding the lock across a blocking wait would stall every other caller of the stream).
// Reports the failure recorded by the check_* helpers first, if there is one.
int acquire_device(void *uc, DeviceCtx *ctx, bool lock = true);
// scratch arena owned by (device, stream); contents valid until the next call that asks for a
// workspace on the same stream (stream order makes reuse across back-to-back calls safe).
int get_workspace(void *uc, const DeviceCtx &ctx, size_t bytes, void **ptr);

// dirty-flag protocol for pipeline arguments (src/InjectHostDevBufferCopies.cpp:197-217,285-304)
int input_to_device(void *uc, const DeviceCtx &ctx, const BufArg &a);
int output_on_device(void *uc, const DeviceCtx &ctx, const BufArg &a);
void mark_output_written(halide_buffer_t *buf);  // device_dirty = 1, host_dirty = 0
// A number that changes whenever the device contents of `buf` may have changed through this runtime (upload of a
// host-dirty buffer, use as a pipeline output or copy target, re-allocation); unique across allocations, so
// (device handle, version) identifies contents.  0 = memory this runtime does not own (wrapped native pointers: their
// owner can rewrite them behind our back) — callers must not cache anything derived from such a buffer.
uint64_t buffer_version(const halide_buffer_t *buf);
// compute units a launch on `stream` is sized for (a frame-queue stream of halide_hip_partition_stream: its 1 / nparts share)
int stream_cu_count(int device, hipStream_t stream);

// Bounds-query helpers.  A buffer takes part in deriving the other buffers' regions when it is real (host or device set) or,
// in query mode, when the caller gave it a shape: RunGen's queries pass EVERY buffer with host == device == 0 — outputs shaped
// by --output_extents / the estimates / the first input, inputs shaped as loaded (tools/RunGen.h:1212-1250) — and Halide's
// bounds inference takes the outputs' shapes as the request whether or not they are allocated.
inline bool buffer_is_real(const halide_buffer_t *b) { return !(b->host == nullptr && b->device == 0); }
inline bool buffer_has_shape(const halide_buffer_t *b) {
    for (int d = 0; d < b->dimensions; d++)
        if (b->dim[d].extent <= 0) return false;
    return b->dimensions > 0;
}
inline bool buffer_known(const halide_buffer_t *b) { return buffer_is_real(b) || buffer_has_shape(b); }

// ---- events across streams.  The special stream handles (NULL, hipStreamLegacy — which callers on torch's default stream pass
// to halide_hip_set_stream — and hipStreamPerThread) launch kernels fine, but they are not safe partners for events in ROCm
// 7.2: an event RECORDED on hipStreamLegacy crashes the next hipStreamWaitEvent on it (scripts/legacy_event_probe.py), and
// one recorded on the NULL handle and waited for from several host threads at once threw std::bad_variant_access inside the
// runtime (tests/test_torch_ops.py followed by tests/test_threads.py).  So no event ever names a special stream: work on one
// is waited for on the host, which leaves the event in its "nothing pending" state.
inline bool stream_is_special(hipStream_t s) { return s == nullptr || s == hipStreamLegacy || s == hipStreamPerThread; }
// "everything enqueued on `producer` so far" as an event other streams can wait for
inline hipError_t record_done(hipEvent_t ev, hipStream_t producer) {
    if (stream_is_special(producer)) return hipStreamSynchronize(producer == hipStreamLegacy ? nullptr : producer);
    return hipEventRecord(ev, producer);
}
// whatever is enqueued on `consumer` from now on happens after `ev`
inline hipError_t wait_done(hipStream_t consumer, hipEvent_t ev) {
    if (stream_is_special(consumer)) return hipEventSynchronize(ev);
    return hipStreamWaitEvent(consumer, ev, 0);
}
// timing events only (never waited for by a stream): hipStreamLegacy is the NULL stream
inline hipStream_t event_stream(hipStream_t s) { return s == hipStreamLegacy ? nullptr : s; }

template<typename T>
inline T *dev_ptr(const halide_buffer_t *b) { return reinterpret_cast<T *>((uintptr_t)b->device); }

// ---------------------------------------------------------------------------------------------
// optional per-kernel HIP-event timing (include/hlmi_runtime.h: hlmi_kernel_timing_*)
bool timing_enabled();
void timing_begin(const char *name, hipStream_t s);
void timing_end(hipStream_t s);
// declares the ALGORITHMIC bytes (compulsory reads + writes) of the next timed launch of this thread
void timing_note_bytes(double bytes);
struct ScopedKernelTimer {
    hipStream_t s;
    bool on;
    ScopedKernelTimer(const char *name, hipStream_t stream) : s(stream), on(timing_enabled()) {
        if (on) timing_begin(name, s);
    }
    ~ScopedKernelTimer() {
        if (on) timing_end(s);
    }
};
// Run-time switches (HLMI_*): one helper so that every switch reads the same way — set and non-zero = on, unset / empty / "0" = off.
// Read at every call (a getenv, ~100 ns): the parity tests flip switches inside one process to reach the alternative paths.
bool env_flag(const char *name);
// launch + error check; kernel errors surface as -23 (device_run_failed)
int launch_failed(void *uc, const char *kernel);
// measurement only (hlmi_kernel_timing_only): while a launch name is selected every OTHER launch is skipped, so that a
// caller can run one kernel of a chain back to back (its inputs are whatever earlier, complete calls left in the workspace)
bool launch_selected(const char *name);
#define HLMI_LAUNCH(uc, name, stream, kernel, grid, block, shmem, ...)                      \
    do {                                                                                    \
        if (!::hlmi::launch_selected(name)) break;                                          \
        ::hlmi::ScopedKernelTimer t__(name, stream);                                        \
        hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__);                \
        if (hipGetLastError() != hipSuccess) return ::hlmi::launch_failed(uc, name);        \
    } while (0)

// ---------------------------------------------------------------------------------------------
// metadata helper: every pipeline defines a static halide_filter_metadata_t
// (layout: src/runtime/HalideRuntime.h:1937-1975; contents as emitted by src/CodeGen_C.cpp:760-912)
extern const char *const kTargetString;  // "x86-64-linux-hip-gfx950" (canonical-style target string)

// conv_layer.hip: argument protocol shared by conv_layer and conv_layer_bf16
int conv_check_args(void *uc, BufArg *args, int *CI, int *CO, int *W, int *H, int *N, bool *query);

inline int floor_div(int a, int b) {  // b > 0 ; Halide integer division rounds toward -inf (src/IR.h:145-166)
    int q = a / b, r = a % b;
    return (r != 0 && r < 0) ? q - 1 : q;
}

}  // namespace hlmi
