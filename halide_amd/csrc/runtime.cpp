// runtime.cpp — the slice of the runtime that callers of the AOT entry points touch:
// error/print/malloc hooks, the HIP device interface with the reference's dirty-flag protocol,
// per-device streams + scratch arenas + an allocation cache, and optional per-kernel event timing.
//
// Reference behaviour restated here (paths relative to /root/reference):
//   src/runtime/posix_error_handler.cpp:9-41   default error handler prints and aborts
//   src/runtime/posix_allocator.cpp            halide_malloc alignment
//   src/runtime/device_interface.cpp:28-727    dirty-flag state machine, validation order
//   src/runtime/cuda.cpp:586-760               allocation cache ("reuse_device_allocations")
//   src/runtime/gpu_device_selection.cpp:29    HL_GPU_DEVICE
#include <atomic>
#include <map>
#include <mutex>
#include <tuple>
#include <string>
#include <unordered_map>
#include <vector>

#include <dlfcn.h>
#include <stdlib.h>

#include "hlmi_internal.h"
#include "hlmi_device_math.h"   // HLMI_CANON_FMA

namespace hlmi {

const char *const kTargetString = "x86-64-linux-hip-gfx950";

// ------------------------------------------------------------------------------------------------
// hooks
static void default_error_handler(void *, const char *msg) {
    fprintf(stderr, "Error: %s\n", msg);
    fflush(stderr);
    abort();
}
static void default_print(void *, const char *msg) { fputs(msg, stderr); }
static void *default_malloc(void *, size_t x) {
    void *p = nullptr;
    // 128-byte alignment like the reference allocator / Buffer<> (HalideBuffer.h:894-951)
    if (posix_memalign(&p, 128, (x + 127) & ~(size_t)127) != 0) return nullptr;
    return p;
}
static void default_free(void *, void *p) { free(p); }

static std::atomic<halide_error_handler_t> g_error_handler{default_error_handler};
static std::atomic<halide_print_t> g_print{default_print};
static std::atomic<halide_malloc_t> g_malloc{default_malloc};
static std::atomic<halide_free_t> g_free{default_free};

// ---- recorded check failures (see hlmi_internal.h: the entry prologue) ------------------------------
struct PendingFailure {
    bool any = false;
    uint32_t key = 0;  // (phase << 24) | (buffer rank << 16) | (dimension << 8) | kind — smaller = checked earlier
    int code = 0;
    char msg[512];
};
static thread_local PendingFailure t_fail;
static thread_local const BufArg *t_args = nullptr;
static thread_local int t_nargs = 0;
static thread_local int t_rank[16];

static int rank_of(const char *name, size_t len) {
    for (int i = 0; i < t_nargs && i < 16; i++) {
        if (strlen(t_args[i].name) == len && strncmp(t_args[i].name, name, len) == 0) return t_rank[i];
    }
    return 15;
}

static int record(int phase, int rank, int dim, int kind, int code, const char *fmt, ...) __attribute__((format(printf, 6, 7)));
static int record(int phase, int rank, int dim, int kind, int code, const char *fmt, ...) {
    uint32_t key = ((uint32_t)phase << 24) | ((uint32_t)(rank & 0xff) << 16) | ((uint32_t)(dim & 0xff) << 8) | (uint32_t)(kind & 0xff);
    if (t_fail.any && t_fail.key <= key) return 0;
    t_fail.any = true, t_fail.key = key, t_fail.code = code;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(t_fail.msg, sizeof t_fail.msg, fmt, ap);
    va_end(ap);
    return 0;
}

int checks_done(void *uc) {
    if (!t_fail.any) return 0;
    t_fail.any = false;
    halide_error(uc, t_fail.msg);
    return t_fail.code;
}

int report(void *uc, int code, const char *fmt, ...) {
    // a failure recorded earlier in the prologue pre-empts anything found later — except the scalar-parameter range
    // checks, which the reference performs before every image check (src/Lower.cpp:189 vs :251)
    if (t_fail.any && code != halide_error_code_param_too_small && code != halide_error_code_param_too_large &&
        code != halide_error_code_buffer_argument_is_null) {
        return checks_done(uc);
    }
    t_fail.any = false;
    char msg[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(msg, sizeof msg, fmt, ap);
    va_end(ap);
    halide_error(uc, msg);
    return code;
}

int hip_failed(void *uc, hipError_t e, const char *what) {
    return report(uc, halide_error_code_gpu_device_error, "HIP: %s failed: %s", what, hipGetErrorString(e));
}

int launch_failed(void *uc, const char *kernel) {
    return report(uc, halide_error_code_device_run_failed, "HIP: launch of kernel %s failed", kernel);
}

const char *type_name(uint32_t abi, char tmp[16]) {
    static const char *codes[] = {"int", "uint", "float", "handle", "bfloat"};
    unsigned code = abi & 0xff, bits = (abi >> 8) & 0xff;
    snprintf(tmp, 16, "%s%u", code < 5 ? codes[code] : "type?", bits);
    return tmp;
}

// ------------------------------------------------------------------------------------------------
// argument checks
int check_not_null(void *uc, const BufArg *args, int n) {
    t_fail.any = false;
    t_args = args, t_nargs = n;
    for (int i = 0; i < n; i++) {
        if (!args[i].buf) {
            return report(uc, halide_error_code_buffer_argument_is_null, "Buffer argument %s is nullptr",
                          args[i].name);
        }
    }
    for (int i = 0; i < n && i < 16; i++) {
        int r = 0;
        for (int j = 0; j < n; j++) r += strcmp(args[j].name, args[i].name) < 0;
        t_rank[i] = r;
    }
    return 0;
}

bool any_bounds_query(const BufArg *args, int n) {
    for (int i = 0; i < n; i++) {
        if (args[i].buf->host == nullptr && args[i].buf->device == 0) return true;
    }
    return false;
}

int check_type_and_dims(void *uc, const BufArg *args, int n) {
    const bool query = any_bounds_query(args, n);
    // buffers in name order; per buffer the type first, then the dimensionality
    for (int rank = 0; rank < n; rank++) {
        for (int i = 0; i < n; i++) {
            if (i < 16 && t_rank[i] != rank) continue;
            const BufArg &a = args[i];
            const char *io = a.is_output ? "Output" : "Input";
            uint32_t given = buf_type_abi(a.buf);
            if (given != a.type) {
                if (query) {
                    if (a.buf->host == nullptr && a.buf->device == 0) memcpy(&a.buf->type, &a.type, 4);
                } else {
                    char t0[16], t1[16];
                    return report(uc, halide_error_code_bad_type,
                                  "%s buffer %s has type %s but type of the buffer passed in is %s", io, a.name,
                                  type_name(a.type, t0), type_name(given, t1));
                }
            }
            if (a.buf->dimensions != a.dims) {
                return report(uc, halide_error_code_bad_dimensions,
                              "%s buffer %s requires a buffer of exactly %d dimensions, but the buffer passed in has %d "
                              "dimensions",
                              io, a.name, a.dims, a.buf->dimensions);
            }
            if (a.dims > 0 && a.buf->dim == nullptr) {
                return report(uc, halide_error_code_buffer_is_null, "%s buffer %s has a null dim pointer", io, a.name);
            }
        }
    }
    return 0;
}

int check_shape(void *uc, const BufArg &a) {
    (void)uc;
    const halide_buffer_t *b = a.buf;
    const char *io = a.is_output ? "Output" : "Input";
    const int rank = rank_of(a.name, strlen(a.name));
    if (b->dimensions > 0 && b->dim[0].stride != 1) {
        record(4, rank, 0, 0, halide_error_code_constraint_violated, "Constraint violated: %s.stride.0 (%d) == 1 (1)",
               a.name, b->dim[0].stride);
    }
    int64_t total = 1;
    for (int d = 0; d < b->dimensions; d++) {
        if (b->dim[d].extent < 0) {
            record(5, rank, d, 1, halide_error_code_buffer_extents_negative,
                   "The extents for buffer %s dimension %d is negative (%d)", a.name, d, b->dim[d].extent);
        }
        int64_t span = (int64_t)b->dim[d].extent * (int64_t)b->dim[d].stride;
        if (span < 0) span = -span;
        if (span > 0x7fffffffLL) {
            record(6, rank, d, 0, halide_error_code_buffer_allocation_too_large,
                   "Total allocation for buffer %s %s is %lld, which exceeds the maximum size of %lld", io, a.name,
                   (long long)span, (long long)0x7fffffffLL);
        }
        total *= b->dim[d].extent;  // |total| <= 2^31 * 2^31 at d = 1, checked before it can grow further
        if (d > 0 && total > 0x7fffffffLL) {
            record(6, rank, d, 1, halide_error_code_buffer_extents_too_large,
                   "Product of extents for buffer %s %s is %lld, which exceeds the maximum size of %lld", io, a.name,
                   (long long)total, (long long)0x7fffffffLL);
            total = 1;  // keep the running product inside int64 for the remaining dimensions
        }
    }
    return 0;
}

int check_covers(void *uc, const BufArg &a, int d, int req_min, int req_extent) {
    (void)uc;
    const halide_dimension_t &dm = a.buf->dim[d];
    // in int64: a negative or huge extent must not wrap the comparison (the reference compares in int32 after its
    // own overflow asserts; the outcome for representable values is the same)
    int64_t req_max = (int64_t)req_min + req_extent - 1, have_max = (int64_t)dm.min + dm.extent - 1;
    if (req_min < dm.min || req_max > have_max) {
        const bool before = req_min < dm.min;
        record(5, rank_of(a.name, strlen(a.name)), d, 0, halide_error_code_access_out_of_bounds,
               "%s buffer %s is accessed at %lld, which is %s the %s (%lld) in dimension %d",
               a.is_output ? "Output" : "Input", a.name, (long long)(before ? req_min : req_max),
               before ? "before" : "beyond", before ? "min" : "max", (long long)(before ? dm.min : have_max), d);
    }
    return 0;
}

int check_equal(void *uc, const char *what, int val, const char *expect_what, int expect) {
    (void)uc;
    if (val != expect) {
        // what = "<buffer>.<stride|min|extent>.<dim>"
        const char *dot = strchr(what, '.');
        int rank = 15, dim = 0, kind = 3;
        if (dot) {
            rank = rank_of(what, (size_t)(dot - what));
            kind = dot[1] == 's' ? 0 : (dot[1] == 'm' ? 1 : 2);
            const char *dot2 = strchr(dot + 1, '.');
            if (dot2) dim = atoi(dot2 + 1);
        }
        record(4, rank, dim, kind, halide_error_code_constraint_violated, "Constraint violated: %s (%d) == %s (%d)", what,
               val, expect_what, expect);
    }
    return 0;
}

void answer_query(halide_buffer_t *buf, const int *mins, const int *extents) {
    if (!(buf->host == nullptr && buf->device == 0)) return;
    int stride = 1;
    for (int d = 0; d < buf->dimensions; d++) {
        buf->dim[d].min = mins[d];
        buf->dim[d].extent = extents[d];
        buf->dim[d].stride = stride;
        stride *= extents[d];
    }
}

// ------------------------------------------------------------------------------------------------
// devices, streams, workspaces, allocation cache
static thread_local int t_gpu_device = -1;
static thread_local hipStream_t t_stream_override = nullptr;

struct Arena {
    void *ptr = nullptr;
    size_t bytes = 0;
    std::recursive_mutex call_mu;  // serialises the ENQUEUE of whole pipeline calls on this (device, stream)
};

// A device allocation made by this runtime.  `last_stream` = the stream the most recent pipeline call, upload or
// copy that touched it was enqueued on: whoever touches it next from a DIFFERENT stream (or from the host) orders
// itself after everything enqueued on that stream so far (an event recorded "now" on last_stream is at or after the
// last use).  The reference keeps the same kind of record for its allocation cache only (src/runtime/cuda.cpp:667-675,
// :815 "Can only safely re-use on the same stream on which it was freed"); extending it to live buffers is what makes
// `set_stream(S); pipeline(); set_stream(NULL); copy_to_host()` and cross-stream chaining of pipelines well-defined.
struct Owned {
    size_t bytes = 0;
    hipStream_t last_stream = nullptr;
    uint64_t version = 0;  // changes whenever the contents may have changed (upload, use as an output, copy target)
};
static std::atomic<uint64_t> g_version_counter{0};
struct Cached {
    void *ptr = nullptr;
    hipStream_t stream = nullptr;  // stream the freed allocation was last used on
    hipEvent_t done = nullptr;     // recorded on `stream` at free time (null: never used on a stream)
};

struct DeviceState {
    bool inited = false;
    bool usable = false;
    hipStream_t stream = nullptr;
    std::map<hipStream_t, Arena> arenas;   // scratch per stream
    std::map<uint64_t, Owned> owned;       // base address -> record (ordered: crops resolve to the containing allocation)
    std::multimap<size_t, Cached> cache;   // free allocations kept for reuse
    size_t cache_bytes = 0;                // their total (bounded: cache_limit())
    hipEvent_t ring[64] = {};              // short-lived ordering events (record + wait back to back)
    unsigned ring_next = 0;
    std::vector<hipEvent_t> graveyard;     // cache events whose wait has been enqueued; destroyed lazily
};

static std::mutex g_mu;  // guards g_dev[*] bookkeeping (device_copy_mutex analogue, device_interface.cpp:28)
static DeviceState g_dev[64];
static std::atomic<int> g_reuse{1};
static int g_device_count = -1;

static int device_count_locked() {
    if (g_device_count < 0) {
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
        (void)hipGetLastError();
        g_device_count = n > 64 ? 64 : n;
    }
    return g_device_count;
}

static int pick_device() {
    if (t_gpu_device >= 0) return t_gpu_device;
    const char *e = getenv("HL_GPU_DEVICE");
    if (e && *e) return atoi(e);
    return 0;
}

int acquire_device(void *uc, DeviceCtx *ctx, bool lock) {
    if (lock) {  // pipeline entry: the recorded argument-check failures come first
        int r = checks_done(uc);
        if (r) return r;
    }
    int dev = pick_device();
    std::recursive_mutex *mu = nullptr;
    {
        std::lock_guard<std::mutex> guard(g_mu);
        int n = device_count_locked();
        if (n <= 0) {
            return report(uc, halide_error_code_gpu_device_error,
                          "hlmi: no HIP device is visible; the gfx950 kernels have no CPU fallback");
        }
        if (dev < 0 || dev >= n) {
            return report(uc, halide_error_code_gpu_device_error, "hlmi: GPU device %d requested, %d visible", dev, n);
        }
        DeviceState &st = g_dev[dev];
        if (!st.inited) {
            st.inited = true;
            hipDeviceProp_t prop;
            if (hipGetDeviceProperties(&prop, dev) == hipSuccess && strncmp(prop.gcnArchName, "gfx950", 6) == 0 &&
                hipSetDevice(dev) == hipSuccess &&
                hipStreamCreateWithFlags(&st.stream, hipStreamNonBlocking) == hipSuccess) {
                st.usable = true;
            }
        }
        if (!st.usable) {
            return report(uc, halide_error_code_gpu_device_error,
                          "hlmi: device %d is not a usable gfx950 (MI355X) device; kernels are built for gfx950 only",
                          dev);
        }
        ctx->device = dev;
        ctx->stream = t_stream_override ? t_stream_override : st.stream;
        if (lock) mu = &st.arenas[ctx->stream].call_mu;  // std::map: the node (and the mutex in it) never moves
    }
    if (mu) ctx->call_lock = std::unique_lock<std::recursive_mutex>(*mu);  // taken with g_mu released (lock order: call -> g_mu)
    HLMI_HIP(uc, hipSetDevice(dev));
    return 0;
}

int get_workspace(void *uc, const DeviceCtx &ctx, size_t bytes, void **ptr) {
    std::lock_guard<std::mutex> lock(g_mu);
    Arena &a = g_dev[ctx.device].arenas[ctx.stream];
    if (a.bytes < bytes) {
        if (a.ptr) {
            HLMI_HIP(uc, hipStreamSynchronize(ctx.stream));
            HLMI_HIP(uc, hipFree(a.ptr));
            a.ptr = nullptr;
            a.bytes = 0;
        }
        size_t want = (bytes + (1u << 20) - 1) & ~(size_t)((1u << 20) - 1);
        HLMI_HIP(uc, hipMalloc(&a.ptr, want));
        a.bytes = want;
    }
    *ptr = a.ptr;
    return 0;
}

// ---- stream ordering ---------------------------------------------------------------------------
// Everything enqueued on `producer` so far happens before whatever is enqueued on `consumer` from now on.  A stream
// that no longer exists (a caller destroyed it) has no pending work we could wait for: errors are dropped.
static void order_after_locked(int dev, hipStream_t consumer, hipStream_t producer) {
    if (!producer || producer == consumer) return;
    DeviceState &st = g_dev[dev];
    hipEvent_t &ev = st.ring[st.ring_next++ % 64];
    if (!ev && hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) {
        ev = nullptr;
        (void)hipGetLastError();
        (void)hipStreamSynchronize(producer);
        (void)hipGetLastError();
        return;
    }
    if (record_done(ev, producer) != hipSuccess || wait_done(consumer, ev) != hipSuccess) (void)hipGetLastError();
}

static void bury_locked(DeviceState &st, hipEvent_t ev) {
    if (!ev) return;
    st.graveyard.push_back(ev);
    if (st.graveyard.size() > 256) {
        for (size_t i = 0; i < 128; i++) {
            (void)hipEventSynchronize(st.graveyard[i]);
            (void)hipEventDestroy(st.graveyard[i]);
        }
        st.graveyard.erase(st.graveyard.begin(), st.graveyard.begin() + 128);
        (void)hipGetLastError();
    }
}

// the allocation of device `*dev_out` that contains `handle` (crops and slices point into their parent), or null
static Owned *find_owned_locked(uint64_t handle, int *dev_out) {
    int n = device_count_locked();
    for (int d = 0; d < n; d++) {
        auto &m = g_dev[d].owned;
        auto it = m.upper_bound(handle);
        if (it == m.begin()) continue;
        --it;
        if (handle >= it->first && handle < it->first + it->second.bytes) {
            if (dev_out) *dev_out = d;
            return &it->second;
        }
    }
    return nullptr;
}

// a pipeline call / copy on ctx.stream is about to touch `handle`
static int note_use(void *uc, const DeviceCtx &ctx, uint64_t handle, const char *name, bool written = false) {
    std::lock_guard<std::mutex> lock(g_mu);
    int dev;
    Owned *o = find_owned_locked(handle, &dev);
    if (!o) return 0;  // wrapped native memory: its owner orders the streams (as with halide_cuda_wrap_device_ptr)
    if (dev != ctx.device) {
        return report(uc, halide_error_code_incompatible_device_interface,
                      "Buffer %s lives on GPU device %d but this call runs on device %d (halide_set_gpu_device)", name, dev,
                      ctx.device);
    }
    order_after_locked(dev, ctx.stream, o->last_stream);
    o->last_stream = ctx.stream;
    if (written) o->version = ++g_version_counter;
    return 0;
}

static std::map<std::tuple<int, int, int, int>, hipStream_t> g_part_streams;   // (device, part, nparts, replica) -> stream (under g_mu)
static std::unordered_map<hipStream_t, int> g_part_cus;                    // frame-queue stream -> the CU share its launches are sized for

// compute units a launch on `stream` is sized for: a 1 / nparts share for a library-owned frame-queue stream, else the device's
int stream_cu_count(int device, hipStream_t stream) {
    {
        std::lock_guard<std::mutex> lock(g_mu);
        auto it = g_part_cus.find(stream);
        if (it != g_part_cus.end()) return it->second;
    }
    static const int share = [] { const char *e = getenv("HLMI_STREAM_SHARE"); return e && atoi(e) > 1 ? atoi(e) : 1; }();   // experiment: caller-made streams take the throughput geometry too
    static std::atomic<int> cached[64];
    int c = cached[device & 63].load();
    if (c <= 0) {
        if (hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || c <= 0) c = 256;
        cached[device & 63].store(c);
    }
    return stream ? c / share : c;
}

uint64_t buffer_version(const halide_buffer_t *buf) {
    if (!buf || !buf->device) return 0;
    std::lock_guard<std::mutex> lock(g_mu);
    Owned *o = find_owned_locked(buf->device, nullptr);
    return o ? o->version : 0;
}

// ---- raw allocation with cache ---------------------------------------------------------------
// *prior_life = true when the block is a cached one whose previous owner may still have work in flight: the caller's
// stream `for_stream` has been ordered behind it (stream order, or an event wait), nobody else has.
static int dev_alloc_locked(void *uc, int dev, size_t bytes, hipStream_t for_stream, void **out, bool *prior_life = nullptr) {
    DeviceState &st = g_dev[dev];
    if (prior_life) *prior_life = false;
    auto range = st.cache.equal_range(bytes);
    auto pick = range.second;
    for (auto it = range.first; it != range.second; ++it) {
        if (it->second.stream == for_stream || !it->second.done) {  // same stream: stream order is enough
            pick = it;
            break;
        }
        if (pick == range.second) pick = it;
    }
    if (pick != range.second) {
        Cached c = pick->second;
        st.cache_bytes -= min(st.cache_bytes, pick->first);
        st.cache.erase(pick);
        if (c.done) {
            if (c.stream != for_stream && wait_done(for_stream, c.done) != hipSuccess) {
                (void)hipGetLastError();
                (void)hipEventSynchronize(c.done);
            }
            bury_locked(st, c.done);
            if (prior_life) *prior_life = true;
        }
        *out = c.ptr;
        return 0;
    }
    hipError_t e = hipMalloc(out, bytes ? bytes : 1);
    if (e != hipSuccess) {
        // drop the cache and retry once (cuda.cpp does the same on OOM)
        (void)hipDeviceSynchronize();
        for (auto &kv : st.cache) {
            (void)hipFree(kv.second.ptr);
            if (kv.second.done) (void)hipEventDestroy(kv.second.done);
        }
        st.cache.clear();
        st.cache_bytes = 0;
        e = hipMalloc(out, bytes ? bytes : 1);
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return report(uc, halide_error_code_device_malloc_failed, "hlmi: hipMalloc(%zu) failed: %s", bytes,
                      hipGetErrorString(e));
    }
    return 0;
}

// bytes of freed allocations kept for reuse per device: HLMI_ALLOC_CACHE_MB, default 16 GiB (of 288)
static size_t cache_limit() {
    static const size_t lim = [] {
        const char *e = getenv("HLMI_ALLOC_CACHE_MB");
        const long mb = (e && *e) ? atol(e) : 16384;
        return (size_t)(mb < 0 ? 0 : mb) << 20;
    }();
    return lim;
}

static void dev_free_locked(int dev, void *base, const Owned &rec) {
    DeviceState &st = g_dev[dev];
    if (g_reuse.load()) {
        Cached c;
        c.ptr = base, c.stream = rec.last_stream;
        if (rec.last_stream) {
            if (hipEventCreateWithFlags(&c.done, hipEventDisableTiming) != hipSuccess ||
                record_done(c.done, rec.last_stream) != hipSuccess) {
                // the stream is gone (destroyed by its owner): nothing of ours can still be pending on it
                (void)hipGetLastError();
                if (c.done) (void)hipEventDestroy(c.done);
                c.done = nullptr;
            }
        }
        st.cache.emplace(rec.bytes, c);
        st.cache_bytes += rec.bytes;
        // The reference's pool (src/runtime/cuda.cpp, halide_reuse_device_allocations) is unbounded; a caller that never
        // repeats a size would make this one hoard the device.  Beyond the limit the largest blocks go back first.
        while (st.cache_bytes > cache_limit() && !st.cache.empty()) {
            auto big = std::prev(st.cache.end());
            if (big->second.done) {
                (void)hipEventSynchronize(big->second.done);
                (void)hipEventDestroy(big->second.done);
            }
            (void)hipFree(big->second.ptr);
            st.cache_bytes -= min(st.cache_bytes, big->first);
            st.cache.erase(big);
        }
    } else {
        (void)hipFree(base);
    }
}

static void purge_cache_locked(DeviceState &st) {
    for (auto &kv : st.cache) {
        (void)hipFree(kv.second.ptr);
        if (kv.second.done) (void)hipEventDestroy(kv.second.done);
    }
    st.cache.clear();
    st.cache_bytes = 0;
    for (hipEvent_t ev : st.graveyard) (void)hipEventDestroy(ev);
    st.graveyard.clear();
    (void)hipGetLastError();
}

// ---- helpers over halide_buffer_t ---------------------------------------------------------------
static ptrdiff_t begin_offset(const halide_buffer_t *b) {
    ptrdiff_t idx = 0;
    for (int i = 0; i < b->dimensions; i++) {
        if (b->dim[i].stride < 0) idx += (ptrdiff_t)b->dim[i].stride * (b->dim[i].extent - 1);
    }
    return idx;
}
static ptrdiff_t end_offset(const halide_buffer_t *b) {
    ptrdiff_t idx = 0;
    for (int i = 0; i < b->dimensions; i++) {
        if (b->dim[i].stride > 0) idx += (ptrdiff_t)b->dim[i].stride * (b->dim[i].extent - 1);
    }
    return idx + 1;
}
static size_t elem_bytes(const halide_buffer_t *b) { return (b->type.bits + 7) / 8; }
static bool is_empty(const halide_buffer_t *b) {
    for (int i = 0; i < b->dimensions; i++) {
        if (b->dim[i].extent <= 0) return true;
    }
    return false;
}

// validation order of device_interface.cpp:84-128
static int validate(void *uc, const halide_buffer_t *buf, const char *routine) {
    if (buf == nullptr) return report(uc, halide_error_code_buffer_is_null, "Buffer pointer passed to %s is null", routine);
    bool has_if = buf->device_interface != nullptr, has_dev = buf->device != 0;
    if (has_dev && !has_if) {
        return report(uc, halide_error_code_no_device_interface, "Buffer has a non-zero device but no device interface");
    }
    if (has_if && !has_dev) {
        return report(uc, halide_error_code_device_interface_no_device, "Buffer has a non-null device_interface but device is 0");
    }
    if ((buf->flags & halide_buffer_flag_host_dirty) && (buf->flags & halide_buffer_flag_device_dirty)) {
        return report(uc, halide_error_code_host_and_device_dirty, "Buffer has both host and device dirty bits set");
    }
    return 0;
}

// strided copy between host and device images of the same buffer (same strides on both sides).
// Contiguous runs are merged; the rest goes through hipMemcpy2DAsync row bundles.
static int copy_strided(void *uc, const halide_buffer_t *b, bool to_device, hipStream_t stream) {
    if (is_empty(b)) return 0;
    const size_t es = elem_bytes(b);
    // sort dims by |stride|
    int order[16], nd = b->dimensions;
    if (nd > 16) return report(uc, halide_error_code_unimplemented, "hlmi: more than 16 dimensions");
    for (int i = 0; i < nd; i++) order[i] = i;
    for (int i = 1; i < nd; i++) {
        for (int j = i; j > 0 && llabs((long long)b->dim[order[j]].stride) < llabs((long long)b->dim[order[j - 1]].stride); j--) {
            int t = order[j];
            order[j] = order[j - 1];
            order[j - 1] = t;
        }
    }
    for (int i = 0; i < nd; i++) {
        if (b->dim[i].stride < 0) return report(uc, halide_error_code_unimplemented, "hlmi: negative strides are not supported on the device");
    }
    // merge the contiguous inner run
    size_t run = 1;  // elements
    int k = 0;
    while (k < nd && (size_t)b->dim[order[k]].stride == run) {
        run *= (size_t)b->dim[order[k]].extent;
        k++;
    }
    uint8_t *h0 = b->host;
    uint8_t *d0 = reinterpret_cast<uint8_t *>((uintptr_t)b->device);
    hipMemcpyKind kind = to_device ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost;
    if (k == nd) {
        HLMI_HIP(uc, hipMemcpyAsync(to_device ? (void *)d0 : (void *)h0, to_device ? (void *)h0 : (void *)d0, run * es, kind, stream));
        return 0;
    }
    // dimension order[k] becomes the "rows" of a 2-D copy; iterate over the remaining dims
    const int rd = order[k];
    const size_t pitch = (size_t)b->dim[rd].stride * es, rows = (size_t)b->dim[rd].extent;
    if (pitch < run * es) {
        // Rows overlap (a row stride smaller than the row: e.g. tools/RunGen.h:787-805 adopts the dense strides a
        // bounds query proposes for the REQUIRED region while keeping its own larger extents).  Host and device images
        // share one layout, so for an upload the whole span can go as one block; a download goes row by row in
        // index order so that aliased elements end up with the value of their last row, as a host loop would leave them.
        if (to_device) {
            const size_t span = (size_t)(end_offset(b) - begin_offset(b)) * es;
            HLMI_HIP(uc, hipMemcpyAsync(d0, h0, span, kind, stream));
            return 0;
        }
        int idx[16] = {0};
        for (;;) {
            size_t off = 0;
            for (int j = k; j < nd; j++) off += (size_t)idx[j] * (size_t)b->dim[order[j]].stride * es;
            HLMI_HIP(uc, hipMemcpyAsync(h0 + off, d0 + off, run * es, kind, stream));
            int j = k;
            for (; j < nd; j++) {
                if (++idx[j] < b->dim[order[j]].extent) break;
                idx[j] = 0;
            }
            if (j >= nd) break;
        }
        return 0;
    }
    int idx[16] = {0};
    for (;;) {
        size_t off = 0;
        for (int j = k + 1; j < nd; j++) off += (size_t)idx[j] * (size_t)b->dim[order[j]].stride * es;
        uint8_t *h = h0 + off, *d = d0 + off;
        HLMI_HIP(uc, hipMemcpy2DAsync(to_device ? (void *)d : (void *)h, pitch, to_device ? (void *)h : (void *)d, pitch,
                                      run * es, rows, kind, stream));
        int j = k + 1;
        for (; j < nd; j++) {
            if (++idx[j] < b->dim[order[j]].extent) break;
            idx[j] = 0;
        }
        if (j >= nd) break;
    }
    return 0;
}

// ---- impl functions of the HIP interface -----------------------------------------------------
static int hip_device_malloc(void *uc, halide_buffer_t *buf) {
    if (buf->device) return 0;  // already allocated (cuda.cpp:601-606)
    DeviceCtx ctx;
    int r = acquire_device(uc, &ctx, false);
    if (r) return r;
    for (int i = 0; i < buf->dimensions; i++) {
        if (buf->dim[i].stride < 0) return report(uc, halide_error_code_unimplemented, "hlmi: negative strides are not supported on the device");
    }
    size_t bytes = (size_t)(end_offset(buf) - begin_offset(buf)) * elem_bytes(buf);
    if (is_empty(buf)) bytes = 0;
    bytes = (bytes + 255) & ~(size_t)255;  // kernels may over-read whole 16-byte vectors at row ends
    if (bytes == 0) bytes = 256;
    void *base = nullptr;
    std::lock_guard<std::mutex> lock(g_mu);
    bool prior_life = false;
    r = dev_alloc_locked(uc, ctx.device, bytes, ctx.stream, &base, &prior_life);
    if (r) return r;
    buf->device = (uint64_t)(uintptr_t)base;
    buf->device_interface = halide_hip_device_interface();
    Owned rec;
    rec.version = ++g_version_counter;
    // A reused block has been ordered behind its previous life on ctx.stream ONLY (stream order or the event wait above):
    // naming ctx.stream as its last stream makes a first use on any other stream order itself behind that too (note_use).
    rec.bytes = bytes, rec.last_stream = prior_life ? ctx.stream : nullptr;
    g_dev[ctx.device].owned[buf->device] = rec;
    return 0;
}

static int hip_device_free(void *uc, halide_buffer_t *buf) {
    (void)uc;
    if (buf->device == 0) return 0;
    {
        std::lock_guard<std::mutex> lock(g_mu);
        int n = device_count_locked();
        for (int dev = 0; dev < n; dev++) {
            auto it = g_dev[dev].owned.find(buf->device);
            if (it == g_dev[dev].owned.end()) continue;
            Owned rec = it->second;
            g_dev[dev].owned.erase(it);
            int cur = -1;
            (void)hipGetDevice(&cur);
            if (cur != dev) (void)hipSetDevice(dev);
            if (!g_reuse.load()) (void)hipDeviceSynchronize();
            dev_free_locked(dev, (void *)(uintptr_t)buf->device, rec);
            if (cur >= 0 && cur != dev) (void)hipSetDevice(cur);
            break;
        }
        // not owned: a wrapped native pointer or a crop — nothing to free
    }
    buf->device = 0;
    buf->device_interface = nullptr;
    buf->flags &= ~(uint64_t)halide_buffer_flag_device_dirty;
    return 0;
}

// where the work that last touched `buf` was enqueued: (device, stream).  Falls back to the calling thread's device
// and stream for memory this runtime does not own (wrapped pointers) and for buf == null.
static int producer_of(void *uc, const halide_buffer_t *buf, int *dev, hipStream_t *stream) {
    DeviceCtx ctx;
    int r = acquire_device(uc, &ctx, false);
    if (r) return r;
    *dev = ctx.device, *stream = ctx.stream;
    if (buf && buf->device) {
        std::lock_guard<std::mutex> lock(g_mu);
        int d;
        Owned *o = find_owned_locked(buf->device, &d);
        if (o) {
            *dev = d;
            if (o->last_stream) *stream = o->last_stream;
            else if (d != ctx.device) *stream = g_dev[d].stream;
        }
    }
    if (*dev != ctx.device) HLMI_HIP(uc, hipSetDevice(*dev));
    return 0;
}
struct RestoreDevice {  // a buffer that lives on another device than the calling thread's: switch back on scope exit
    int dev;
    ~RestoreDevice() { (void)hipSetDevice(dev); }
};

static int hip_device_sync(void *uc, halide_buffer_t *buf) {
    int dev;
    hipStream_t s;
    int r = producer_of(uc, buf, &dev, &s);
    if (r) return r;
    RestoreDevice back{pick_device()};
    hipError_t e = hipStreamSynchronize(s);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        // the producing stream was destroyed by its owner, which completes its work
        if (e == hipErrorContextIsDestroyed || e == hipErrorInvalidHandle || e == hipErrorInvalidResourceHandle) return 0;
        return hip_failed(uc, e, "hipStreamSynchronize");
    }
    return 0;
}

static int hip_device_release(void *uc) {
    std::lock_guard<std::mutex> lock(g_mu);
    int n = device_count_locked();
    for (int d = 0; d < n; d++) {
        DeviceState &st = g_dev[d];
        if (!st.inited || !st.usable) continue;
        (void)hipSetDevice(d);
        (void)hipDeviceSynchronize();
        purge_cache_locked(st);
        // The arena NODES stay: a thread entering a pipeline keeps a pointer to the call lock inside its node
        // (acquire_device), and get_workspace() re-allocates on demand.
        for (auto &kv : st.arenas) {
            if (kv.second.ptr) (void)hipFree(kv.second.ptr);
            kv.second.ptr = nullptr;
            kv.second.bytes = 0;
        }
    }
    (void)hipSetDevice(pick_device());
    (void)hipGetLastError();
    (void)uc;
    return 0;
}

static int hip_copy_to_host(void *uc, halide_buffer_t *buf) {
    // on the stream that produced the data (NOT necessarily the caller's current one), then wait for it
    int dev;
    hipStream_t s;
    int r = producer_of(uc, buf, &dev, &s);
    if (r) return r;
    RestoreDevice back{pick_device()};
    if (!stream_is_special(s)) {
        // the producing stream may have been destroyed by its owner (a torch stream that was garbage-collected): that
        // completes its work, and the download goes on the device's own stream instead (hip_device_sync tolerates the same)
        const hipError_t q = hipStreamQuery(s);
        if (q != hipSuccess && q != hipErrorNotReady) {
            (void)hipGetLastError();
            std::lock_guard<std::mutex> lock(g_mu);
            s = g_dev[dev].stream;
        }
    }
    r = copy_strided(uc, buf, false, s);
    if (r) return r;
    HLMI_HIP(uc, hipStreamSynchronize(s));
    return 0;
}

static int hip_copy_to_device(void *uc, halide_buffer_t *buf) {
    DeviceCtx ctx;
    int r = acquire_device(uc, &ctx, false);
    if (r) return r;
    if ((r = note_use(uc, ctx, buf->device, "passed to halide_copy_to_device", true))) return r;
    r = copy_strided(uc, buf, true, ctx.stream);
    if (r) return r;
    // the reference's copy_to_device is complete on return (the host buffer may be rewritten right away)
    HLMI_HIP(uc, hipStreamSynchronize(ctx.stream));
    return 0;
}

static int hip_wrap_native(void *uc, halide_buffer_t *buf, uint64_t handle) {
    if (buf->device != 0) {
        return report(uc, halide_error_code_device_wrap_native_failed, "hlmi: wrap_native on a buffer that already has a device allocation");
    }
    buf->device = handle;
    buf->device_interface = halide_hip_device_interface();
    return 0;
}

static int hip_detach_native(void *uc, halide_buffer_t *buf) {
    // Like halide_cuda_detach_device_ptr (src/runtime/cuda.cpp:1240-1249): forget the handle, free nothing.  Ownership
    // is the CALLER's bookkeeping (Halide::Runtime::Buffer tracks it as BufferDeviceOwnership::WrappedNative,
    // HalideBuffer.h:346-347); a handle is a plain device address here, so an alias that wraps an address inside one of
    // the runtime's own allocations is indistinguishable from that allocation and must be detachable.
    (void)uc;
    buf->device = 0;
    buf->device_interface = nullptr;
    return 0;
}

static int hip_device_crop(void *uc, const halide_buffer_t *src, halide_buffer_t *dst) {
    // dst already has the cropped dim[]; the device handle is the address of dst's min element
    (void)uc;
    ptrdiff_t off = 0;
    for (int i = 0; i < src->dimensions; i++) off += (ptrdiff_t)(dst->dim[i].min - src->dim[i].min) * src->dim[i].stride;
    dst->device = src->device + (uint64_t)(off * (ptrdiff_t)elem_bytes(src));
    dst->device_interface = src->device_interface;
    return 0;
}

static int hip_device_slice(void *uc, const halide_buffer_t *src, int slice_dim, int slice_pos, halide_buffer_t *dst) {
    (void)uc;
    ptrdiff_t off = (ptrdiff_t)(slice_pos - src->dim[slice_dim].min) * src->dim[slice_dim].stride;
    dst->device = src->device + (uint64_t)(off * (ptrdiff_t)elem_bytes(src));
    dst->device_interface = src->device_interface;
    return 0;
}

static int hip_device_release_crop(void *uc, halide_buffer_t *buf) {
    (void)uc;
    buf->device = 0;
    buf->device_interface = nullptr;
    return 0;
}

// ---- pipeline argument protocol -------------------------------------------------------------------
// true when the DMA engine may still be reading `p` after hipMemcpyAsync returns (pinned / registered host memory)
static bool host_memory_is_pinned(const void *p) {
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
        (void)hipGetLastError();
        return false;  // plain pageable memory is unknown to the runtime
    }
    return attr.type == hipMemoryTypeHost;
}

int input_to_device(void *uc, const DeviceCtx &ctx, const BufArg &a) {
    halide_buffer_t *b = a.buf;
    int r = validate(uc, b, a.name);
    if (r) return r;
    if (b->device && b->device_interface != halide_hip_device_interface()) {
        return report(uc, halide_error_code_incompatible_device_interface,
                      "Input buffer %s carries a device allocation of a different device interface", a.name);
    }
    bool fresh = false;
    if (b->device == 0) {
        if (b->host == nullptr) return report(uc, halide_error_code_host_is_null, "Input buffer %s host pointer is null", a.name);
        r = hip_device_malloc(uc, b);
        if (r) return r;
        fresh = true;
    }
    // order ctx.stream behind the stream that produced / last read this buffer
    const bool upload = (b->flags & halide_buffer_flag_host_dirty) || fresh;
    if ((r = note_use(uc, ctx, b->device, a.name, upload))) return r;
    if (upload) {
        if (b->host == nullptr) return report(uc, halide_error_code_host_is_null, "Input buffer %s host pointer is null", a.name);
        r = copy_strided(uc, b, true, ctx.stream);
        if (r) return halide_error_code_copy_to_device_failed;
        // Pageable host memory has been staged by the time the async copy returns; pinned or registered memory is read
        // by the DMA engine later, and the reference's copy_to_device is complete on return: wait for it then.
        if (host_memory_is_pinned(b->host)) HLMI_HIP(uc, hipStreamSynchronize(ctx.stream));
        b->flags &= ~(uint64_t)halide_buffer_flag_host_dirty;
    }
    return 0;
}

int output_on_device(void *uc, const DeviceCtx &ctx, const BufArg &a) {
    halide_buffer_t *b = a.buf;
    int r = validate(uc, b, a.name);
    if (r) return r;
    if (b->device && b->device_interface != halide_hip_device_interface()) {
        return report(uc, halide_error_code_incompatible_device_interface,
                      "Output buffer %s carries a device allocation of a different device interface", a.name);
    }
    if (b->device == 0) {
        r = hip_device_malloc(uc, b);
        if (r) return r;
    }
    return note_use(uc, ctx, b->device, a.name, true);
}

void mark_output_written(halide_buffer_t *b) {
    b->flags |= halide_buffer_flag_device_dirty;
    b->flags &= ~(uint64_t)halide_buffer_flag_host_dirty;
}

// ------------------------------------------------------------------------------------------------
// per-kernel timing
struct TimedLaunch {
    std::string name;
    hipEvent_t e0, e1;
    double alg_bytes;  // algorithmic bytes of this launch (0 = not declared by the pipeline)
};
static thread_local double t_next_alg_bytes = 0;
void timing_note_bytes(double bytes) { t_next_alg_bytes = bytes; }
static std::atomic<int> g_timing{0};
static std::mutex g_timing_mu;
static std::vector<TimedLaunch> g_launches;
static thread_local hipEvent_t t_pending_e1;

bool env_flag(const char *name) {
    const char *e = getenv(name);
    return e && *e && atoi(e) != 0;
}

bool timing_enabled() { return g_timing.load(std::memory_order_relaxed) != 0; }

static std::atomic<int> g_only_active{0};
static std::string g_only_name;   // written under g_timing_mu, read only while g_only_active != 0
bool launch_selected(const char *name) {
    // timing experiments only (WRONG results): HLMI_SKIP_LAUNCH=name[,name..] leaves those launches out — what a launch costs the
    // frame rate is the frame rate without it
    static const std::string skip = [] { const char *e = getenv("HLMI_SKIP_LAUNCH"); return std::string(e ? e : ""); }();
    if (!skip.empty()) {
        const size_t n = strlen(name);
        for (size_t pos = 0; (pos = skip.find(name, pos)) != std::string::npos; pos += n) {
            const bool b0 = pos == 0 || skip[pos - 1] == ',', b1 = pos + n == skip.size() || skip[pos + n] == ',';
            if (b0 && b1) return false;
        }
    }
    if (g_only_active.load(std::memory_order_acquire) == 0) return true;
    std::lock_guard<std::mutex> lock(g_timing_mu);
    return g_only_name == name;
}

void timing_begin(const char *name, hipStream_t s) {
    TimedLaunch t;
    t.name = name;
    t.alg_bytes = t_next_alg_bytes;
    t_next_alg_bytes = 0;
    if (hipEventCreate(&t.e0) != hipSuccess || hipEventCreate(&t.e1) != hipSuccess) return;
    (void)hipEventRecord(t.e0, event_stream(s));
    t_pending_e1 = t.e1;
    std::lock_guard<std::mutex> lock(g_timing_mu);
    g_launches.push_back(t);
}

void timing_end(hipStream_t s) {
    if (t_pending_e1) (void)hipEventRecord(t_pending_e1, event_stream(s));
    t_pending_e1 = nullptr;
}

}  // namespace hlmi

// ====================================================================================================
// exported C ABI
using namespace hlmi;

extern "C" {

void halide_error(void *uc, const char *msg) { g_error_handler.load()(uc, msg); }
halide_error_handler_t halide_set_error_handler(halide_error_handler_t h) {
    return g_error_handler.exchange(h ? h : default_error_handler);
}
void halide_print(void *uc, const char *msg) { g_print.load()(uc, msg); }
halide_print_t halide_set_custom_print(halide_print_t p) { return g_print.exchange(p ? p : default_print); }
// the defaults stay reachable for hooks that wrap them (tools/RunGenMain.cpp:220,243 — its allocation tracker)
void *halide_default_malloc(void *uc, size_t x) { return default_malloc(uc, x); }
void halide_default_free(void *uc, void *p) { default_free(uc, p); }
// dlsym / dlopen hooks (src/runtime/posix_get_symbol.cpp): RunGenMain.cpp:531 probes for runtime symbols through them
static std::atomic<halide_get_symbol_t> g_get_symbol{nullptr};
static std::atomic<halide_load_library_t> g_load_library{nullptr};
static std::atomic<halide_get_library_symbol_t> g_get_library_symbol{nullptr};
void *halide_default_get_symbol(const char *name) { return dlsym(RTLD_DEFAULT, name); }
void *halide_default_load_library(const char *name) { return dlopen(name, RTLD_LAZY); }
void *halide_default_get_library_symbol(void *lib, const char *name) { return dlsym(lib, name); }
void *halide_get_symbol(const char *name) {
    halide_get_symbol_t f = g_get_symbol.load();
    return f ? f(name) : halide_default_get_symbol(name);
}
void *halide_load_library(const char *name) {
    halide_load_library_t f = g_load_library.load();
    return f ? f(name) : halide_default_load_library(name);
}
void *halide_get_library_symbol(void *lib, const char *name) {
    halide_get_library_symbol_t f = g_get_library_symbol.load();
    return f ? f(lib, name) : halide_default_get_library_symbol(lib, name);
}
halide_get_symbol_t halide_set_custom_get_symbol(halide_get_symbol_t f) {
    halide_get_symbol_t old = g_get_symbol.exchange(f);
    return old ? old : halide_default_get_symbol;
}
halide_load_library_t halide_set_custom_load_library(halide_load_library_t f) {
    halide_load_library_t old = g_load_library.exchange(f);
    return old ? old : halide_default_load_library;
}
halide_get_library_symbol_t halide_set_custom_get_library_symbol(halide_get_library_symbol_t f) {
    halide_get_library_symbol_t old = g_get_library_symbol.exchange(f);
    return old ? old : halide_default_get_library_symbol;
}
void *halide_malloc(void *uc, size_t x) { return g_malloc.load()(uc, x); }
void halide_free(void *uc, void *p) { g_free.load()(uc, p); }
halide_malloc_t halide_set_custom_malloc(halide_malloc_t m) { return g_malloc.exchange(m ? m : default_malloc); }
halide_free_t halide_set_custom_free(halide_free_t f) { return g_free.exchange(f ? f : default_free); }

// ---- generic wrappers: validation + dispatch through buf->device_interface (device_interface.cpp) ----
static const halide_device_interface_t *resolve(const halide_buffer_t *buf, const halide_device_interface_t *di) {
    return di ? di : buf->device_interface;
}

int halide_device_malloc(void *uc, halide_buffer_t *buf, const halide_device_interface_t *di) {
    int r = validate(uc, buf, "halide_device_malloc");
    if (r) return r;
    if (buf->device_interface && di && buf->device_interface != di) {
        return report(uc, halide_error_code_incompatible_device_interface, "halide_device_malloc doesn't support switching interfaces");
    }
    di = resolve(buf, di);
    if (!di) return report(uc, halide_error_code_no_device_interface, "halide_device_malloc: no device interface");
    return di->device_malloc(uc, buf, di);
}

int halide_device_free(void *uc, halide_buffer_t *buf) {
    int r = validate(uc, buf, "halide_device_free");
    if (r) return r;
    if (buf->device_interface) return buf->device_interface->device_free(uc, buf);
    buf->flags &= ~(uint64_t)halide_buffer_flag_device_dirty;
    return 0;
}

int halide_device_sync(void *uc, halide_buffer_t *buf) {
    int r = validate(uc, buf, "halide_device_sync");
    if (r) return r;
    if (!buf->device_interface) return report(uc, halide_error_code_no_device_interface, "halide_device_sync: buffer has no device interface");
    return buf->device_interface->device_sync(uc, buf);
}

int halide_device_sync_global(void *uc, const halide_device_interface_t *di) {
    if (!di) return halide_error_code_no_device_interface;
    return di->device_sync(uc, nullptr);
}

int halide_copy_to_host(void *uc, halide_buffer_t *buf) {
    int r = validate(uc, buf, "halide_copy_to_host");
    if (r) return r;
    if (!(buf->flags & halide_buffer_flag_device_dirty)) return 0;  // nothing to do (device_interface.cpp:48-50)
    if (!buf->device_interface) return report(uc, halide_error_code_no_device_interface, "halide_copy_to_host: device dirty but no device interface");
    return buf->device_interface->copy_to_host(uc, buf);
}

int halide_copy_to_device(void *uc, halide_buffer_t *buf, const halide_device_interface_t *di) {
    int r = validate(uc, buf, "halide_copy_to_device");
    if (r) return r;
    di = resolve(buf, di);
    if (!di) return report(uc, halide_error_code_no_device_interface, "halide_copy_to_device: no device interface");
    if (buf->device && buf->device_interface != di) {
        return report(uc, halide_error_code_incompatible_device_interface, "halide_copy_to_device does not support switching interfaces");
    }
    return di->copy_to_device(uc, buf, di);
}

int halide_device_and_host_malloc(void *uc, halide_buffer_t *buf, const halide_device_interface_t *di) {
    int r = validate(uc, buf, "halide_device_and_host_malloc");
    if (r) return r;
    di = resolve(buf, di);
    if (!di) return report(uc, halide_error_code_no_device_interface, "halide_device_and_host_malloc: no device interface");
    return di->device_and_host_malloc(uc, buf, di);
}

int halide_device_and_host_free(void *uc, halide_buffer_t *buf) {
    int r = validate(uc, buf, "halide_device_and_host_free");
    if (r) return r;
    if (buf->device_interface) return buf->device_interface->device_and_host_free(uc, buf);
    if (buf->host) {
        halide_free(uc, buf->host);
        buf->host = nullptr;
    }
    buf->flags &= ~(uint64_t)halide_buffer_flag_device_dirty;
    return 0;
}

int halide_buffer_copy(void *uc, halide_buffer_t *src, const halide_device_interface_t *dst_di, halide_buffer_t *dst) {
    const halide_device_interface_t *di = dst_di ? dst_di : src->device_interface;
    if (!di) di = halide_hip_device_interface();
    return di->buffer_copy(uc, src, dst_di, dst);
}

int halide_device_wrap_native(void *uc, halide_buffer_t *buf, uint64_t handle, const halide_device_interface_t *di) {
    int r = validate(uc, buf, "halide_device_wrap_native");
    if (r) return r;
    if (!di) return report(uc, halide_error_code_no_device_interface, "halide_device_wrap_native: no device interface");
    return di->wrap_native(uc, buf, handle, di);
}

int halide_device_detach_native(void *uc, halide_buffer_t *buf) {
    int r = validate(uc, buf, "halide_device_detach_native");
    if (r) return r;
    if (!buf->device_interface) return 0;
    return buf->device_interface->detach_native(uc, buf);
}

void halide_device_release(void *uc, const halide_device_interface_t *di) {
    if (di) di->device_release(uc, di);
}

int halide_reuse_device_allocations(void *uc, bool flag) {
    g_reuse.store(flag ? 1 : 0);
    if (!flag) return halide_hip_release_unused_device_allocations(uc);
    return 0;
}

// ---- the interface table ---------------------------------------------------------------------------
static int if_device_malloc(void *uc, halide_buffer_t *buf, const halide_device_interface_t *) {
    int r = hip_device_malloc(uc, buf);
    return r ? (r == halide_error_code_gpu_device_error ? r : halide_error_code_device_malloc_failed) : 0;
}
static int if_device_free(void *uc, halide_buffer_t *buf) { return hip_device_free(uc, buf); }
static int if_device_sync(void *uc, halide_buffer_t *buf) {
    return hip_device_sync(uc, buf) ? halide_error_code_device_sync_failed : 0;
}
static void if_device_release(void *uc, const halide_device_interface_t *) { (void)hip_device_release(uc); }
static int if_copy_to_host(void *uc, halide_buffer_t *buf) {
    // device_interface.cpp:43-80
    if (!(buf->flags & halide_buffer_flag_device_dirty)) return 0;
    if (buf->host == nullptr) return report(uc, halide_error_code_host_is_null, "copy_to_host: host pointer is null");
    if (buf->device == 0) return report(uc, halide_error_code_no_device_interface, "copy_to_host: device dirty without a device allocation");
    if (hip_copy_to_host(uc, buf)) return halide_error_code_copy_to_host_failed;
    buf->flags &= ~(uint64_t)halide_buffer_flag_device_dirty;
    return 0;
}
static int if_copy_to_device(void *uc, halide_buffer_t *buf, const halide_device_interface_t *di) {
    // device_interface.cpp:155-205
    if (buf->device == 0) {
        int r = if_device_malloc(uc, buf, di);
        if (r) return r;
    }
    if (buf->flags & halide_buffer_flag_host_dirty) {
        if (buf->flags & halide_buffer_flag_device_dirty) return halide_error_code_copy_to_device_failed;
        if (buf->host == nullptr) return report(uc, halide_error_code_host_is_null, "copy_to_device: host pointer is null");
        if (hip_copy_to_device(uc, buf)) return halide_error_code_copy_to_device_failed;
        buf->flags &= ~(uint64_t)halide_buffer_flag_host_dirty;
    }
    return 0;
}
static int if_device_and_host_malloc(void *uc, halide_buffer_t *buf, const halide_device_interface_t *di) {
    // default implementation of the reference (device_interface.cpp:390-420): separate host + device allocations
    size_t bytes = (size_t)(end_offset(buf) - begin_offset(buf)) * elem_bytes(buf);
    buf->host = (uint8_t *)halide_malloc(uc, bytes ? bytes : 1);
    if (!buf->host) return halide_error_code_out_of_memory;
    int r = if_device_malloc(uc, buf, di);
    if (r) {
        halide_free(uc, buf->host);
        buf->host = nullptr;
    }
    return r;
}
static int if_device_and_host_free(void *uc, halide_buffer_t *buf) {
    int r = hip_device_free(uc, buf);
    if (buf->host) {
        halide_free(uc, buf->host);
        buf->host = nullptr;
    }
    buf->flags &= ~(uint64_t)(halide_buffer_flag_host_dirty | halide_buffer_flag_device_dirty);
    return r;
}
static int if_buffer_copy(void *uc, halide_buffer_t *src, const halide_device_interface_t *dst_di, halide_buffer_t *dst) {
    // Supported: same-shape copies host<->device / device<->device / host<->host over the overlap of
    // src and dst (device_buffer_utils.h semantics for the cases the drivers use).
    if (src->dimensions != dst->dimensions || buf_type_abi(src) != buf_type_abi(dst)) {
        return report(uc, halide_error_code_device_buffer_copy_failed, "buffer_copy: type/dimension mismatch");
    }
    bool from_host = !(src->device && (src->flags & halide_buffer_flag_device_dirty)) && src->host;
    bool to_host = dst_di == nullptr;
    if (!to_host && dst->device == 0) {
        int r = if_device_malloc(uc, dst, dst_di);
        if (r) return r;
    }
    if (!from_host && src->device == 0) return report(uc, halide_error_code_device_buffer_copy_failed, "buffer_copy: source has no data");
    if (to_host && dst->host == nullptr) return report(uc, halide_error_code_host_is_null, "buffer_copy: destination host is null");
    DeviceCtx ctx;
    int r = acquire_device(uc, &ctx, false);
    if (r) return r;
    if (!from_host && (r = note_use(uc, ctx, src->device, "src of halide_buffer_copy"))) return r;
    if (!to_host && (r = note_use(uc, ctx, dst->device, "dst of halide_buffer_copy", true))) return r;
    // element-run copies over the overlapping box
    int nd = src->dimensions;
    int lo[16], ext[16];
    for (int i = 0; i < nd; i++) {
        int a = src->dim[i].min > dst->dim[i].min ? src->dim[i].min : dst->dim[i].min;
        int s1 = src->dim[i].min + src->dim[i].extent, d1 = dst->dim[i].min + dst->dim[i].extent;
        int bmax = s1 < d1 ? s1 : d1;
        lo[i] = a;
        ext[i] = bmax - a;
        if (ext[i] <= 0) return 0;
    }
    const size_t es = elem_bytes(src);
    uint8_t *sbase = from_host ? src->host : (uint8_t *)(uintptr_t)src->device;
    uint8_t *dbase = to_host ? dst->host : (uint8_t *)(uintptr_t)dst->device;
    hipMemcpyKind kind = from_host ? (to_host ? hipMemcpyHostToHost : hipMemcpyHostToDevice)
                                   : (to_host ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice);
    int idx[16] = {0};
    bool rows_contig = nd >= 1 && src->dim[0].stride == 1 && dst->dim[0].stride == 1;
    for (;;) {
        ptrdiff_t so = 0, doff = 0;
        for (int i = 0; i < nd; i++) {
            so += (ptrdiff_t)(lo[i] + idx[i] - src->dim[i].min) * src->dim[i].stride;
            doff += (ptrdiff_t)(lo[i] + idx[i] - dst->dim[i].min) * dst->dim[i].stride;
        }
        size_t n = rows_contig ? (size_t)ext[0] : 1;
        HLMI_HIP(uc, hipMemcpyAsync(dbase + doff * (ptrdiff_t)es, sbase + so * (ptrdiff_t)es, n * es, kind, ctx.stream));
        int j = rows_contig ? 1 : 0;
        for (; j < nd; j++) {
            if (++idx[j] < ext[j]) break;
            idx[j] = 0;
        }
        if (j >= nd) break;
    }
    HLMI_HIP(uc, hipStreamSynchronize(ctx.stream));
    if (to_host) {
        dst->flags |= halide_buffer_flag_host_dirty;
    } else {
        dst->flags |= halide_buffer_flag_device_dirty;
        dst->flags &= ~(uint64_t)halide_buffer_flag_host_dirty;
    }
    return 0;
}
static int if_device_crop(void *uc, const halide_buffer_t *src, halide_buffer_t *dst) { return hip_device_crop(uc, src, dst); }
static int if_device_slice(void *uc, const halide_buffer_t *src, int d, int pos, halide_buffer_t *dst) {
    return hip_device_slice(uc, src, d, pos, dst);
}
static int if_device_release_crop(void *uc, halide_buffer_t *buf) { return hip_device_release_crop(uc, buf); }
static int if_wrap_native(void *uc, halide_buffer_t *buf, uint64_t handle, const halide_device_interface_t *) {
    return hip_wrap_native(uc, buf, handle);
}
static int if_detach_native(void *uc, halide_buffer_t *buf) { return hip_detach_native(uc, buf); }
static int if_compute_capability(void *uc, int *major, int *minor) {
    (void)uc;
    *major = 9;  // gfx950
    *minor = 5;
    return 0;
}

static const halide_device_interface_t g_hip_interface = {
    if_device_malloc, if_device_free, if_device_sync, if_device_release, if_copy_to_host, if_copy_to_device,
    if_device_and_host_malloc, if_device_and_host_free, if_buffer_copy, if_device_crop, if_device_slice,
    if_device_release_crop, if_wrap_native, if_detach_native, if_compute_capability, nullptr};

const halide_device_interface_t *halide_hip_device_interface(void) { return &g_hip_interface; }

int halide_hip_wrap_device_ptr(void *uc, halide_buffer_t *buf, uint64_t device_ptr) {
    return halide_device_wrap_native(uc, buf, device_ptr, &g_hip_interface);
}
int halide_hip_detach_device_ptr(void *uc, halide_buffer_t *buf) {
    if (buf->device_interface && buf->device_interface != &g_hip_interface) {
        return report(uc, halide_error_code_incompatible_device_interface, "halide_hip_detach_device_ptr: not a HIP buffer");
    }
    return halide_device_detach_native(uc, buf);
}
uintptr_t halide_hip_get_device_ptr(void *uc, halide_buffer_t *buf) {
    (void)uc;
    if (buf->device == 0 || buf->device_interface != &g_hip_interface) return 0;
    return (uintptr_t)buf->device;
}
int halide_hip_release_unused_device_allocations(void *uc) {
    (void)uc;
    std::lock_guard<std::mutex> lock(g_mu);
    int n = device_count_locked();
    for (int d = 0; d < n; d++) {
        DeviceState &st = g_dev[d];
        if (st.cache.empty()) continue;
        (void)hipSetDevice(d);
        (void)hipDeviceSynchronize();
        purge_cache_locked(st);
    }
    (void)hipSetDevice(pick_device());
    (void)hipGetLastError();
    return 0;
}

int hlmi_device_count(void) {
    std::lock_guard<std::mutex> lock(g_mu);
    int n = device_count_locked(), usable = 0;
    for (int d = 0; d < n; d++) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, d) == hipSuccess && strncmp(prop.gcnArchName, "gfx950", 6) == 0) usable++;
    }
    (void)hipGetLastError();
    return usable;
}

void halide_set_gpu_device(int n) { t_gpu_device = n; }
int halide_get_gpu_device(void *) { return pick_device(); }
void halide_hip_set_stream(void *stream) { t_stream_override = (hipStream_t)stream; }
// One of `nparts` library-owned streams for frames in flight, each with a hardware queue of its own (a stream made by
// hipExtStreamCreateWithCUMask never shares its HSA queue; plain hipStreamCreate streams are multiplexed onto a small pool,
// which is what made four of them 20 % slower: 89 against 108-112 Gpx/s on local_laplacian 4K).  Launches on such a stream
// size their grids for a 1 / nparts share of the device (stream_cu_count), i.e. for throughput with nparts frames in
// flight rather than for the latency of one call.
//
// What the CU mask really does (round 6, scripts/ubench/cu_mask_probe.hip, profiles/r06_cu_mask_probe.txt): bit b names
// compute unit b / 8 of XCD b % 8, and an XCD whose share of the mask is EMPTY runs the queue on ALL of its CUs.  Rounds
// 3-5 set every nparts-th bit — for nparts = 4 all CUs of two XCDs and none of the other six — so what they called "four
// 64-CU partitions" were four queues on the whole device.  Real partitions (HLMI_PART_MASK=1: the same CU slots on every
// XCD, 2: contiguous slots) measure 3 % SLOWER than the unmasked queues (108.7 against 111.8 Gpx/s), so the default mask
// is now explicitly the full one (layout 3); 0 is the old layout, kept for A/B.
// Streams are created once per (device, part, nparts, replica) and owned by the library.  Returns NULL on failure.
void *halide_hip_partition_stream(int part, int nparts) { return halide_hip_partition_stream_replica(part, nparts, 0); }

// Further queues with the same mask and geometry share (replica 0, 1, ..).
void *halide_hip_partition_stream_replica(int part, int nparts, int replica) {
    DeviceCtx ctx;
    if (nparts < 1 || part < 0 || part >= nparts || replica < 0 || replica > 15 || acquire_device(nullptr, &ctx, false)) return nullptr;
    std::lock_guard<std::mutex> lock(g_mu);
    auto &streams = g_part_streams;
    auto key = std::make_tuple(ctx.device, part, nparts, replica);
    auto it = streams.find(key);
    if (it != streams.end()) return (void *)it->second;
    int ncu = 0;
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, ctx.device) != hipSuccess || ncu < nparts) return nullptr;
    std::vector<uint32_t> mask((size_t)(ncu + 31) / 32, 0u);
    const char *ml = getenv("HLMI_PART_MASK");
    const int layout = ml && *ml ? atoi(ml) : 3;
    const int nxcc = ncu >= 64 ? ncu / 32 : 1, slots = ncu / nxcc;
    int mine = 0;
    for (int b = 0; b < ncu; b++) {
        const int slot = b / nxcc;
        const bool on = layout == 1 ? (slot % nparts == part) : layout == 2 ? (slot * nparts / slots == part) : layout == 3 ? true : (b % nparts == part);
        if (on) mask[(size_t)b / 32] |= 1u << (b % 32), mine++;
    }
    if (mine == 0) return nullptr;   // (a real partition of more parts than an XCD has CU slots: an empty mask would mean the whole device)
    hipStream_t s = nullptr;
    if (hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    streams[key] = s;
    const char *gc = getenv("HLMI_PART_GEOM_CUS");   // experiment: the CU count the launch geometry is sized for
    g_part_cus[s] = gc && atoi(gc) > 0 ? atoi(gc) : layout == 3 ? ncu / nparts : mine;
    return (void *)s;
}


void *halide_hip_get_stream(void *uc) {
    DeviceCtx ctx;
    if (acquire_device(uc, &ctx, false)) return nullptr;
    return (void *)ctx.stream;
}

void hlmi_kernel_timing_enable(int on) { g_timing.store(on ? 1 : 0); }
void hlmi_kernel_timing_only(const char *name) {
    std::lock_guard<std::mutex> lock(g_timing_mu);
    g_only_name = name ? name : "";
    g_only_active.store(name && *name ? 1 : 0, std::memory_order_release);
}
void hlmi_kernel_timing_reset(void) {
    std::lock_guard<std::mutex> lock(g_timing_mu);
    for (auto &t : g_launches) {
        (void)hipEventDestroy(t.e0);
        (void)hipEventDestroy(t.e1);
    }
    g_launches.clear();
}
size_t hlmi_kernel_timing_report(char *out, size_t cap) {
    std::lock_guard<std::mutex> lock(g_timing_mu);
    struct Agg {
        int calls = 0;
        double ms = 0, bytes = 0;
    };
    std::vector<std::string> order;
    std::unordered_map<std::string, Agg> agg;
    for (auto &t : g_launches) {
        (void)hipEventSynchronize(t.e1);
        float ms = 0;
        if (hipEventElapsedTime(&ms, t.e0, t.e1) != hipSuccess) continue;
        if (!agg.count(t.name)) order.push_back(t.name);
        Agg &a = agg[t.name];
        a.calls++;
        a.ms += ms;
        a.bytes += t.alg_bytes;
    }
    std::string s = "[";
    for (size_t i = 0; i < order.size(); i++) {
        const Agg &a = agg[order[i]];
        char line[256];
        snprintf(line, sizeof line,
                 "%s{\"name\":\"%s\",\"calls\":%d,\"total_ms\":%.6f,\"avg_ms\":%.6f,\"alg_bytes\":%.0f}", i ? "," : "",
                 order[i].c_str(), a.calls, a.ms, a.ms / (a.calls ? a.calls : 1), a.bytes / (a.calls ? a.calls : 1));
        s += line;
    }
    s += "]";
    if (out && cap) {
        size_t n = s.size() < cap - 1 ? s.size() : cap - 1;
        memcpy(out, s.data(), n);
        out[n] = 0;
    }
    return s.size() + 1;
}

const char *hlmi_version(void) { return "hlmi 0.1 gfx950"; }
int hlmi_canon_fma(void) { return HLMI_CANON_FMA; }

}  // extern "C"
