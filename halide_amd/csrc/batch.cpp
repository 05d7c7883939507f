// batch.cpp — hlmi_run_batch: the in-process frame sharder.
//
// Frames are independent units (every AOT entry point is a pure function of its own buffers) and the path has no
// exchange step, so a batch spreads over the GPUs of a node with no data-path collective: one HOST THREAD and one HIP
// stream per worker, every worker bound to its device with halide_set_gpu_device() — the per-thread device selection the
// reference's GPU runtimes offer for exactly this (src/runtime/HalideRuntime.h:1014-1019; the calling pattern is that
// of test/generator/gpu_multi_context_threaded_aottest.cpp: one thread per context, the same pipeline in each).
// The one-process-per-GPU launch (bench.py under torch.distributed.run, RCCL only for the timing barrier) stays the
// multi-process form of the same sharding; this is the form a C++ host application links.
#include <atomic>
#include <map>
#include <mutex>
#include <thread>
#include <vector>

#include "hlmi_internal.h"

namespace {

std::mutex g_batch_mu;
std::map<std::pair<int, int>, hipStream_t> g_worker_streams;  // (device, worker slot on that device) -> stream

// Library-owned stream of worker `slot` on `device`.  With `parts` > 1 the slot's stream is frame queue slot % parts
// (a hardware queue of its own, launches sized for frames in flight: halide_hip_partition_stream).
hipStream_t worker_stream(int device, int slot, int parts) {
    if (parts > 1) return (hipStream_t)halide_hip_partition_stream(slot % parts, parts);
    std::lock_guard<std::mutex> lock(g_batch_mu);
    auto key = std::make_pair(device, slot);
    auto it = g_worker_streams.find(key);
    if (it != g_worker_streams.end()) return it->second;
    hipStream_t s = nullptr;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    g_worker_streams[key] = s;
    return s;
}

}  // namespace

extern "C" int hlmi_run_batch(hlmi_argv_fn fn, void ***frame_args, int n_frames, const int *devices, int n_devices,
                              int streams_per_device) {
    if (!fn || n_frames < 0 || (n_frames > 0 && !frame_args) || n_devices < 1 || !devices) {
        return hlmi::report(nullptr, halide_error_code_generic_error, "hlmi_run_batch: bad arguments");
    }
    if (streams_per_device < 1) streams_per_device = 1;
    const int n_workers = n_devices * streams_per_device;
    std::atomic<int> first_error{0};
    std::vector<std::thread> threads;
    // slot of a worker = how many earlier workers use the same device (a device may be listed more than once)
    std::vector<int> slot((size_t)n_workers, 0);
    for (int w = 0; w < n_workers; w++) {
        for (int v = 0; v < w; v++) slot[(size_t)w] += devices[v % n_devices] == devices[w % n_devices];
    }
    for (int w = 0; w < n_workers && w < n_frames; w++) {
        threads.emplace_back([=, &first_error, &slot] {
            const int device = devices[w % n_devices];
            halide_set_gpu_device(device);
            int parts = 0;
            for (int v = 0; v < n_workers; v++) parts += devices[v % n_devices] == device;
            hipStream_t s = worker_stream(device, slot[(size_t)w], streams_per_device > 1 ? parts : 1);
            int r = 0;
            if (!s) {
                r = hlmi::report(nullptr, halide_error_code_gpu_device_error, "hlmi_run_batch: no stream on device %d", device);
            } else {
                halide_hip_set_stream(s);
                for (int i = w; i < n_frames && r == 0 && first_error.load() == 0; i += n_workers) r = fn(frame_args[i]);
                // the batch is complete when hlmi_run_batch returns: results are valid on their devices
                if (hipStreamSynchronize(s) != hipSuccess && r == 0) r = halide_error_code_device_sync_failed;
                halide_hip_set_stream(nullptr);
            }
            halide_set_gpu_device(-1);
            int expected = 0;
            if (r) first_error.compare_exchange_strong(expected, r);
        });
    }
    for (auto &t : threads) t.join();
    return first_error.load();
}
