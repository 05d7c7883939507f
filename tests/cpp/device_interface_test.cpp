// device_interface_test.cpp — TEST INFRASTRUCTURE (own text; built by oracle/ref.mk against the REFERENCE's
// src/runtime/HalideBuffer.h + HalideRuntime.h, so Halide::Runtime::Buffer here is the reference's class).
//
// Drives the parts of this library's halide_device_interface_t that no app driver touches, the way a user of
// Halide::Runtime::Buffer reaches them (src/runtime/HalideBuffer.h:1465-1562 cropped / :1660-1709 sliced ->
// device_crop / device_slice / device_release_crop; :1881-1912 device_and_host_malloc / _free; halide_buffer_copy,
// src/runtime/HalideRuntime.h:958-1011; halide_device_wrap_native / detach_native), plus the stream-ordering
// guarantees of the HIP runtime slice (a result produced on one stream read back or consumed on another; an
// allocation freed on one stream and reused on another — the rule of src/runtime/cuda.cpp:667-675, :815).
// The pipeline is halide_blur (the one pipeline whose result is pinned by the reference's own scalar loop,
// apps/blur/test.cpp:18-33, restated in blur_ref below).  Prints "Success!" or the failed checks.
#include "HalideBuffer.h"
#include "HalideRuntime.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "halide_blur.h"
#include "stencil_chain.h"

using Halide::Runtime::Buffer;

namespace {

int failures = 0;
char last_message[1024];

void quiet_handler(void *, const char *msg) { strncpy(last_message, msg, sizeof last_message - 1); }

#define CHECK(cond, ...)                          \
    do {                                          \
        if (!(cond)) {                            \
            printf("FAIL %s:%d: ", __FILE__, __LINE__); \
            printf(__VA_ARGS__);                  \
            printf(" (%s)\n", last_message);      \
            failures++;                           \
        }                                         \
    } while (0)

// out(x, y) = ((in(x,y)+in(x+1,y)+in(x+2,y))/3 + ... rows y, y+1, y+2 ...)/3 in uint16 arithmetic
uint16_t blur_ref(const Buffer<uint16_t, 2> &in, int x, int y) {
    uint16_t r[3];
    for (int j = 0; j < 3; j++) r[j] = (uint16_t)((uint16_t)(in(x, y + j) + in(x + 1, y + j) + in(x + 2, y + j)) / 3);
    return (uint16_t)((uint16_t)(r[0] + r[1] + r[2]) / 3);
}

void fill(Buffer<uint16_t, 2> &b, unsigned seed) {
    b.for_each_element([&](int x, int y) { b(x, y) = (uint16_t)((x * 131u + y * 977u + seed * 7919u) ^ (x * y)); });
    b.set_host_dirty();
}

int count_blur_mismatches(const Buffer<uint16_t, 2> &in, Buffer<uint16_t, 2> &out) {
    int bad = 0;
    out.for_each_element([&](int x, int y) { bad += out(x, y) != blur_ref(in, x, y); });
    return bad;
}

}  // namespace

int main() {
    halide_set_error_handler(&quiet_handler);
    const halide_device_interface_t *hip = halide_hip_device_interface();
    const int W = 200, H = 120;

    Buffer<uint16_t, 2> in(W + 2, H + 2), truth(W + 2, H + 2);
    fill(in, 1);
    truth.copy_from(in);
    truth.set_host_dirty(false);

    // ---- plain call, result through Buffer::copy_to_host -------------------------------------------------------
    {
        Buffer<uint16_t, 2> out(W, H);
        CHECK(halide_blur(in, out) == 0, "halide_blur");
        CHECK(out.device_dirty() && out.has_device_allocation(), "output left device-dirty");
        CHECK(out.copy_to_host() == 0, "copy_to_host");
        CHECK(count_blur_mismatches(truth, out) == 0, "blur result");
    }
    // From here on the HOST copy of `in` is deliberately stale (zeros, not marked dirty): whatever reads the right
    // values reads them from the device allocation.
    CHECK(in.has_device_allocation() && !in.host_dirty(), "input resident");
    memset(in.data(), 0, in.size_in_bytes());

    // ---- device_crop of an input (HalideBuffer.h:1501-1517) -----------------------------------------------------
    {
        const int cx = 8, cy = 4, cw = 64, ch = 40;
        Buffer<uint16_t, 2> crop = in.cropped({{cx, cw + 2}, {cy, ch + 2}});
        CHECK(crop.has_device_allocation(), "crop has a device handle");
        CHECK(crop.raw_buffer()->device == in.raw_buffer()->device + (uint64_t)(cy * in.dim(1).stride() + cx) * 2, "crop handle = address of its min element");
        Buffer<uint16_t, 2> out(cw, ch);
        out.set_min(cx, cy);
        CHECK(halide_blur(crop, out) == 0, "halide_blur on a device crop");
        out.copy_to_host();
        int bad = 0;
        out.for_each_element([&](int x, int y) { bad += out(x, y) != blur_ref(truth, x, y); });
        CHECK(bad == 0, "blur of a device crop: %d wrong", bad);
    }   // ~crop -> device_release_crop; the parent allocation must survive
    CHECK(in.has_device_allocation(), "parent allocation survives its crop");

    // ---- device_crop of an OUTPUT: the pipeline writes into the middle of a larger device image ----------------
    {
        Buffer<uint16_t, 2> big(W, H);
        big.fill(0xabcd);
        big.set_host_dirty();
        CHECK(big.copy_to_device(hip) == 0, "copy_to_device");
        const int cx = 16, cy = 10, cw = 96, ch = 50;
        {
            Buffer<uint16_t, 2> window = big.cropped({{cx, cw}, {cy, ch}});
            CHECK(halide_blur(in, window) == 0, "halide_blur into a device crop");
            CHECK(window.device_dirty(), "crop marked device-dirty");
        }
        big.set_device_dirty();   // the parent learns about writes through a crop from its owner (HalideBuffer.h:1455-1463)
        CHECK(big.copy_to_host() == 0, "copy_to_host of the parent");
        int bad = 0;
        big.for_each_element([&](int x, int y) {
            const bool inside = x >= cx && x < cx + cw && y >= cy && y < cy + ch;
            bad += big(x, y) != (inside ? blur_ref(truth, x, y) : 0xabcd);
        });
        CHECK(bad == 0, "write through a device crop: %d wrong", bad);
    }

    // ---- device_slice (HalideBuffer.h:1660-1675) ------------------------------------------------------------------
    {
        Buffer<uint16_t, 3> planes(W + 2, H + 2, 3);
        planes.for_each_element([&](int x, int y, int c) { planes(x, y, c) = c == 1 ? truth(x, y) : (uint16_t)(x + c); });
        planes.set_host_dirty();
        CHECK(planes.copy_to_device(hip) == 0, "copy_to_device 3-D");
        memset(planes.data(), 0, planes.size_in_bytes());
        Buffer<uint16_t> plane1 = planes.sliced(2, 1);
        CHECK(plane1.dimensions() == 2 && plane1.has_device_allocation(), "slice is 2-D with a device handle");
        Buffer<uint16_t, 2> out(W, H);
        CHECK(halide_blur(plane1, out) == 0, "halide_blur on a device slice");
        out.copy_to_host();
        CHECK(count_blur_mismatches(truth, out) == 0, "blur of a device slice");
    }

    // ---- halide_buffer_copy: host -> device, device -> device (shifted boxes), device -> host --------------------
    {
        Buffer<uint16_t, 2> a(W, H), b(W, H), c(W, H);
        fill(a, 5);
        Buffer<uint16_t, 2> a_truth = a.copy();
        a_truth.set_host_dirty(false);
        b.fill(0), c.fill(0);
        b.set_min(8, 4);   // overlaps a on [8, W) x [4, H)
        CHECK(halide_buffer_copy(nullptr, a.raw_buffer(), hip, b.raw_buffer()) == 0, "buffer_copy host -> device");
        CHECK(b.has_device_allocation() && b.device_dirty(), "destination allocated and device-dirty");
        Buffer<uint16_t, 2> d(W, H);
        d.fill(0);
        d.set_host_dirty();
        CHECK(d.copy_to_device(hip) == 0, "copy_to_device");
        CHECK(halide_buffer_copy(nullptr, b.raw_buffer(), hip, d.raw_buffer()) == 0, "buffer_copy device -> device");
        d.set_device_dirty();
        CHECK(halide_buffer_copy(nullptr, d.raw_buffer(), nullptr, c.raw_buffer()) == 0, "buffer_copy device -> host");
        int bad = 0;
        c.for_each_element([&](int x, int y) { bad += c(x, y) != ((x >= 8 && y >= 4) ? a_truth(x, y) : 0); });
        CHECK(bad == 0, "buffer_copy chain: %d wrong", bad);
    }

    // ---- device_and_host_malloc / device_and_host_free --------------------------------------------------------------
    {
        halide_dimension_t shape[2] = {{0, W + 2, 1}, {0, H + 2, W + 2}};
        halide_buffer_t raw = {0};
        raw.type = halide_type_of<uint16_t>();
        raw.dim = shape, raw.dimensions = 2;
        CHECK(hip->device_and_host_malloc(nullptr, &raw, hip) == 0, "device_and_host_malloc");
        CHECK(raw.host != nullptr && raw.device != 0 && raw.device_interface == hip, "both allocations present");
        if (raw.host) {
            memcpy(raw.host, truth.data(), truth.size_in_bytes());
            raw.set_host_dirty();
            Buffer<uint16_t, 2> out(W, H);
            CHECK(halide_blur(&raw, out) == 0, "halide_blur on a device_and_host buffer");
            out.copy_to_host();
            CHECK(count_blur_mismatches(truth, out) == 0, "blur of a device_and_host buffer");
        }
        CHECK(hip->device_and_host_free(nullptr, &raw) == 0 && raw.host == nullptr && raw.device == 0, "device_and_host_free");
    }

    // ---- wrap_native / detach_native ------------------------------------------------------------------------------------
    {
        halide_dimension_t shape[2] = {{0, W + 2, 1}, {0, H + 2, W + 2}};
        halide_buffer_t alias = {0};
        alias.type = halide_type_of<uint16_t>();
        alias.dim = shape, alias.dimensions = 2;
        const uint64_t ptr = (uint64_t)halide_hip_get_device_ptr(nullptr, in.raw_buffer());
        CHECK(ptr != 0, "halide_hip_get_device_ptr");
        CHECK(halide_hip_wrap_device_ptr(nullptr, &alias, ptr) == 0, "wrap_device_ptr");
        CHECK(halide_hip_wrap_device_ptr(nullptr, &alias, ptr) == halide_error_code_device_wrap_native_failed, "double wrap is refused");
        Buffer<uint16_t, 2> out(W, H);
        CHECK(halide_blur(&alias, out) == 0, "halide_blur on a wrapped pointer (host null, device set)");
        out.copy_to_host();
        CHECK(count_blur_mismatches(truth, out) == 0, "blur of a wrapped pointer");
        CHECK(halide_hip_detach_device_ptr(nullptr, &alias) == 0 && alias.device == 0, "detach");
        CHECK(in.has_device_allocation(), "detaching an alias leaves the allocation it pointed into alone");
        Buffer<uint16_t, 2> again(W, H);
        CHECK(halide_blur(in, again) == 0, "the aliased allocation is still usable");
        again.copy_to_host();
        CHECK(count_blur_mismatches(truth, again) == 0, "blur after the alias was detached");
    }

    // ---- stream ordering ------------------------------------------------------------------------------------------------
    {
        void *s1 = halide_hip_partition_stream(0, 2), *s2 = halide_hip_partition_stream(1, 2);
        CHECK(s1 && s2 && s1 != s2, "two partition streams");
        const int BW = 1536, BH = 2560;
        Buffer<uint16_t, 2> src(BW, BH), mid(BW, BH), dst(BW, BH);
        src.for_each_element([&](int x, int y) { src(x, y) = (uint16_t)(x * 3 + y * 5); });
        src.set_host_dirty();
        // (a) produced on s1, read back with the thread's stream reset: copy_to_host must follow the PRODUCER
        halide_hip_set_stream(s1);
        CHECK(stencil_chain(src, mid) == 0, "stencil_chain on s1");
        halide_hip_set_stream(nullptr);
        CHECK(mid.copy_to_host() == 0, "copy_to_host after set_stream(NULL)");
        Buffer<uint16_t, 2> mid_truth = mid.copy();
        // (b) chain across streams through a device-resident buffer: s1 -> s2
        Buffer<uint16_t, 2> mid2(BW, BH), want(BW, BH);
        halide_hip_set_stream(s1);
        CHECK(stencil_chain(src, mid2) == 0, "first stage on s1");
        halide_hip_set_stream(s2);
        CHECK(stencil_chain(mid2, dst) == 0, "second stage on s2");
        halide_hip_set_stream(nullptr);
        dst.copy_to_host();
        mid_truth.set_host_dirty();
        CHECK(stencil_chain(mid_truth, want) == 0, "reference chain on one stream");
        want.copy_to_host();
        int bad = 0;
        dst.for_each_element([&](int x, int y) { bad += dst(x, y) != want(x, y); });
        CHECK(bad == 0, "cross-stream chain: %d wrong", bad);
        // (c) an allocation freed while its stream is still busy, reused from another stream
        for (int rep = 0; rep < 3; rep++) {
            halide_hip_set_stream(s1);
            {
                Buffer<uint16_t, 2> scratch(BW, BH);
                for (int i = 0; i < 12; i++) CHECK(stencil_chain(src, scratch) == 0, "busy work on s1");
                scratch.device_free();   // back to the cache with s1 still writing into it
            }
            halide_hip_set_stream(s2);
            Buffer<uint16_t, 2> reuse(BW, BH), out2(BW, BH);   // same size: takes the cached allocation
            reuse.copy_from(mid_truth);
            reuse.set_host_dirty();
            CHECK(stencil_chain(reuse, out2) == 0, "consumer on s2");
            halide_hip_set_stream(nullptr);
            out2.copy_to_host();
            bad = 0;
            out2.for_each_element([&](int x, int y) { bad += out2(x, y) != want(x, y); });
            CHECK(bad == 0, "reuse of a freed allocation across streams (rep %d): %d wrong", rep, bad);
        }
    }

    in.device_free();
    halide_device_release(nullptr, hip);
    if (failures) return 1;
    printf("Success!\n");
    return 0;
}
