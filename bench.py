#!/usr/bin/env python3
"""bench.py — throughput of the hot path on MI355X.

Workload (BASELINE.json configs[2], the one `metric` is quoted on): apps/local_laplacian, 8 pyramid
levels (J=8), levels=8, alpha=1/7, beta=1, uint16 RGB planar 3840x2160 in/out, fp32 internal arithmetic.
Input of the headline `value`: UNIFORM FULL-RANGE NOISE — what the reference's own benchmark protocol feeds the pipeline
(tools/RunGen.h:482-505 fills inputs with uniform random values; apps/local_laplacian's CMake test runs the driver on whatever
image it is handed); the smooth synthetic frame and the reference checkout's natural image, tiled, are reported beside it
(`value_smooth`, `value_natural_tiled`).
A "step" = PASSES (default 16) passes of the pipeline over a batch of FRAMES_PER_STEP distinct synthetic frames per GPU
(128 frames = 1.06 Gpx per GPU and step, so that the driver's `--steps 20` times >= 0.2 s), called through the C ABI
(libhlmi.so, `local_laplacian(halide_buffer_t*, ...)`), inputs and outputs resident in HBM (uploaded once before the
timed region, like apps/local_laplacian/process.cpp:31-39 where the first call pays the copies and `benchmark()` times
calls + device_sync).

    python bench.py [--gpus N] [--steps K] [--warmup W]

`--gpus N` is the whole command for any N: when the script is not already running under a launcher it starts N ranks of
itself under `python -m torch.distributed.run` (one process per GPU, rendezvous on 127.0.0.1) and refuses when fewer than
N devices are visible (halide_amd/launcher.py).  The driver's own
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` works the same; a WORLD_SIZE that disagrees
with --gpus is an error, so `n_gpus` in the output is always the N asked for.

Frames are independent units: ranks shard them with no data-path collective ("scaling": "weak");
torch.distributed (RCCL) is used only for the barrier and the max-over-ranks of the elapsed time.
Prints ONE JSON line on rank 0.  After the timed region rank 0 (N = 1 only) also measures the other BASELINE.json
configs — bilateral_grid 1920x1080, nl_means 1920x1080x3, conv_layer bf16 N=16 — into `config.other_configs`, each with its
roofline and a bounded oracle `cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W, H, C = 3840, 2160, 3
LEVELS, ALPHA, BETA = 8, 1.0 / 7.0, 1.0
FRAMES_PER_STEP = int(os.environ.get("HLMI_BENCH_FRAMES", "8"))  # distinct frames per GPU and pass (8 x 99.5 MB of u16 I/O > 256 MB MALL)
PASSES = int(os.environ.get("HLMI_BENCH_PASSES", "16"))           # passes over those frames per step
HBM_PEAK_GBS = 8000.0         # MI355X_MICROARCH.md: 8.0 TB/s spec
ALG_BYTES_PER_PX = 12         # SURVEY.md §8(d) primary figure: 6 B read + 6 B written per pixel


HEADLINE_KIND = "noise"


def synth_frame(seed, w=W, h=H, kind="smooth"):
    """SURVEY.md §8d input variants: (i) uniform full-range noise — the HEADLINE input since round 6: what the reference's RunGen
    feeds a benchmark run, tools/RunGen.h:482-505, and the worst case for anything data dependent —, (ii) a smooth natural-like
    frame + mild noise and (iii) the one natural image of the reference
    checkout (apps/images/rgb_small.png, 192x320 8-bit RGB: tests/golden/rgb_small_u8.npz, made by scripts/make_golden.py),
    widened to 16 bits the way the reference's image loader does (x * 257) and tiled to the frame size, shifted per frame."""
    import numpy as np
    rng = np.random.default_rng(seed)
    if kind == "noise":
        return rng.integers(0, 65536, (C, h, w), dtype=np.uint16)
    if kind == "natural":
        small = np.load(os.path.join(ROOT, "tests", "golden", "rgb_small_u8.npz"))["rgb"]   # (3, 320, 192) u8
        tile = small.astype(np.uint16) * 257
        reps = (1, -(-h // tile.shape[1]) + 1, -(-w // tile.shape[2]) + 1)
        big = np.tile(tile, reps)
        oy, ox = (37 * seed) % tile.shape[1], (53 * seed) % tile.shape[2]
        return np.ascontiguousarray(big[:, oy:oy + h, ox:ox + w])
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    base = (np.sin(xx / 311.0 + seed) + np.cos(yy / 173.0) + np.sin((xx + yy) / 97.0) + 3.3) / 6.6
    img = np.stack([base * 65535.0, np.roll(base, 64, 1) * 52000.0, base[::-1] * 46000.0])
    img += rng.normal(0.0, 900.0, img.shape).astype(np.float32)
    return np.clip(img, 0, 65535).astype(np.uint16)


class ClockSampler:
    """Shader clock (MHz) and socket power (W) of the device while a leg runs, read from the amdgpu driver's hwmon files every
    20 ms on a thread (on a multi-GPU node: the k-th card that has the files, for local rank k): latency-shaped figures (one call + sync) move with the box's clock state, and a reader must be able to tell
    a regression from a box that was not clocked up.  Every field is None where the files are absent."""

    def __init__(self, device_index=0):
        import glob
        self.freq, self.power = None, None
        cards = sorted(glob.glob("/sys/class/drm/card[0-9]*/device/hwmon/hwmon*"))
        cards = [c for c in cards if os.path.exists(os.path.join(c, "freq1_input"))]
        if cards:
            h = cards[min(device_index, len(cards) - 1)]
            self.freq = os.path.join(h, "freq1_input")
            for name in ("power1_average", "power1_input"):
                if os.path.exists(os.path.join(h, name)):
                    self.power = os.path.join(h, name)
                    break
        self.samples = []

    @staticmethod
    def _read(path):
        try:
            with open(path) as f:
                return float(f.read().strip())
        except (OSError, ValueError, TypeError):
            return None

    def __enter__(self):
        import threading
        self.samples, self._stop = [], threading.Event()

        def loop():
            while not self._stop.is_set():
                self.samples.append((self._read(self.freq) if self.freq else None, self._read(self.power) if self.power else None))
                self._stop.wait(0.02)
        self._t = threading.Thread(target=loop, daemon=True)
        self._t.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        self._t.join()
        return False

    def summary(self):
        import statistics
        f = [a / 1e6 for a, _ in self.samples if a]
        p = [b / 1e6 for _, b in self.samples if b]
        # freq1_input is the INSTANTANEOUS shader clock of the device (it reads ~100 MHz whenever a sample falls between two kernels of a
        # call + sync pattern, scripts/clock_files_probe.sh): the maximum says what the clock reaches under the pattern, the median
        # how much of the time the device was clocked up at all
        return {"sclk_mhz_median": round(statistics.median(f)) if f else None, "sclk_mhz_min": round(min(f)) if f else None,
                "sclk_mhz_max": round(max(f)) if f else None,
                "power_w_median": round(statistics.median(p)) if p else None, "power_w_max": round(max(p)) if p else None,
                "samples": len(self.samples)}


def cpu_baseline(frame):
    """The tuned CPU evaluation of the oracle's algorithm (oracle/local_laplacian_fast_oracle.c: the oracle's operations
    in the oracle's order — tests/test_local_laplacian.py pins the two bit for bit — with level 0 of the processed pyramid
    computed at its consumer, the colour stage fused, buffers kept between calls; OpenMP over all host threads) timed on
    a bounded sample of the same workload.  kind "port": NOT Halide's own x86 schedule, which cannot be built here (the
    generator quotes 184 Mpx/s on 16 cores for it, local_laplacian_generator.cpp:139-140)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib  # cpu_baseline leg only
    import halide_amd
    oracle_lib.set_canon(halide_amd.canon_fma())   # the CPU evaluates the same canonical float form the library computes
    ncpu = os.cpu_count() or 1
    oracle_lib.local_laplacian(frame, LEVELS, ALPHA, BETA)       # the plain oracle, before the thread count is touched
    plain0 = time.perf_counter()
    oracle_lib.local_laplacian(frame, LEVELS, ALPHA, BETA)
    plain = time.perf_counter() - plain0
    oracle_lib.local_laplacian_fast(frame, LEVELS, ALPHA, BETA)  # warm (first touch of the arena)
    # the thread count that runs fastest on this host (the pyramid's small levels do not feed 256 threads)
    best_t, best_dt = ncpu, None
    for t in sorted({t for t in (8, 16, 32, 64, 128, ncpu) if t <= ncpu}):
        oracle_lib.set_threads(t)
        oracle_lib.local_laplacian_fast(frame, LEVELS, ALPHA, BETA)
        t0 = time.perf_counter()
        for _ in range(3):
            oracle_lib.local_laplacian_fast(frame, LEVELS, ALPHA, BETA)
        d = (time.perf_counter() - t0) / 3
        if best_dt is None or d < best_dt:
            best_t, best_dt = t, d
    oracle_lib.set_threads(best_t)
    n, t0 = 0, time.perf_counter()
    while True:
        oracle_lib.local_laplacian_fast(frame, LEVELS, ALPHA, BETA)
        n += 1
        dt = time.perf_counter() - t0
        if dt > 10.0 or n >= 256:
            break
    oracle_lib.set_threads(0)
    return {"value": round(n * W * H / dt / 1e6, 3), "unit": "Mpx/s", "cores": best_t, "kind": "port",
            "sample": f"{n} frames of {W}x{H} u16 RGB in {dt:.2f} s (tuned CPU evaluation, OpenMP; {best_t} threads = the fastest of "
                      f"8..{ncpu} on this host; the plain oracle on all {ncpu}: {W * H / plain / 1e6:.1f} Mpx/s)"}


def stub_main(args, rank, local_rank, world):
    """Launcher self-test (HLMI_BENCH_STUB=1, tests/test_launcher.py): the rendezvous, barrier, max-over-ranks and JSON path
    of the real run with the pipeline call replaced by a no-op and gloo instead of RCCL.  Not a measurement: "data": "stub"."""
    from halide_amd import launcher, sharding
    dist, device, ranks_seen = launcher.init_process_group(local_rank, world)
    mine = sharding.shard(world * FRAMES_PER_STEP, rank, world)

    def barrier():
        if dist is not None:
            dist.barrier()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(1e-4 * len(mine))
    barrier()
    elapsed = sharding.max_over_ranks(time.perf_counter() - t0, dist, device)
    # who is where: (rank, local rank = the HIP device the real run binds, OMP threads of the CPU legs) of every rank, on rank 0
    me = [rank, local_rank, int(os.environ.get("OMP_NUM_THREADS", "0"))]
    table = [me]
    if dist is not None:
        table = [None] * world
        dist.all_gather_object(table, me)
    if rank == 0:
        print(json.dumps({"metric": "launcher self-test (no pipeline ran)", "value": 0.0, "unit": "Mpx/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / max(1, args.steps) * 1e3, 4),
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "none", "data": "stub",
                          "config": {"workload": "stub", "rccl_ranks": ranks_seen, "backend": launcher.backend(),
                                     "frames_of_rank0": mine, "rank_table": table}}), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20, help="timed steps (a step = --passes passes over the frames: 20 steps ≈ 0.27 s of GPU time)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--passes", type=int, default=PASSES, help="passes over the FRAMES_PER_STEP distinct frames that make one step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the bilateral_grid / nl_means / conv_layer_bf16 leg")
    ap.add_argument("--no-ceiling", action="store_true", help="skip the live HBM copy-ceiling sweep (profiler runs)")
    ap.add_argument("--no-variants", action="store_true", help="skip the extra input variants (uniform noise, 7680x4320)")
    ap.add_argument("--partitions", type=int, default=int(os.environ.get("HLMI_BENCH_PARTITIONS", "4")),
                    help="frames of a step are spread over this many frame-queue streams (halide_hip_partition_stream: library "
                         "streams with a hardware queue each, launches sized for that many frames in flight; NOT CU "
                         "partitions - profiles/NOTES.md round 6), one frame per queue at a time: frames are independent units, "
                         "so the latency-bound coarse pyramid levels of one frame run beside the large kernels of the others; "
                         "0 = plain streams (--streams)")
    ap.add_argument("--streams-per-partition", type=int, default=int(os.environ.get("HLMI_BENCH_SPP", "1")),
                    help="replicas of each frame queue (more queues of the same kind)")
    ap.add_argument("--streams", type=int, default=int(os.environ.get("HLMI_BENCH_STREAMS", "2")),
                    help="with --partitions 0: plain HIP streams the frames of a step are spread over")
    args = ap.parse_args()

    from halide_amd import launcher
    launcher.ensure_ranks(args.gpus, os.path.abspath(__file__), sys.argv[1:])   # N > 1 and no launcher yet: start N ranks, exit
    rank, local_rank, world = launcher.check_world(args.gpus)
    if launcher.stub_mode():
        return stub_main(args, rank, local_rank, world)

    import numpy as np
    import torch
    import halide_amd as hl

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU fallback for the product path)")
    torch.cuda.set_device(local_rank)
    hl.set_gpu_device(local_rank)
    # one process per GPU over RCCL; rccl_ranks = an all-reduced count of the ranks (proof the backend spans `world` GPUs)
    dist, _, rccl_ranks = launcher.init_process_group(local_rank, world)

    # --- synthetic frames, resident in HBM before the timed region
    from halide_amd import sharding
    # the step's batch is world * FRAMES_PER_STEP frames; rank r owns frames r, r+world, ... (weak scaling)
    mine = sharding.shard(world * FRAMES_PER_STEP, rank, world)
    frames = [synth_frame(i, kind=HEADLINE_KIND) for i in mine]
    ins = [hl.Buffer(f) for f in frames]
    outs = [hl.Buffer(np.zeros_like(f)) for f in frames]
    for a, o in zip(ins, outs):  # first call uploads the input and allocates the output on the device
        hl.local_laplacian(a, LEVELS, ALPHA, BETA, o)
    outs[-1].device_sync()

    def run_variant(w, h, kind, nframes, steps):
        """Mpx/s of the same call on another input variant of configs[2], same stream scheduling, inputs
        resident; untimed with respect to the headline figure (runs after it)."""
        fr = [synth_frame(100 + i, w, h, kind) for i in range(nframes)]
        vi, vo = [hl.Buffer(f) for f in fr], [hl.Buffer(np.zeros_like(f)) for f in fr]

        def vstep():
            for i, (a, o) in enumerate(zip(vi, vo)):
                if streams:
                    hl.set_stream(streams[i % len(streams)])
                hl.local_laplacian(a, LEVELS, ALPHA, BETA, o)
            if streams:
                hl.set_stream(None)
        for _ in range(3):
            vstep()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(steps):
            vstep()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        for b in vi + vo:
            b.device_free()
        return round(nframes * steps * w * h / dt / 1e6, 1)

    def one_call_protocol(kind, samples=10):
        """The reference driver's own timing (apps/local_laplacian/process.cpp:36-39 = `benchmark(timing, 1, ...)` of
        tools/halide_benchmark.h:82-95, with the `10` samples its CMakeLists.txt:52 passes): ONE call + output.device_sync() per
        sample on the device's own stream, nothing else in flight, the minimum over the samples.  A latency, not a throughput."""
        f = synth_frame(200, W, H, kind)
        a, o = hl.Buffer(f), hl.Buffer(np.zeros_like(f))
        hl.local_laplacian(a, LEVELS, ALPHA, BETA, o)     # first call: upload + allocation, as in process.cpp:31
        o.device_sync()
        best = float("inf")
        for _ in range(samples):
            t = time.perf_counter()
            hl.local_laplacian(a, LEVELS, ALPHA, BETA, o)
            o.device_sync()
            best = min(best, time.perf_counter() - t)
        # the clock / power state under this call pattern: the same call + sync repeated for 0.3 s (untimed) while the sampler reads
        with ClockSampler(local_rank) as cs:
            t_end = time.perf_counter() + 0.3
            while time.perf_counter() < t_end:
                hl.local_laplacian(a, LEVELS, ALPHA, BETA, o)
                o.device_sync()
        a.device_free()
        o.device_free()
        return best, cs.summary()

    streams, keep, mode = [], [], "1 stream"
    if args.partitions > 1:
        spp = max(1, args.streams_per_partition)
        streams = [hl.partition_stream(p, args.partitions, r) for r in range(spp) for p in range(args.partitions)]
        mode = f"{args.partitions} frame-queue streams" + (f" x {spp}" if spp > 1 else "")
        if not all(streams):
            streams = []            # the device refused a CU mask: plain streams instead
    if not streams and args.streams > 1:
        keep = [torch.cuda.Stream() for _ in range(args.streams)]
        streams, mode = [s.cuda_stream for s in keep], f"{args.streams} streams"

    def one_pass(use_streams=True):
        for i, (a, o) in enumerate(zip(ins, outs)):
            if streams and use_streams:  # frame i goes to stream i % n
                hl.set_stream(streams[i % len(streams)])
            hl.local_laplacian(a, LEVELS, ALPHA, BETA, o)
        if streams and use_streams:
            hl.set_stream(None)

    def step():
        for _ in range(args.passes):
            one_pass()

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    sampler = ClockSampler(local_rank)
    with sampler:
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        elapsed = time.perf_counter() - t0
    elapsed = sharding.max_over_ranks(elapsed, dist, "cuda")

    # --- per-kernel durations: HIP events around every launch on the launch stream, in a separate untimed pass on ONE
    #     stream (kernels of different frames must not overlap while a single kernel is being timed)
    hl.kernel_timing_reset()
    hl.kernel_timing(True)
    for _ in range(3):
        one_pass(use_streams=False)
    torch.cuda.synchronize()
    hl.kernel_timing(False)
    kernels = hl.kernel_timing_report()
    hl.kernel_timing_reset()

    # --- the dominant kernel ALONE, back to back: the event brackets above include the gap to the neighbouring launches of the chain
    #     (they read 10-25 % longer than rocprofv3's kernel trace), so the duration `roofline.frac` is computed from is taken the way a
    #     profiler sees it: every other launch of the chain skipped (hlmi_kernel_timing_only), REPS calls enqueued on the device's
    #     stream without a sync in between, HIP events on that stream around the whole run, three rounds, the minimum.  The kernel's
    #     inputs are the planes the complete calls above left in the stream's workspace (same geometry, same data).
    dom_name = max(kernels, key=lambda k: k["avg_ms"])["name"]
    REPS = 100
    hip = hl.hip_runtime()
    import ctypes
    ev0, ev1 = ctypes.c_void_p(), ctypes.c_void_p()
    hip.hipEventCreate(ctypes.byref(ev0)), hip.hipEventCreate(ctypes.byref(ev1))
    lib_stream = ctypes.c_void_p(hl.lib.halide_hip_get_stream(None))
    hl.kernel_timing_only(dom_name)
    try:
        back_to_back_ms = None
        for _ in range(3):
            hl.local_laplacian(ins[0], LEVELS, ALPHA, BETA, outs[0])
            torch.cuda.synchronize()
            hip.hipEventRecord(ev0, lib_stream)
            for i in range(REPS):
                hl.local_laplacian(ins[i % len(ins)], LEVELS, ALPHA, BETA, outs[i % len(outs)])
            hip.hipEventRecord(ev1, lib_stream)
            hip.hipEventSynchronize(ev1)
            ms = ctypes.c_float()
            hip.hipEventElapsedTime(ctypes.byref(ms), ev0, ev1)
            back_to_back_ms = ms.value / REPS if back_to_back_ms is None else min(back_to_back_ms, ms.value / REPS)
    finally:
        hl.kernel_timing_only(None)
    hip.hipEventDestroy(ev0), hip.hipEventDestroy(ev1)
    for a, o in zip(ins, outs):   # the outputs were left half-made by the selective runs: complete calls again
        hl.local_laplacian(a, LEVELS, ALPHA, BETA, o)
    torch.cuda.synchronize()

    # the practical HBM ceiling of this device, measured live: a float4 copy kernel over 1 GiB buffers (4x the MALL), HIP
    # events over 10 launches (halide_amd/csrc/membench.hip) — what "HBM-bound" can reach at best for mixed read/write traffic
    copy_ceiling, ceiling_detail = None, None
    if rank == 0 and not args.no_ceiling:
        try:
            ceiling_detail = hl.membench(1 << 30, 10)
            copy_ceiling = ceiling_detail["copy_gbs"]
        except hl.HalideError:   # e.g. no room for two 1 GiB buffers: the headline line does not depend on it
            copy_ceiling = None

    one_call, one_call_clock = None, None
    if rank == 0 and world == 1:
        torch.cuda.synchronize()
        one_call_full = {k: one_call_protocol(k) for k in ("noise", "smooth", "natural")}
        one_call = {k: v[0] for k, v in one_call_full.items()}
        one_call_clock = {k: v[1] for k, v in one_call_full.items()}

    variants = None
    if rank == 0 and world == 1 and not args.no_variants:
        variants = {"unit": "Mpx/s", "smooth_3840x2160": run_variant(W, H, "smooth", 8, 100),
                    "natural_tiled_3840x2160": run_variant(W, H, "natural", 8, 100),
                    "smooth_7680x4320": run_variant(2 * W, 2 * H, "smooth", 4, 40),
                    "uniform_noise_7680x4320": run_variant(2 * W, 2 * H, "noise", 4, 40)}
        variants["frame_ms"] = {k: round((4 if "7680" in k else 1) * W * H / v / 1e3, 4) for k, v in variants.items() if k != "unit"}

    other_configs = None
    if rank == 0 and world == 1 and not args.no_other_configs:
        # BASELINE.json configs[1], [3], [4] in the same driver-run line: ms per call, the roofline that bounds each, and a
        # bounded oracle cpu_baseline beside it (bench_apps.py holds the measurement code)
        for b in ins + outs:
            b.device_free()
        import bench_apps
        other_configs = []
        bench_apps.run(("bilateral_grid", "nl_means", "conv_layer_bf16"), 8, other_configs.append, cpu=not args.no_cpu_baseline)
        # configs[3] as BASELINE.json states it: the BATCH of 32 nl_means frames (bench_batch.py's workload at N = 1: resident on
        # this GPU, no exchange step), enqueued back to back, one sync at the end
        other_configs.append(bench_apps.nl_means_batch32())

    if rank == 0:
        frames_per_step = FRAMES_PER_STEP * args.passes
        px_per_step = world * frames_per_step * W * H
        value = px_per_step * args.steps / elapsed / 1e6
        frame_ms = elapsed / (args.steps * frames_per_step) * 1e3
        # per-launch view: every launch of the chain is reported under its own name (ll_down_strip:1 ... ll_up:6 ...)
        per_frame = {k["name"]: k["total_ms"] / (3 * FRAMES_PER_STEP) for k in kernels}
        # dominant kernel = the launch with the longest average duration; its ALGORITHMIC bytes (compulsory reads +
        # writes of that launch, declared next to the launch in halide_amd/csrc/local_laplacian.hip)
        dom_rec = max(kernels, key=lambda k: k["avg_ms"])
        dom = dom_rec["name"]
        alg_bytes = dom_rec.get("alg_bytes") or ALG_BYTES_PER_PX * W * H
        # duration: the kernel alone, back to back (comparable with rocprofv3's kernel trace); the event bracket of the chain is
        # kept beside it as kernel_avg_ms_hip_events
        kernel_ms = back_to_back_ms or dom_rec["avg_ms"]
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
        # measured HBM traffic of that kernel (rocprofv3 PMC passes, FETCH_SIZE corrected x2 as
        # MI355X_MICROARCH.md prescribes for wide coalesced reads + WRITE_SIZE), committed under profiles/
        traffic = None
        frame_traffic = None   # all launches of one frame: what the whole pipeline moves through HBM / MALL
        traffic_source = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            import hashlib
            with open(tpath) as f:
                tj = json.load(f)
            per_launch = tj.get("bytes_per_launch", {})
            with open(os.path.join(ROOT, "halide_amd", "csrc", "local_laplacian.hip"), "rb") as f:
                sha_now = hashlib.sha256(f.read()).hexdigest()
            # NOT measured in this run: counters of a separate rocprofv3 PMC run (committed under profiles/).  Printed only while the
            # kernel source is byte for byte the one that run profiled — a later edit of the kernels makes them stale
            if tj.get("kernel_source_sha256") == sha_now:
                traffic = per_launch.get(dom)      # the dominant kernel alone on a stream that owns the device: that geometry
                # the whole frame: the counters of the geometry the headline loop ran — a frame queue's when it ran on frame queues
                # (fewer seam rows and tile halos re-read than in the one-call geometry), else the device-wide ones
                per_q = tj.get("bytes_per_launch_frame_queue") or {}
                on_queues = args.partitions > 1 and bool(streams)
                frame_set = per_q if (on_queues and all(k in per_q for k in per_frame)) else per_launch
                if all(k in frame_set for k in per_frame):
                    frame_traffic = int(sum(frame_set[k] for k in per_frame))
                traffic_source = {"file": tj.get("source"), "git_head_when_collected": tj.get("git_head_when_collected"),
                                  "measured_in_this_run": False,
                                  "pipeline_traffic_geometry": "frame queue (" + str(tj.get("frame_queue_source")) + ")" if frame_set is per_q
                                  else "one stream that owns the device"}
            else:
                traffic_source = {"file": tj.get("source"), "stale": "local_laplacian.hip changed since these counters were collected; traffic withheld"}
        result = {
            "metric": "megapixels/sec local_laplacian 8-level fp32 4K",
            "value": round(value, 2), "unit": "Mpx/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            # `value` is measured on uniform noise, the input the reference's own benchmark protocol feeds (RunGen); the same call on
            # the two other input classes of SURVEY.md §8d (details under config.variants): the smooth synthetic frame (rounds 1-5's
            # headline input) and the reference checkout's one natural image, tiled
            "value_input": "uniform full-range noise (tools/RunGen.h:482-505)",
            "value_smooth": None if not variants else variants.get("smooth_3840x2160"),
            "value_natural_tiled": None if not variants else variants.get("natural_tiled_3840x2160"),
            # which canonical float form the library computes (hlmi_canon_fma(): 1 = mul+add pairs fused as LLVM contracts the
            # reference's float operations, 0 = one rounding per operator) and the clock / power state during the timed region
            "canon_fma": hl.canon_fma(),
            "clock_state_timed_region": sampler.summary(),
            # the reference driver's protocol beside the frames-in-flight `value`: one call + device_sync on the device's own
            # stream, min over 10 samples (apps/local_laplacian/process.cpp:36-39) — ms per call and the Mpx/s that is
            "ms_per_call_one_stream": None if not one_call else {k: round(v * 1e3, 4) for k, v in one_call.items()},
            "mpx_per_s_one_call_one_stream": None if not one_call else {k: round(W * H / v / 1e6, 1) for k, v in one_call.items()},
            "clock_state_one_call_one_stream": one_call_clock,
            "config": {"workload": "apps/local_laplacian J=8 levels=8 alpha=1/7 beta=1, u16 RGB planar 3840x2160",
                       "frames_per_step_per_gpu": frames_per_step, "distinct_frames_per_gpu": FRAMES_PER_STEP,
                       "passes_per_step": args.passes, "frame_ms": round(frame_ms, 4),
                       # remap(x) depends on (levels, alpha) only: its 3585-entry table is memoised across calls, so 8 of the
                       # reference's 9 stages run inside the timed region (HLMI_LL_NO_LUT_CACHE=1 re-runs it: +1 launch of ~4 us)
                       "lut_cached": True,
                       "streams_per_gpu": max(1, len(streams)), "frame_scheduling": mode,
                       "boundary": "C ABI local_laplacian(halide_buffer_t*,int32,float,float,halide_buffer_t*)",
                       "input": "uniform full-range noise, seeded per frame (SURVEY.md §8d (i): the reference's RunGen benchmark input); smooth / natural under `variants`",
                       "variants": variants,
                       "other_configs": other_configs,
                       "rccl_ranks": rccl_ranks,
                       "parallelism": f"frames sharded over {world} GPU(s), no data-path collective"},
            "roofline": {"bound": "hbm", "kernel": dom,
                         # the three fractions of the 8 TB/s peak, first: (1) the dominant kernel on its algorithmic bytes over its
                         # back-to-back duration (what profiles/*kernel_stats.csv gives for the same kernel); (2) the whole pipeline on
                         # the compulsory 12 B/px over the frame time of the timed region; (3) the pipeline on its measured (PMC) bytes
                         "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "pipeline_frac": round(ALG_BYTES_PER_PX * W * H / (frame_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                         "pipeline_traffic_frac": None if frame_traffic is None else
                         round(frame_traffic / (frame_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "traffic": traffic,
                         "traffic_source": traffic_source,
                         # the same fraction on the bytes the kernel actually moved (PMC) instead of its algorithmic bytes
                         "frac_moved": None if traffic is None else round(traffic / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                         "alg_bytes_per_launch": int(alg_bytes),
                         "kernel_avg_ms": round(kernel_ms, 5),
                         "kernel_avg_ms_method": f"{REPS} launches of the kernel alone, back to back on the device's stream, HIP events around "
                                                 "the run, min of 3 rounds (hlmi_kernel_timing_only)",
                         "kernel_avg_ms_hip_events": round(dom_rec["avg_ms"], 5),
                         "pipeline_alg_bytes_per_frame": ALG_BYTES_PER_PX * W * H,
                         # measured (PMC) bytes of all launches of a frame over the frame time: the pipeline's real HBM load
                         "pipeline_traffic_per_frame": frame_traffic,
                         # the same against the measured copy ceiling instead of the 8 TB/s spec figure
                         "hbm_copy_ceiling_gbs": None if copy_ceiling is None else round(copy_ceiling, 1),
                         "hbm_ceiling_detail": ceiling_detail,
                         "pipeline_traffic_frac_of_copy_ceiling": None if (frame_traffic is None or not copy_ceiling) else
                         round(frame_traffic / (frame_ms * 1e-3) / 1e9 / copy_ceiling, 4),
                         # ... and against the float4-copy figure MI355X_MICROARCH.md measured (6.29 TB/s)
                         "pipeline_traffic_frac_of_guide_copy_6290": None if frame_traffic is None else
                         round(frame_traffic / (frame_ms * 1e-3) / 1e9 / 6290.0, 4),
                         # HIP-event brackets around each launch on ONE stream: they include the gap to the previous launch and read
                         # longer than the kernels run (rocprofv3's kernel trace, profiles/, has the durations) — relative weights
                         "kernel_ms_hip_events_one_stream": {k: round(v, 5) for k, v in per_frame.items()},
                         # per-launch figures (HIP events, PMC) are taken on ONE device-wide stream; the headline loop runs on four
                         # frame queues, where the kernels use non-temporal frame accesses and a coarser launch geometry (same
                         # bytes within 1 %)
                         "notes": "the headline loop keeps four frames in flight on four hardware queues that all run on the whole "
                                  "device (the CU masks of rounds 3-5 confined nothing: profiles/r06_cu_mask_probe.txt); the frame "
                                  "time is the sum of the kernels' back-to-back floors (profiles/r06_partition_kernel_probe.txt); "
                                  "profiles/r04_power_and_partition_scaling.txt read 1040 W with one queue busy, ~1370 W with two or four"},
        }
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(frames[0])
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
