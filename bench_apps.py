#!/usr/bin/env python3
"""bench_apps.py — the other rows of SURVEY.md §8(a) at their BASELINE.json configs, one JSON line per pipeline.

bench.py measures the headline metric (local_laplacian 4K); this script times the remaining entry points of
libhlmi.so the same way — inputs resident in HBM, calls through the C ABI, `benchmark()`-style best-of-samples of
`iters` back-to-back calls + device sync (tools/halide_benchmark.h:165-241 of the reference) — and prices each
against the roofline that bounds it: HBM bytes for the stencil pipelines, fp32 VALU for
nl_means, bf16 / f32 matrix-core flops for conv_layer.

    python bench_apps.py [--only name,name] [--samples 5]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md
VALU_F32_PEAK_TF = 157.3       # packed-FMA vector peak
MFMA_BF16_PEAK_TF = 2500.0     # dense
MFMA_F32_PEAK_TF = 157.3


def cpu_time(fn, budget_s=4.0, max_reps=8):
    """Bounded timing of one oracle call on the host cores (cpu_baseline leg only): one warm call, then repetitions until
    `budget_s` is spent.  Returns (seconds per call, calls timed)."""
    fn()
    n, t0 = 0, time.perf_counter()
    while True:
        fn()
        n += 1
        dt = time.perf_counter() - t0
        if dt > budget_s or n >= max_reps:
            return dt / n, n


def run(only=(), samples=5, sink=None, cpu=False, batched=True):
    """Times the pipelines named in `only` (all when empty) and hands one dict per pipeline to `sink` (default: print as a
    JSON line).  cpu=True adds a bounded `cpu_baseline` (the C oracle, OpenMP, kind "port") beside the BASELINE.json
    configs (bilateral_grid, nl_means, conv_layer_bf16) — bench.py's `other_configs` leg."""
    import numpy as np
    import halide_amd as hl
    if hl.device_count() < 1:   # (no torch here: its first import on a fresh box can take minutes)
        raise SystemExit("bench_apps.py needs a gfx950 device (no CPU fallback)")
    rng = np.random.default_rng(0)
    only = set(only)
    if sink is None:
        sink = lambda d: print(json.dumps(d), flush=True)

    class _A:
        pass
    args = _A()
    args.samples = samples

    def cpu_base(fn, units, unit, what):
        if not cpu:
            return {}
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib  # cpu_baseline leg only: the checker timed as the host-CPU reference point
        oracle_lib.set_canon(hl.canon_fma())
        sec, n = cpu_time(lambda: fn(oracle_lib))
        return {"cpu_baseline": {"value": round(units / sec / 1e6, 3), "unit": unit, "cores": os.cpu_count() or 1, "kind": "port",
                                 "ms_per_call": round(sec * 1e3, 2), "sample": f"{n} calls of {what} (C oracle, OpenMP, all host threads)"}}

    last_clock = [None]

    def timed(call, sync_buf, iters):
        call()
        sync_buf.device_sync()
        best = 1e30
        for _ in range(args.samples):
            t0 = time.perf_counter()
            for _ in range(iters):
                call()
            sync_buf.device_sync()
            best = min(best, (time.perf_counter() - t0) / iters)
        # shader clock / package power under this call pattern (0.25 s of it, untimed): per-call figures of short pipelines move with
        # the box's clock state (bench.ClockSampler)
        import bench
        with bench.ClockSampler(0) as cs:
            t_end = time.perf_counter() + 0.25
            while time.perf_counter() < t_end:
                for _ in range(iters):
                    call()
                sync_buf.device_sync()
        last_clock[0] = cs.summary()
        return best

    def timed_batched(make_call, nframes=8, rounds=6):
        """Throughput mode: `nframes` independent frames (own inputs and outputs) enqueued back to back without host
        synchronisation — on one stream, and spread over 2 / 4 frame-queue streams (halide_hip_partition_stream, as bench.py
        runs the headline pipeline); returns (seconds per frame, scheduling) of the fastest.  The single-call figure is a
        latency: two or three short dependent launches cannot fill 256 CUs, several frames side by side can."""
        if not batched:   # profiler runs (rocprofv3 over this script): one call at a time only
            return None
        calls = [make_call(i) for i in range(nframes)]
        hip = hl.hip_runtime()
        best, how = 1e30, None
        for nparts in (1, 2, 4):
            streams = [None] if nparts == 1 else [hl.partition_stream(p, nparts) for p in range(nparts)]
            if nparts > 1 and not all(streams):
                continue

            def one_round():
                for i, c in enumerate(calls):
                    hl.set_stream(streams[i % nparts])
                    c()
                hl.set_stream(None)
            one_round()
            hip.hipDeviceSynchronize()
            for _ in range(args.samples):
                t0 = time.perf_counter()
                for _ in range(rounds):
                    one_round()
                hip.hipDeviceSynchronize()
                dt = (time.perf_counter() - t0) / (rounds * nframes)
                if dt < best:
                    best, how = dt, ("1 stream, back to back" if nparts == 1 else f"{nparts} frame-queue streams")
        return best, how

    def batched_fields(tb, unit_work, peak, what):
        if tb is None:
            return {}
        t, how = tb
        return {"batched": {"ms_per_frame": round(t * 1e3, 4), "frames_in_flight": 8, "streams": how,
                            "roofline_frac": round(unit_work / t / peak, 4), "bound": what}}

    def kernels(call, sync_buf):
        hl.kernel_timing_reset()
        hl.kernel_timing(True)
        for _ in range(3):
            call()
        sync_buf.device_sync()
        hl.kernel_timing(False)
        rep = hl.kernel_timing_report()
        hl.kernel_timing_reset()
        return {k["name"]: round(k["total_ms"] / 3, 5) for k in rep}

    def emit(name, workload, t, mpx, bound, achieved, peak, unit, extra):
        sink({"pipeline": name, "workload": workload, "ms_per_call": round(t * 1e3, 4),
              "value": round(mpx / t / 1e6, 1), "unit": "Mpx/s",
              "roofline": {"bound": bound, "achieved": round(achieved, 1), "peak": peak, "unit": unit,
                           "frac": round(achieved / peak, 4)}, "clock_state": last_clock[0], **extra})

    # ---- the practical HBM ceiling: copy / read / write kernels over 1 GiB buffers (4x the MALL)
    if not only or "membench" in only:
        m = hl.membench(1 << 30, 10)
        sink({"pipeline": "membench", "workload": "grid-stride float4 kernels over 1 GiB buffers, HIP events over 10 launches",
              "copy_gbs": round(m["copy_gbs"], 1), "read_gbs": round(m["read_gbs"], 1),
              "write_gbs": round(m["write_gbs"], 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
              "copy_frac_of_peak": round(m["copy_gbs"] / HBM_PEAK_GBS, 4)})

    # ---- the headline pipeline on the other inputs SURVEY.md §8(d) lists for configs[2] (bench.py uses the smooth synthetic
    #      frame): uniform full-range noise — the worst case for the data-dependent plane gathers of the up pass — and
    #      7680x4320; one stream, one frame at a time
    if not only or "local_laplacian" in only:
        for tag, (W, H), kind in (("uniform noise 3840x2160", (3840, 2160), "uniform"), ("smooth 7680x4320", (7680, 4320), "smooth"),
                                  ("uniform noise 7680x4320", (7680, 4320), "uniform")):
            if kind == "uniform":
                img = rng.integers(0, 65536, (3, H, W), dtype=np.uint16)
            else:
                yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
                base = (np.sin(xx / 311.0) + np.cos(yy / 173.0) + np.sin((xx + yy) / 97.0) + 3.3) / 6.6
                img = np.clip(np.stack([base * 65535.0, np.roll(base, 64, 1) * 52000.0, base[::-1] * 46000.0]) +
                              rng.normal(0.0, 900.0, (3, H, W)).astype(np.float32), 0, 65535).astype(np.uint16)
            a, o = hl.Buffer(img), hl.Buffer(np.zeros_like(img))
            call = lambda: hl.local_laplacian(a, 8, 1.0 / 7.0, 1.0, o)
            t = timed(call, o, 20)
            emit("local_laplacian", f"apps/local_laplacian J=8 levels=8, u16 RGB, {tag}, 1 stream", t, W * H, "hbm",
                 12.0 * W * H / t / 1e9, HBM_PEAK_GBS, "GB/s", {"alg_bytes": 12 * W * H})

    # ---- configs[0]: blur 3x3, u16 1536x2560 (input 1538x2562)
    if not only or "blur" in only:
        W, H = 1536, 2560
        a = hl.Buffer(rng.integers(0, 65536, (H + 2, W + 2), dtype=np.uint16))
        o = hl.Buffer(np.zeros((H, W), np.uint16))
        call = lambda: hl.halide_blur(a, o)
        t = timed(call, o, 50)
        emit("halide_blur", "apps/blur 3x3 box, u16 1536x2560", t, W * H, "hbm", 4.0 * W * H / t / 1e9, HBM_PEAK_GBS, "GB/s",
             {"alg_bytes": 4 * W * H, "kernels_ms": kernels(call, o)})

    # ---- configs[1]: bilateral_grid f32 1920x1080, r_sigma 0.1
    if not only or "bilateral_grid" in only:
        W, H = 1920, 1080
        yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
        img = (0.5 + 0.25 * np.sin(xx / 97.0) * np.cos(yy / 61.0) + 0.2 * (xx > W / 2) + rng.normal(0, 0.03, (H, W))).clip(0, 1)
        a, o = hl.Buffer(img.astype(np.float32)), hl.Buffer(np.zeros((H, W), np.float32))
        call = lambda: hl.bilateral_grid(a, 0.1, o)
        t = timed(call, o, 50)
        img32 = img.astype(np.float32)

        def mk_bg(i):
            ai, oi = hl.Buffer(np.roll(img32, 17 * i, 1).copy()), hl.Buffer(np.zeros((H, W), np.float32))
            return lambda: hl.bilateral_grid(ai, 0.1, oi)
        tb = timed_batched(mk_bg)
        emit("bilateral_grid", "apps/bilateral_grid f32 1920x1080 s_sigma=8 r_sigma=0.1", t, W * H, "hbm",
             8.0 * W * H / t / 1e9, HBM_PEAK_GBS, "GB/s",
             {"alg_bytes": 8 * W * H, "kernels_ms": kernels(call, o), **batched_fields(tb, 8.0 * W * H / 1e9, HBM_PEAK_GBS, "hbm"),
              **cpu_base(lambda ol: ol.bilateral_grid(img32, 0.1), W * H, "Mpx/s", "1920x1080 f32")})

    # ---- stencil_chain u16 1536x2560, 32 stages
    if not only or "stencil_chain" in only:
        W, H = 1536, 2560
        a = hl.Buffer(rng.integers(0, 65536, (H, W), dtype=np.uint16))
        o = hl.Buffer(np.zeros((H, W), np.uint16))
        call = lambda: hl.stencil_chain(a, o)
        t = timed(call, o, 20)
        # Priced on the operations the kernel EXECUTES: the separable 5 + 5 form is 10 u16 multiply-adds per pixel and stage
        # (the reference's 25-tap form would be 2.5x more), over the 128 x 128 register window of a tile whose 8 fused stages
        # leave 96 x 96 outputs (1.78x halo recomputation; the LDS-window kernel, HLMI_SC_LDS=1: 128 x 96 -> 96 x 64, 2.0x) = 569
        # executed multiply-adds per output pixel; bound: the packed 16-bit integer VALU rate (v_pk_mad_u16: 2 MACs per lane
        # and clock = the packed-f32 figure).  Next to it the figure on the ALGORITHMIC count (32 stages x 25 taps = 1600 MACs
        # per pixel) and the fraction of the HBM roofline SURVEY.md §8(d) assigns the pipeline (4 B/px: read + write once).
        halo = (128.0 * 128.0) / (96.0 * 96.0)
        ops_exec = 2.0 * 32 * 10 * halo * W * H
        ops_alg = 2.0 * 1600 * W * H
        sc_src = rng.integers(0, 65536, (H, W), dtype=np.uint16)

        def mk_sc(i):
            ai, oi = hl.Buffer(np.roll(sc_src, 13 * i, 1).copy()), hl.Buffer(np.zeros((H, W), np.uint16))
            return lambda: hl.stencil_chain(ai, oi)
        tb = timed_batched(mk_sc, rounds=3)
        emit("stencil_chain", "apps/stencil_chain 32 stages 5x5, u16 1536x2560", t, W * H, "valu", ops_exec / t / 1e12,
             VALU_F32_PEAK_TF, "TOP/s (u16, packed, executed)",
             {"executed_ops": ops_exec, "halo_recompute": halo, "alg_ops": ops_alg,
              "alg_ops_frac": round(ops_alg / t / 1e12 / VALU_F32_PEAK_TF, 4), "alg_bytes": 4 * W * H,
              "hbm_gbs": 4.0 * W * H / t / 1e9, "hbm_frac": round(4.0 * W * H / t / 1e9 / HBM_PEAK_GBS, 4),
              "kernels_ms": kernels(call, o), **batched_fields(tb, ops_exec / 1e12, VALU_F32_PEAK_TF, "valu")})

    # ---- camera_pipe 2592x1968 raw -> 2560x1920x3 u8
    if not only or "camera_pipe" in only:
        IW, IH, OW, OH = 2592, 1968, 2560, 1920
        raw = hl.Buffer(rng.integers(0, 1024, (IH, IW), dtype=np.uint16))
        m3 = hl.Buffer(np.array([[1.6697, -0.2693, -0.4004, -42.4346], [-0.3576, 1.0615, 1.5949, -37.1158],
                                 [-0.2175, -1.8751, 6.9640, -26.6970]], np.float32))
        m7 = hl.Buffer(np.array([[2.2997, -0.4478, 0.1706, -39.0923], [-0.3826, 1.5906, -0.2080, -25.4311],
                                 [-0.0888, -0.7344, 2.2832, -20.0826]], np.float32))
        o = hl.Buffer(np.zeros((3, OH, OW), np.uint8))
        call = lambda: hl.camera_pipe(raw, m3, m7, 3700.0, 2.0, 50.0, 1.0, 25, 1023, o)
        t = timed(call, o, 50)
        cp_src = rng.integers(0, 1024, (IH, IW), dtype=np.uint16)

        def mk_cp(i):
            ri, oi = hl.Buffer(np.roll(cp_src, 2 * i, 1).copy()), hl.Buffer(np.zeros((3, OH, OW), np.uint8))
            return lambda: hl.camera_pipe(ri, m3, m7, 3700.0, 2.0, 50.0, 1.0, 25, 1023, oi)
        tb = timed_batched(mk_cp)
        # the same call on a raw frame with spatial structure (a smooth scene under the Bayer mosaic + sensor noise): the tone-curve
        # look-ups of neighbouring pixels then mostly share LDS words, where the uniform-noise frame above (what RunGen's benchmark
        # fills its inputs with) makes every one of them a bank conflict lottery
        yy, xx = np.mgrid[0:IH, 0:IW].astype(np.float32)
        scene = (np.sin(xx / 173.0) + np.cos(yy / 97.0) + np.sin((xx + yy) / 311.0) + 3.2) / 6.4
        gain = np.where((yy % 2 == 0) & (xx % 2 == 1), 0.6, np.where((yy % 2 == 1) & (xx % 2 == 0), 0.5, 1.0))   # R and B sites darker than G
        raw_scene = hl.Buffer(np.clip(scene * gain * 900.0 + 40.0 + rng.normal(0.0, 6.0, (IH, IW)), 0, 1023).astype(np.uint16))
        t_scene = timed(lambda: hl.camera_pipe(raw_scene, m3, m7, 3700.0, 2.0, 50.0, 1.0, 25, 1023, o), o, 50)
        emit("camera_pipe", "apps/camera_pipe u16 2592x1968 -> u8 2560x1920x3", t, OW * OH, "hbm", 5.0 * OW * OH / t / 1e9,
             HBM_PEAK_GBS, "GB/s", {"alg_bytes": 5 * OW * OH, "kernels_ms": kernels(call, o), "input": "uniform 10-bit noise",
                                   "ms_per_call_scene_input": round(t_scene * 1e3, 4),
                                   **batched_fields(tb, 5.0 * OW * OH / 1e9, HBM_PEAK_GBS, "hbm")})

    # ---- configs[3]: nl_means 7x7 / 7x7, f32 1920x1080x3 (one frame per call; frames of a batch are independent)
    if not only or "nl_means" in only:
        W, H = 1920, 1080
        nlm_in = rng.random((3, H, W), dtype=np.float32)
        a = hl.Buffer(nlm_in)
        o = hl.Buffer(np.zeros((3, H, W), np.float32))
        call = lambda: hl.nl_means(a, 7, 7, 0.12, o)
        t = timed(call, o, 10)
        flops = 2200.0 * W * H       # ~49 offsets x ~45 flops per pixel

        def mk_nlm(i):
            ai, oi = hl.Buffer(np.roll(nlm_in, 13 * i, 2).copy()), hl.Buffer(np.zeros((3, H, W), np.float32))
            return lambda: hl.nl_means(ai, 7, 7, 0.12, oi)
        tb = timed_batched(mk_nlm, rounds=2)
        emit("nl_means", "apps/nl_means patch 7 search 7 sigma 0.12, f32 1920x1080x3", t, W * H, "valu",
             flops / t / 1e12, VALU_F32_PEAK_TF, "TFLOP/s",
             {"alg_flops": flops, "alg_bytes": 24 * W * H, "kernels_ms": kernels(call, o),
              **batched_fields(tb, flops / 1e12, VALU_F32_PEAK_TF, "valu"),
              **cpu_base(lambda ol: ol.nl_means(nlm_in, 7, 7, 0.12), W * H, "Mpx/s", "1920x1080x3 f32, patch 7 search 7")})

    # ---- unsharp f32 1536x2560x3 (generator estimates)
    if not only or "unsharp" in only:
        W, H = 1536, 2560
        a = hl.Buffer((rng.random((3, H, W), dtype=np.float32) * 0.9 + 0.05).astype(np.float32))
        o = hl.Buffer(np.zeros((3, H, W), np.float32))
        call = lambda: hl.unsharp(a, o)
        t = timed(call, o, 50)
        emit("unsharp", "apps/unsharp sigma=1.5, f32 1536x2560x3", t, W * H, "hbm", 24.0 * W * H / t / 1e9, HBM_PEAK_GBS, "GB/s",
             {"alg_bytes": 24 * W * H, "kernels_ms": kernels(call, o)})

    # ---- max_filter f32 1536x2560x3 (generator estimates).  LDS bound.  `lds_bytes` is the ALGORITHMIC sample count, not the bytes the
    # LDS moves: every output folds 55 footprint rows x 2 four-byte samples (440 B) and the four doubling slices of a 64x64 tile's 120
    # staged rows are written once each (15 chunks x 4 x 1024 words x 4 B / 4096 outputs = 60 B).  With 16 output rows per lane a lane's
    # outputs share sample reads (about 70 LDS read instructions per output row of eight, max_filter.hip), so the instructions issued
    # move fewer bytes than this figure: `frac` is the algorithmic sample rate against the ds_read_b32 rate of 128 B/clk/CU
    # (MI355X_MICROARCH.md, LDS table), an upper bound on the LDS utilisation, and is labelled that way.
    if not only or "max_filter" in only:
        W, H = 1536, 2560
        a = hl.Buffer(rng.random((3, H, W), dtype=np.float32))
        o = hl.Buffer(np.zeros((3, H, W), np.float32))
        call = lambda: hl.max_filter(a, o)
        t = timed(call, o, 20)
        lds_bytes = (440.0 + 60.0) * 3 * W * H
        emit("max_filter", "apps/max_filter radius 26, f32 1536x2560x3", t, W * H, "lds", lds_bytes / t / 1e9, 256 * 128 * 2.4, "GB/s",
             {"alg_bytes": 24 * W * H, "hbm_gbs": 24.0 * W * H / t / 1e9, "lds_bytes": lds_bytes,
              "lds_bytes_are": "algorithmic samples (440 B read + 60 B written per output), not LDS instructions issued: shared reads make the real traffic smaller",
              "kernels_ms": kernels(call, o)})

    # ---- hist u8 1536x2560x3 (generator estimates)
    if not only or "hist" in only:
        W, H = 1536, 2560
        a = hl.Buffer(rng.integers(0, 256, (3, H, W), dtype=np.uint8))
        o = hl.Buffer(np.zeros((3, H, W), np.uint8))
        call = lambda: hl.hist(a, o)
        t = timed(call, o, 50)
        emit("hist", "apps/hist histogram equalisation, u8 1536x2560x3", t, W * H, "hbm", 9.0 * W * H / t / 1e9, HBM_PEAK_GBS, "GB/s",
             {"alg_bytes": 9 * W * H, "kernels_ms": kernels(call, o)})

    # ---- harris f32 1536x2560x3 -> 1530x2554 (generator estimates)
    if not only or "harris" in only:
        W, H = 1536, 2560
        a = hl.Buffer(rng.random((3, H, W), dtype=np.float32))
        o = hl.Buffer(np.zeros((H - 6, W - 6), np.float32)).set_min(3, 3)
        call = lambda: hl.harris(a, o)
        t = timed(call, o, 50)
        emit("harris", "apps/harris corner response, f32 1536x2560x3 -> 1530x2554", t, (W - 6) * (H - 6), "hbm", 16.0 * W * H / t / 1e9,
             HBM_PEAK_GBS, "GB/s", {"alg_bytes": 16 * W * H, "kernels_ms": kernels(call, o)})

    # ---- interpolate f32 1536x2560x4 -> x3 (generator estimates)
    if not only or "interpolate" in only:
        W, H = 1536, 2560
        img = rng.random((4, H, W), dtype=np.float32)
        img[3][rng.random((H, W)) < 0.4] = 0.0
        a = hl.Buffer(img)
        o = hl.Buffer(np.zeros((3, H, W), np.float32))
        call = lambda: hl.interpolate(a, o)
        t = timed(call, o, 20)
        emit("interpolate", "apps/interpolate 10-level pull-push, f32 1536x2560x4 -> x3", t, W * H, "hbm", 28.0 * W * H / t / 1e9,
             HBM_PEAK_GBS, "GB/s", {"alg_bytes": 28 * W * H, "kernels_ms": kernels(call, o)})

    # ---- iir_blur f32 1536x2560x3, alpha 0.1 (the shape the generator pins)
    if not only or "iir_blur" in only:
        W, H = 1536, 2560
        a = hl.Buffer(rng.random((3, H, W), dtype=np.float32))
        o = hl.Buffer(np.zeros((3, H, W), np.float32))
        call = lambda: hl.iir_blur(a, 0.1, o)
        t = timed(call, o, 10)
        emit("iir_blur", "apps/iir_blur alpha=0.1, f32 1536x2560x3", t, W * H, "hbm", 24.0 * W * H / t / 1e9, HBM_PEAK_GBS, "GB/s",
             {"alg_bytes": 24 * W * H, "kernels_ms": kernels(call, o),
              "note": "bound by the sequential recurrence: 2 (W + H) dependent multiply-add steps per scan line"})

    # ---- lens_blur u8 stereo pair 768x1280 (the size of apps/images/rgb.png the reference's Makefile feeds process.cpp), 32 slices, 32 samples
    if not only or "lens_blur" in only:
        W, H = 768, 1280
        left = rng.integers(0, 256, (3, H, W), dtype=np.uint8)
        right = np.roll(left, 7, 2)
        a, b = hl.Buffer(left), hl.Buffer(right)
        o = hl.Buffer(np.zeros((3, H, W), np.float32))
        call = lambda: hl.lens_blur(a, b, 32, 13, 0.5, 32, o)
        t = timed(call, o, 5)
        # traffic of the decomposition the library runs: level 0 of the push pyramid (33 planes: 32 costs + the one confidence
        # plane the generator's 32 copies collapse to) is never stored; levels >= 1 of the push pyramid are written once and
        # read twice (the next level, the pull level), the pull levels written once and read once: 5 accesses of 33 * 4 B
        # per level-1 element, 4/3 of that for the coarser levels, + the depth record / radius / output planes at level 0.
        # (The staged front end, HLMI_LB_UNFUSED=1, adds 3 * 132 B per pixel.)
        bytes_px = (5.0 / 4.0) * (4.0 / 3.0) * 33 * 4 + 6 + 4 * 4 + 12
        emit("lens_blur", "apps/lens_blur 32 slices, 32 aperture samples, u8 768x1280x3 stereo pair -> f32", t, W * H, "hbm",
             bytes_px * W * H / t / 1e9, HBM_PEAK_GBS, "GB/s", {"alg_bytes": bytes_px * W * H, "kernels_ms": kernels(call, o),
                                                              "note": "instruction-bound (cost stack ~800, depth ~1750, samples ~2400 VALU instructions per pixel); "
                                                                      "level 0 of the push pyramid is never stored"})

    # ---- bgu at the generator's estimates (bgu_generator.cpp:674-687): 192x320 low-res pair, 1536x2560 full-res image
    if not only or "bgu" in only:
        W, H = 1536, 2560
        yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
        hi = np.stack([(np.sin(xx / (91.0 + 7 * c)) + np.cos(yy / (133.0 - 9 * c))) * 0.22 + 0.5 for c in range(3)]).astype(np.float32)
        hi += rng.random((3, H, W), dtype=np.float32) * 0.06
        lo = hi.reshape(3, H // 8, 8, W // 8, 8).mean(axis=(2, 4)).astype(np.float32)
        val = (lo * lo * (3 - 2 * lo)).astype(np.float32)
        bl, bv, bh = hl.Buffer(lo), hl.Buffer(val), hl.Buffer(hi)
        o = hl.Buffer(np.zeros((3, H, W), np.float32))
        call = lambda: hl.bgu(1.0 / 8.0, 16, bl, bv, bh, o)
        t = timed(call, o, 20)
        emit("bgu", "apps/bgu r_sigma=1/8 s_sigma=16, f32 192x320x3 pair -> 1536x2560x3", t, W * H, "hbm", 24.0 * W * H / t / 1e9,
             HBM_PEAK_GBS, "GB/s", {"alg_bytes": 24 * W * H, "kernels_ms": kernels(call, o)})

    # ---- depthwise_separable_conv at the driver's shape (MobileNet-v2 layer 2, process.cpp:13)
    if not only or "depthwise_separable_conv" in only:
        N, Hh, Ww, CI, CO = 4, 112, 112, 32, 16
        bufs = [hl.Buffer(rng.uniform(-1, 1, sh).astype(np.float32)) for sh in ((N, Hh, Ww, CI), (3, 3, CI, 1), (CI, CO), (CO,))]
        o = hl.Buffer(np.zeros((N, Hh, Ww, CO), np.float32))
        call = lambda: hl.depthwise_separable_conv(*bufs, o)
        t = timed(call, o, 50)
        nbytes = 4.0 * N * Hh * Ww * (CI + CO)
        emit("depthwise_separable_conv", "apps/depthwise_separable_conv N=4 CI=32 CO=16 CM=1 112x112 3x3", t, N * Hh * Ww, "hbm",
             nbytes / t / 1e9, HBM_PEAK_GBS, "GB/s", {"alg_bytes": nbytes, "alg_flops": 2.0 * N * Hh * Ww * (CI * 9 + CI * CO),
                                                      "kernels_ms": kernels(call, o)})

    # ---- configs[4]: conv_layer N=16 CI=CO=128 56x56 k=3 — bf16 matrix cores and the exact f32 path
    for name, fn, peak in (("conv_layer_bf16", "conv_layer_bf16", MFMA_BF16_PEAK_TF), ("conv_layer", "conv_layer", MFMA_F32_PEAK_TF)):
        if only and name not in only:
            continue
        N, Hh, Ww, CI, CO = 16, 56, 56, 128, 128
        c_in = rng.uniform(-1, 1, (N, Hh + 2, Ww + 2, CI)).astype(np.float32)
        c_f = rng.uniform(-1, 1, (CI, 3, 3, CO)).astype(np.float32)
        c_b = rng.uniform(-1, 1, CO).astype(np.float32)
        inp, filt, bias = hl.Buffer(c_in), hl.Buffer(c_f), hl.Buffer(c_b)
        o = hl.Buffer(np.zeros((N, Hh, Ww, CO), np.float32))
        f = getattr(hl, fn)
        call = lambda: f(inp, filt, bias, o)
        t = timed(call, o, 50)
        flops = 2.0 * N * Hh * Ww * CI * CO * 9

        def mk_conv(i):
            ii, oi = hl.Buffer(np.roll(c_in, i, 1).copy()), hl.Buffer(np.zeros((N, Hh, Ww, CO), np.float32))
            return lambda: f(ii, filt, bias, oi)
        tb = timed_batched(mk_conv)
        cold = None
        if "bf16" in name:
            # the bf16 re-ordering of the filter is cached across calls with the same filter buffer (a layer's weights do not change
            # between inferences): the figure above is the cached case; this one re-orders the filter inside every timed call
            os.environ["HLMI_CONV_NO_FILTER_CACHE"] = "1"
            cold = timed(call, o, 50)
            del os.environ["HLMI_CONV_NO_FILTER_CACHE"]
        io_bytes = 4 * (N * (Hh + 2) * (Ww + 2) * CI + N * Hh * Ww * CO)   # the reference's f32 input and output, each moved once
        orc = (lambda ol: ol.conv_layer_bf16(c_in, c_f, c_b)) if "bf16" in name else (lambda ol: ol.conv_layer(c_in, c_f, c_b))
        emit(name, f"apps/conv_layer N=16 CI=CO=128 56x56 k=3 ({'bf16 operands, f32 accumulate' if 'bf16' in name else 'exact f32'})",
             t, N * Hh * Ww, "mfma", flops / t / 1e12, peak, "TFLOP/s",
             {"alg_flops": flops, "alg_bytes": io_bytes, "kernels_ms": kernels(call, o),
              **batched_fields(tb, flops / 1e12, peak, "mfma"),
              **({"filter_cached": True, "ms_per_call_filter_uncached": round(cold * 1e3, 4)} if cold is not None else {}),
              # the second bound: with f32 I/O the call cannot beat its bytes — reported beside the matrix-core fraction
              "hbm_bound": {"alg_bytes": io_bytes, "achieved_gbs": round(io_bytes / t / 1e9, 1), "peak": HBM_PEAK_GBS,
                            "frac": round(io_bytes / t / 1e9 / HBM_PEAK_GBS, 4)},
              **(cpu_base(orc, flops / 1e6, "TFLOP/s", "N=16 56x56 128->128 k=3") if "bf16" in name else {})})


def nl_means_batch32(samples=3):
    """BASELINE.json configs[3] at N = 1: the batch of 32 nl_means frames (7x7 search / 7x7 patch, f32 1920x1080x3, seeds 0..31 as
    SURVEY.md §8d names them) resident on one GPU — what every rank of bench_batch.py does with its share, without the exchange
    step — enqueued back to back over four frame-queue streams, ONE sync at the end; min over `samples` batches."""
    import numpy as np
    import halide_amd as hl
    W, H, B = 1920, 1080, 32
    frames = [np.random.default_rng(seed).random((3, H, W), dtype=np.float32) for seed in range(B)]
    ins = [hl.Buffer(f) for f in frames]
    outs = [hl.Buffer(np.zeros((3, H, W), np.float32)) for _ in range(B)]
    hip = hl.hip_runtime()
    results = {}
    for nparts in (1, 4):
        streams = [None] if nparts == 1 else [hl.partition_stream(p, nparts) for p in range(nparts)]
        if nparts > 1 and not all(streams):
            continue

        def batch():
            for i, (a, o) in enumerate(zip(ins, outs)):
                hl.set_stream(streams[i % nparts])
                hl.nl_means(a, 7, 7, 0.12, o)
            hl.set_stream(None)
            hip.hipDeviceSynchronize()
        batch()      # uploads the inputs, allocates the outputs
        best = 1e30
        for _ in range(samples):
            t0 = time.perf_counter()
            batch()
            best = min(best, time.perf_counter() - t0)
        results["1 stream" if nparts == 1 else f"{nparts} frame-queue streams"] = best
    for b in ins + outs:
        b.device_free()
    how, t = min(results.items(), key=lambda kv: kv[1])
    flops = 2200.0 * W * H * B
    return {"pipeline": "nl_means_batch32", "workload": "BASELINE configs[3] at N=1: 32 frames of apps/nl_means patch 7 search 7 sigma 0.12, "
            "f32 1920x1080x3, resident on one GPU (bench_batch.py shards the same batch over N GPUs with RCCL send/recv)",
            "ms_per_batch": round(t * 1e3, 3), "ms_per_frame": round(t * 1e3 / B, 4), "value": round(B * W * H / t / 1e6, 1), "unit": "Mpx/s",
            "streams": how, "ms_per_batch_by_scheduling": {k: round(v * 1e3, 3) for k, v in results.items()},
            "roofline": {"bound": "valu", "achieved": round(flops / t / 1e12, 2), "peak": VALU_F32_PEAK_TF, "unit": "TFLOP/s",
                         "frac": round(flops / t / 1e12 / VALU_F32_PEAK_TF, 4)}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    ap.add_argument("--samples", type=int, default=5)
    ap.add_argument("--cpu-baseline", action="store_true", help="time the C oracle beside the BASELINE.json configs")
    ap.add_argument("--no-batched", action="store_true", help="skip the frames-in-flight leg (profiler runs)")
    a = ap.parse_args()
    names = [n for n in a.only.split(",") if n]
    if "nl_means_batch32" in names:      # configs[3] as a batch (not one of run()'s per-call pipelines)
        print(json.dumps(nl_means_batch32(a.samples)), flush=True)
        names.remove("nl_means_batch32")
        if not names:
            return
    run(names, a.samples, None, a.cpu_baseline, not a.no_batched)


if __name__ == "__main__":
    main()
