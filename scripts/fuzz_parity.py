#!/usr/bin/env python3
"""Randomised parity sweep: every pipeline on random shapes / origins / parameters, GPU result vs oracle, bit for bit.

    python scripts/fuzz_parity.py [--seed S] [--seconds T] [--only a,b]

Not part of the pytest suite (the suite pins fixed cases); this is the tool that looks for cases the suite does not have.
Prints one line per pipeline (cases run, failures) and the parameters of every failure.  Uses oracle/ as the checker
only (tests/oracle_lib.py)."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import halide_amd as hl  # noqa: E402
import oracle_lib as oracle  # noqa: E402

oracle.set_canon(hl.canon_fma())   # the oracle evaluates the canonical form the loaded library was built for

f32 = np.float32


def image_u16(rng, w, h, kind):
    if kind == 0:
        return rng.integers(0, 65536, (3, h, w), dtype=np.uint16)
    yy, xx = np.mgrid[0:h, 0:w].astype(f32)
    base = (np.sin(xx / 31.0 + rng.random() * 6) + np.cos(yy / 17.0) + 2.2) / 4.4
    img = np.stack([base * 65535.0, np.roll(base, 5, 1) * 52000.0, base[::-1] * 46000.0]) + rng.normal(0, 900, (3, h, w))
    if kind == 2:
        img[:, : h // 2] = rng.integers(0, 2) * 65535          # flat saturated / black half
    return np.clip(img, 0, 65535).astype(np.uint16)


def image_f32(rng, c, w, h):
    yy, xx = np.mgrid[0:h, 0:w].astype(f32)
    return np.stack([(np.sin(xx / (7.0 + i) + rng.random() * 6) + np.cos(yy / (11.0 - i))) * 0.23 + 0.5 + rng.random((h, w)) * 0.1
                     for i in range(c)]).astype(f32)


def same(a, b):
    return a.shape == b.shape and np.array_equal(a.view(np.uint8), b.view(np.uint8))


def rdim(rng, lo, hi):
    """Sizes with a bias to the small and to tile multiples +-1."""
    r = rng.random()
    if r < 0.3:
        return int(rng.integers(lo, min(hi, lo + 20) + 1))
    if r < 0.6:
        base = int(rng.choice([32, 64, 128, 256])) * int(rng.integers(1, 4)) + int(rng.integers(-1, 2))
        return int(min(max(base, lo), hi))
    return int(rng.integers(lo, hi + 1))


# ---- one random case per call; each returns (description, ok)
def case_local_laplacian(rng):
    w, h = rdim(rng, 1, 700), rdim(rng, 1, 500)
    # the fast path (ll_down01e / ll_up0h: levels == 8, width a multiple of 4, even output origin and width) gets most of the draws
    levels = 8 if rng.random() < 0.6 else int(rng.integers(2, 9))
    if rng.random() < 0.6:
        w = max(4, (w + 3) & ~3)
    alpha, beta = f32(rng.choice([1.0 / 7.0, 0.5, 1.0, 0.05]) / max(levels - 1, 1)), f32(rng.choice([1.0, 0.5, 2.0, 0.0]))
    ox, oy = (int(rng.integers(-40, 40)), int(rng.integers(-40, 40))) if rng.random() < 0.5 else (0, 0)
    inp = image_u16(rng, w, h, int(rng.integers(0, 3)))
    a = hl.Buffer(inp).set_min(ox, oy, 0)
    want = oracle.local_laplacian(inp, levels, alpha, beta, origin=(ox, oy))
    crop = ""
    if rng.random() < 0.25 and w >= 8 and h >= 4:
        # the output region strictly inside the input (taps clamp at the INPUT's edges: the crop of the full result is expected)
        x0, y0 = int(rng.integers(0, w // 2)) & ~1, int(rng.integers(0, h // 2))
        cw, ch = max(2, int(rng.integers(1, w - x0 + 1)) & ~1), int(rng.integers(1, h - y0 + 1))
        o = hl.Buffer(np.zeros((3, ch, cw), np.uint16)).set_min(ox + x0, oy + y0, 0)
        want = want[:, y0:y0 + ch, x0:x0 + cw]
        crop = f" crop=({x0},{y0},{cw},{ch})"
    else:
        o = hl.Buffer(np.zeros_like(inp)).set_min(ox, oy, 0)
    hl.local_laplacian(a, levels, alpha, beta, o)
    return f"{w}x{h} levels={levels} alpha={alpha} beta={beta} origin=({ox},{oy}){crop}", same(o.numpy(), want)


def case_bilateral_grid(rng):
    w, h = rdim(rng, 1, 600), rdim(rng, 1, 400)
    r_sigma = f32(rng.choice([0.1, 0.05, 0.25, 0.5]))
    inp = image_f32(rng, 1, w, h)[0]
    a, o = hl.Buffer(inp), hl.Buffer(np.zeros_like(inp))
    hl.bilateral_grid(a, r_sigma, o)
    return f"{w}x{h} r_sigma={r_sigma}", same(o.numpy(), oracle.bilateral_grid(inp, r_sigma))


def case_nl_means(rng):
    w, h = rdim(rng, 1, 300), rdim(rng, 1, 200)
    fast = rng.random() < 0.7
    patch, search = (7, 7) if fast else (int(rng.choice([1, 3, 5])), int(rng.choice([1, 3, 5, 9])))
    if not fast:
        w, h = min(w, 64), min(h, 48)
    sigma = f32(rng.choice([0.12, 0.05, 0.5]))
    inp = image_f32(rng, 3, w, h)
    a, o = hl.Buffer(inp), hl.Buffer(np.zeros_like(inp))
    hl.nl_means(a, patch, search, sigma, o)
    return f"{w}x{h} patch={patch} search={search} sigma={sigma}", same(o.numpy(), oracle.nl_means(inp, patch, search, sigma))


def case_stencil_chain(rng):
    w, h = rdim(rng, 1, 500), rdim(rng, 1, 400)
    inp = rng.integers(0, 65536, (h, w), dtype=np.uint16)
    a, o = hl.Buffer(inp), hl.Buffer(np.zeros_like(inp))
    hl.stencil_chain(a, o)
    return f"{w}x{h}", same(o.numpy(), oracle.stencil_chain(inp))


def case_halide_blur(rng):
    w, h = rdim(rng, 1, 600), rdim(rng, 1, 400)
    inp = rng.integers(0, 65536, (h + 2, w + 2), dtype=np.uint16)
    a, o = hl.Buffer(inp), hl.Buffer(np.zeros((h, w), np.uint16))
    hl.halide_blur(a, o)
    return f"{w}x{h}", same(o.numpy(), oracle.blur(inp))


def case_unsharp(rng):
    w, h = rdim(rng, 1, 500), rdim(rng, 1, 400)
    inp = image_f32(rng, 3, w + 6, h + 6)
    a, o = hl.Buffer(inp).set_min(-3, -3, 0), hl.Buffer(np.zeros((3, h, w), f32))
    hl.unsharp(a, o)
    return f"{w}x{h}", same(o.numpy(), oracle.unsharp(inp, (0, 0), (w, h), (-3, -3)))


def case_harris(rng):
    w, h = rdim(rng, 1, 500), rdim(rng, 1, 400)
    inp = image_f32(rng, 3, w + 4, h + 4)
    a, o = hl.Buffer(inp).set_min(1, 1, 0), hl.Buffer(np.zeros((h, w), f32)).set_min(3, 3)
    hl.harris(a, o)
    return f"{w}x{h}", same(o.numpy(), oracle.harris(inp, (3, 3), (w, h), (1, 1)))


def case_max_filter(rng):
    w, h = rdim(rng, 1, 300), rdim(rng, 1, 200)
    inp = image_f32(rng, 3, w, h)
    a, o = hl.Buffer(inp), hl.Buffer(np.zeros_like(inp))
    hl.max_filter(a, o)
    return f"{w}x{h}", same(o.numpy(), oracle.max_filter(inp))


def case_hist(rng):
    w, h = rdim(rng, 1, 600), rdim(rng, 1, 400)
    inp = rng.integers(0, 256, (3, h, w), dtype=np.uint8) if rng.random() < 0.5 else (image_f32(rng, 3, w, h) * 255).astype(np.uint8)
    a, o = hl.Buffer(inp), hl.Buffer(np.zeros_like(inp))
    hl.hist(a, o)
    return f"{w}x{h}", same(o.numpy(), oracle.hist(inp))


def case_interpolate(rng):
    w, h = rdim(rng, 1, 500), rdim(rng, 1, 400)
    inp = image_f32(rng, 4, w, h)
    inp[3] = (rng.random((h, w)) > rng.random()).astype(f32) * inp[3]
    a, o = hl.Buffer(inp), hl.Buffer(np.zeros((3, h, w), f32))
    hl.interpolate(a, o)
    return f"{w}x{h}", same(o.numpy(), oracle.interpolate(inp))


def case_iir_blur(rng):
    w, h, c = rdim(rng, 1, 400), rdim(rng, 1, 400), int(rng.integers(1, 4))
    alpha = f32(rng.choice([0.1, 0.5, 0.9, 1.0]))
    inp = image_f32(rng, c, w, h)
    a, o = hl.Buffer(inp), hl.Buffer(np.zeros_like(inp))
    hl.iir_blur(a, alpha, o)
    return f"{w}x{h}x{c} alpha={alpha}", same(o.numpy(), oracle.iir_blur(inp, alpha))


def case_bgu(rng):
    W, H = rdim(rng, 8, 500), rdim(rng, 8, 400)
    factor = int(rng.choice([1, 2, 4, 8]))
    lw, lh = max(1, W // factor), max(1, H // factor)
    hi = image_f32(rng, 3, W, H)
    lo = image_f32(rng, 3, lw, lh)
    val = np.clip(lo * lo * (3 - 2 * lo) + rng.normal(0, 0.02, lo.shape), 0, 1).astype(f32)
    r_sigma, s_sigma = f32(rng.choice([1 / 8, 1 / 4, 1 / 16, 0.3])), int(rng.choice([16, 8, 5, 3, 2]))
    x0, y0 = int(rng.integers(0, W)), int(rng.integers(0, H))
    reg = (x0, y0, int(rng.integers(1, W - x0 + 1)), int(rng.integers(1, H - y0 + 1))) if rng.random() < 0.4 else (0, 0, W, H)
    bo = hl.Buffer(np.zeros((3, reg[3], reg[2]), f32)).set_min(reg[0], reg[1], 0)
    hl.bgu(r_sigma, s_sigma, hl.Buffer(lo), hl.Buffer(val), hl.Buffer(hi), bo)
    want = oracle.bgu(r_sigma, s_sigma, lo, val, hi, region=reg)
    return f"{W}x{H} /{factor} r={r_sigma} s={s_sigma} region={reg}", same(bo.numpy(), want)


def case_lens_blur(rng):
    w, h = rdim(rng, 1, 160), rdim(rng, 1, 120)
    slices = int(rng.integers(1, 33))
    focus = int(rng.integers(1, min(slices, 32) + 1))
    scale, samples = f32(rng.choice([0.5, 1.0, 0.2, 0.0])), int(rng.integers(1, 40))
    left = rng.integers(0, 256, (3, h, w), dtype=np.uint8)
    right = np.roll(left, int(rng.integers(0, 9)), 2)
    o = hl.Buffer(np.zeros((3, h, w), f32))
    hl.lens_blur(hl.Buffer(left), hl.Buffer(right), slices, focus, scale, samples, o)
    return f"{w}x{h} slices={slices} focus={focus} scale={scale} samples={samples}", same(o.numpy(), oracle.lens_blur(left, right, slices, focus, scale, samples))


def case_camera_pipe(rng):
    ow, oh = 2 * rdim(rng, 1, 160), 2 * rdim(rng, 1, 120)   # even output sizes, like the reference's 2x2 demosaic tiles
    raw = rng.integers(0, 1024, (oh + 32, ow + 40), dtype=np.uint16)   # the generator's footprint: 40 x 32 more than the output
    if rng.random() < 0.3:
        raw[rng.integers(0, raw.shape[0], 20), rng.integers(0, raw.shape[1], 20)] = 65535   # hot pixels
    m3 = np.array([[1.6697, -0.2693, -0.4004, -42.4346], [-0.3576, 1.0615, 1.5949, -37.1158], [-0.2175, -1.8751, 6.9640, -26.6970]], f32)
    m7 = np.array([[2.2997, -0.4478, 0.1706, -39.0923], [-0.3826, 1.5906, -0.2080, -25.4311], [-0.0888, -0.7344, 2.2832, -20.0826]], f32)
    ct, gamma, contrast, sharpen = f32(rng.choice([3700.0, 3200.0, 7000.0, 5000.0])), f32(rng.choice([2.0, 1.0, 2.2])), f32(rng.choice([50.0, 0.0, 100.0])), f32(rng.choice([1.0, 0.0, 2.5]))
    black, white = int(rng.choice([25, 0, 64])), int(rng.choice([1023, 900, 4095]))
    o = hl.Buffer(np.zeros((3, oh, ow), np.uint8))
    hl.camera_pipe(hl.Buffer(raw), hl.Buffer(m3), hl.Buffer(m7), ct, gamma, contrast, sharpen, black, white, o)
    want = oracle.camera_pipe(raw, m3, m7, ct, gamma, contrast, sharpen, black, white, ow, oh)
    return f"{ow}x{oh} ct={ct} gamma={gamma} contrast={contrast} sharpen={sharpen} black={black} white={white}", same(o.numpy(), want)


def case_depthwise_separable_conv(rng):
    n, hh, ww = int(rng.integers(1, 4)), rdim(rng, 1, 40), rdim(rng, 1, 40)
    ci, co = int(rng.choice([32, 16, 8, 3])), int(rng.choice([16, 8, 32, 5]))
    inp = rng.uniform(-1, 1, (n, hh, ww, ci)).astype(f32)   # same box as the output: the generator pads with zeros
    dw, pw, bias = rng.uniform(-1, 1, (3, 3, ci, 1)).astype(f32), rng.uniform(-1, 1, (ci, co)).astype(f32), rng.uniform(-1, 1, (co,)).astype(f32)
    o = hl.Buffer(np.zeros((n, hh, ww, co), f32))
    hl.depthwise_separable_conv(hl.Buffer(inp), hl.Buffer(dw), hl.Buffer(pw), hl.Buffer(bias), o)
    return f"n={n} {ww}x{hh} ci={ci} co={co}", same(o.numpy(), oracle.depthwise_separable_conv(inp, dw, pw, bias))


def _conv_data(rng, ci_choices=(32, 64, 128, 192)):
    n, hh, ww = int(rng.integers(1, 4)), rdim(rng, 1, 60), rdim(rng, 1, 130)     # W <= 125: input-linear kernel, wider: im2col kernel
    ci, co = int(rng.choice(ci_choices)), int(rng.choice([128, 256]))
    inp = rng.uniform(-1, 1, (n, hh + 2, ww + 2, ci)).astype(f32)
    filt, bias = rng.uniform(-1, 1, (ci, 3, 3, co)).astype(f32), rng.uniform(-1, 1, (co,)).astype(f32)
    return n, hh, ww, ci, co, inp, filt, bias


def case_conv_layer(rng):
    n, hh, ww, ci, co, inp, filt, bias = _conv_data(rng)
    o = hl.Buffer(np.zeros((n, hh, ww, co), f32))
    hl.conv_layer(hl.Buffer(inp), hl.Buffer(filt), hl.Buffer(bias), o)
    return f"n={n} {ww}x{hh} ci={ci} co={co}", same(o.numpy(), oracle.conv_layer(inp, filt, bias))


def case_conv_layer_bf16(rng):
    """tolerance, not bit-exactness: bf16 operands, f32 accumulation in hardware order (tests/test_conv_layer.py); the bf16
    entry point wants input channels in multiples of 64"""
    n, hh, ww, ci, co, inp, filt, bias = _conv_data(rng, (64, 128, 192))
    o = hl.Buffer(np.zeros((n, hh, ww, co), f32))
    hl.conv_layer_bf16(hl.Buffer(inp), hl.Buffer(filt), hl.Buffer(bias), o)
    want, mag = oracle.conv_layer_bf16(inp, filt, bias)
    err = np.abs(o.numpy().astype(np.float64) - want.astype(np.float64))
    return f"n={n} {ww}x{hh} ci={ci} co={co}", bool((err <= 2e-6 * mag.astype(np.float64) + 1e-6).all())


CASES = {k[5:]: v for k, v in list(globals().items()) if k.startswith("case_")}


def stress(args, only):
    """Host threads calling different pipelines at once: the caches (remap tables, filter images, camera set-up), the
    allocation cache and the per-stream arenas under contention.  Results must still be the oracle's."""
    import threading
    names = [n for n in CASES if not only or n in only]
    hip = hl.hip_runtime()
    log, lock = [], threading.Lock()
    counts = {n: [0, 0, 0] for n in names}
    t_end = time.time() + args.seconds

    def worker(i):
        rng = np.random.default_rng(args.seed * 7919 + i)
        stream = None
        if i % 2 == 1:
            import ctypes
            stream = ctypes.c_void_p()
            assert hip.hipStreamCreateWithFlags(ctypes.byref(stream), 1) == 0
            hl.set_stream(stream.value)
        while time.time() < t_end:
            name = names[int(rng.integers(0, len(names)))]
            try:
                desc, ok = CASES[name](rng)
                with lock:
                    counts[name][0] += 1
                    if not ok:
                        counts[name][1] += 1
                        log.append(f"thread {i}: {name}: MISMATCH {desc}")
            except Exception as e:  # noqa: BLE001
                with lock:
                    counts[name][0] += 1
                    counts[name][2] += 1
                    log.append(f"thread {i}: {name}: EXCEPTION {type(e).__name__}: {str(e)[:200]}")
        if stream is not None:
            hl.set_stream(None)
            hip.hipStreamSynchronize(stream)

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(args.threads)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for line in log[:20]:
        print("  " + line)
    bad = 0
    for n in names:
        c = counts[n]
        bad += c[1] + c[2]
        print(f"{n}: {c[0]} cases, {c[1]} mismatches, {c[2]} exceptions")
    print(f"STRESS ({args.threads} threads) " + ("CLEAN" if bad == 0 else f"FOUND {bad}"))
    return 0 if bad == 0 else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--seconds", type=float, default=20.0, help="time budget per pipeline")
    ap.add_argument("--only", default="")
    ap.add_argument("--threads", type=int, default=1, help="> 1: concurrency stress — that many host threads draw random cases of ALL "
                    "pipelines for --seconds in total, every second thread on a stream of its own")
    ap.add_argument("--frame-queue", action="store_true", help="run the cases with the calling thread on a frame-queue stream "
                    "(halide_hip_partition_stream(1, 4)): local_laplacian takes its throughput geometry there — one ll_down01e "
                    "workgroup per CU, fewer and taller units, taller ll_up0h tiles, non-temporal frame accesses")
    args = ap.parse_args()
    only = [s for s in args.only.split(",") if s]
    if args.threads > 1:
        return stress(args, only)
    if args.frame_queue:
        q = hl.partition_stream(1, 4)
        assert q, "the device refused a frame-queue stream"
        hl.set_stream(q)
    bad = 0
    for name, fn in CASES.items():
        if only and name not in only:
            continue
        rng = np.random.default_rng(args.seed * 1000 + sum(map(ord, name)))
        n = fails = errors = 0
        t0 = time.time()
        while time.time() - t0 < args.seconds:
            try:
                desc, ok = fn(rng)
            except Exception as e:  # a case the entry point rejects is a finding too: print it
                errors += 1
                if errors <= 5:
                    print(f"  {name}: EXCEPTION {type(e).__name__}: {str(e)[:200]}", flush=True)
                n += 1
                continue
            n += 1
            if not ok:
                fails += 1
                if fails <= 8:
                    print(f"  {name}: MISMATCH {desc}", flush=True)
        bad += fails + errors
        print(f"{name}: {n} cases, {fails} mismatches, {errors} exceptions", flush=True)
    print("FUZZ " + ("CLEAN" if bad == 0 else f"FOUND {bad}"))
    return 0 if bad == 0 else 1


if __name__ == "__main__":
    sys.exit(main())
