#!/usr/bin/env python3
"""Pin the CPU oracle against a REAL Halide build (row c of SURVEY.md §8: "parity unpinned" until somebody with LLVM runs this).

Halide cannot be built in this repository's container (no LLVM), so the float pipelines are only pinned against the oracle's own
restatement.  Whoever has a Halide checkout with its apps built can close that in one command:

    python scripts/pin_against_halide.py inputs                # writes the seeded inputs + tests/golden/halide/manifest.json
    HALIDE_RUNGEN_DIR=<dir with <app>.rungen binaries> bash scripts/pin_against_halide.sh
    python -m pytest tests/test_reference_goldens.py -q        # compares the oracle with what Halide produced

(`python scripts/pin_against_halide.py commands` / `driver_commands` print the command lines instead of running them.  With
HALIDE_DRIVER_DIR=<dir with harris_filter, lens_blur_process, bgu_filter of the same build> the script also runs the three
pipelines RunGen cannot drive through the apps' own drivers: the "driver_cases" of the manifest.)

File layout: every buffer is a .npy in the REFERENCE's layout (tools/halide_image_io.h:1313-1470: the shape tuple lists the
Halide extents, dimension 0 first, over a payload with dimension 0 innermost) — exactly what `<app>.rungen name=file.npy` reads
and `--output=`-style `name=file.npy` writes (tools/RunGenMain.cpp:41-190)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.environ.get("HLMI_PIN_DIR") or os.path.join(ROOT, "tests", "golden", "halide")   # HLMI_PIN_DIR: a scratch copy (tests)


def save_halide_npy(path, arr):
    """numpy array (slowest axis first) -> the reference's .npy: same bytes, shape tuple reversed."""
    arr = np.ascontiguousarray(arr)
    np.save(path, arr.reshape(arr.shape[::-1]))


def load_halide_npy(path, np_shape):
    a = np.load(path)
    assert a.shape == tuple(np_shape[::-1]), (path, a.shape, np_shape)
    return a.reshape(np_shape)


def cases():
    """name -> dict(rungen=<generator name>, args=[(arg name, array or scalar), ...] in the generator's order, output=(name,
    numpy shape, dtype), oracle=callable on tests/oracle_lib -> array, exact=bool).  Inputs are seeded and small: the oracle
    finishes each in seconds; sizes respect every generator's minimum extents."""
    r = lambda seed: np.random.default_rng(seed)
    f32 = lambda seed, shape: r(seed).random(shape, dtype=np.float32)
    u16 = lambda seed, shape: r(seed).integers(0, 65536, shape, dtype=np.uint16)
    m3 = np.array([[1.6697, -0.2693, -0.4004, -42.4346], [-0.3576, 1.0615, 1.5949, -37.1158], [-0.2175, -1.8751, 6.9640, -26.6970]], np.float32)
    m7 = np.array([[2.2997, -0.4478, 0.1706, -39.0923], [-0.3826, 1.5906, -0.2080, -25.4311], [-0.0888, -0.7344, 2.2832, -20.0826]], np.float32)
    c = {}
    yy, xx = np.mgrid[0:240, 0:320].astype(np.float32)
    base = (np.sin(xx / 41.0) + np.cos(yy / 23.0) + 2.2) / 4.4
    smooth = np.clip(np.stack([base * 65535, base * 52000, base[::-1] * 46000]) + r(21).normal(0, 900, (3, 240, 320)), 0, 65535).astype(np.uint16)
    for tag, img in (("smooth", smooth), ("noise", u16(11, (3, 240, 320)))):
        c[f"local_laplacian_{tag}_320x240"] = dict(
            rungen="local_laplacian", args=[("input", img), ("levels", 8), ("alpha", 1.0 / 7.0), ("beta", 1.0)], output=("output", (3, 240, 320), np.uint16),
            oracle=lambda o, img=img: o.local_laplacian(img, 8, np.float32(1.0 / 7.0), 1.0), exact=True)
    b_in = u16(1, (52, 72))
    c["halide_blur_70x50"] = dict(rungen="halide_blur", args=[("input", b_in)], output=("blur_y", (50, 70), np.uint16), oracle=lambda o: o.blur(b_in), exact=True)
    s_in = u16(2, (60, 80))
    c["stencil_chain_80x60"] = dict(rungen="stencil_chain", args=[("input", s_in)], output=("output", (60, 80), np.uint16), oracle=lambda o: o.stencil_chain(s_in), exact=True)
    g_in = f32(3, (72, 96))
    c["bilateral_grid_96x72"] = dict(rungen="bilateral_grid", args=[("input", g_in), ("r_sigma", 0.1)], output=("bilateral_grid", (72, 96), np.float32),
                                     oracle=lambda o: o.bilateral_grid(g_in, 0.1), exact=False)
    n_in = f32(4, (3, 30, 40))
    c["nl_means_40x30"] = dict(rungen="nl_means", args=[("input", n_in), ("patch_size", 7), ("search_area", 7), ("sigma", 0.12)],
                               output=("non_local_means", (3, 30, 40), np.float32), oracle=lambda o: o.nl_means(n_in, 7, 7, 0.12), exact=False)
    ci, cf, cb = r(5).uniform(-1, 1, (2, 9, 11, 32)).astype(np.float32), r(6).uniform(-1, 1, (32, 3, 3, 128)).astype(np.float32), r(7).uniform(-1, 1, 128).astype(np.float32)
    c["conv_layer_2x7x9_32to128"] = dict(rungen="conv_layer", args=[("input", ci), ("filter", cf), ("bias", cb)], output=("relu", (2, 7, 9, 128), np.float32),
                                         oracle=lambda o: o.conv_layer(ci, cf, cb), exact=False)
    raw = r(8).integers(0, 1024, (152, 200), dtype=np.uint16)
    c["camera_pipe_160x120"] = dict(rungen="camera_pipe", args=[("input", raw), ("matrix_3200", m3), ("matrix_7000", m7), ("color_temp", 3700.0), ("gamma", 2.0),
                                                               ("contrast", 50.0), ("sharpen_strength", 1.0), ("blackLevel", 25), ("whiteLevel", 1023)],
                                    output=("processed", (3, 120, 160), np.uint8),
                                    oracle=lambda o: o.camera_pipe(raw, m3, m7, 3700.0, 2.0, 50.0, 1.0, 25, 1023, 160, 120), exact=True)
    d_in, d_dw, d_pw, d_b = (r(9).uniform(-1, 1, (2, 6, 7, 8)).astype(np.float32), r(10).uniform(-1, 1, (3, 3, 8, 1)).astype(np.float32),
                             r(11).uniform(-1, 1, (8, 5)).astype(np.float32), r(12).uniform(-1, 1, 5).astype(np.float32))
    c["depthwise_separable_conv_2x6x7"] = dict(rungen="depthwise_separable_conv",
                                               args=[("input", d_in), ("depthwise_filter", d_dw), ("pointwise_filter", d_pw), ("bias", d_b)],
                                               output=("output", (2, 6, 7, 5), np.float32), oracle=lambda o: o.depthwise_separable_conv(d_in, d_dw, d_pw, d_b), exact=False)
    u_in = f32(13, (3, 40, 50)) * np.float32(0.9) + np.float32(0.05)
    c["unsharp_50x40"] = dict(rungen="unsharp", args=[("input", u_in)], output=("output", (3, 40, 50), np.float32), oracle=lambda o: o.unsharp(u_in), exact=False)
    mf_in = f32(17, (3, 64, 70))
    c["max_filter_70x64"] = dict(rungen="max_filter", args=[("input", mf_in)], output=("output", (3, 64, 70), np.float32), oracle=lambda o: o.max_filter(mf_in), exact=True)
    h_in = r(14).integers(0, 256, (3, 60, 90), dtype=np.uint8)
    c["hist_90x60"] = dict(rungen="hist", args=[("input", h_in)], output=("output", (3, 60, 90), np.uint8), oracle=lambda o: o.hist(h_in), exact=True)
    # harris, lens_blur and bgu are NOT here: harris's output region starts at (3, 3) (apps/harris/filter.cpp:26 sets the output's
    # min), which RunGen's command line cannot express, and the other two have drivers that build part of their input themselves.
    # They are pinned through the apps' OWN drivers, which write their result to a file: driver_cases() below.
    rgba = f32(9, (4, 21, 34))
    rgba[3][r(10).random((21, 34)) < 0.4] = 0
    c["interpolate_34x21"] = dict(rungen="interpolate", args=[("input", rgba)], output=("output", (3, 21, 34), np.float32), oracle=lambda o: o.interpolate(rgba), exact=False)
    i_in = f32(16, (3, 20, 30))
    c["iir_blur_30x20"] = dict(rungen="iir_blur", args=[("input", i_in), ("alpha", 0.3)], output=("output", (3, 20, 30), np.float32), oracle=lambda o: o.iir_blur(i_in, 0.3), exact=False)
    return c


def scene8(w, h, seed, ch=3):
    """a smooth 8-bit test scene + noise (the drivers read images: PPM is the one format tools/halide_image_io.h reads without
    libpng / libjpeg)"""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    base = (np.sin(xx / 19.0 + seed) + np.cos(yy / 13.0) + 2.2) / 4.4
    img = np.stack([base * (255 - 20 * c) for c in range(ch)]) + rng.normal(0, 7, (ch, h, w))
    return np.clip(img, 0, 255).astype(np.uint8)


def write_ppm8(path, img):   # (3, H, W) uint8 -> binary PPM
    c, h, w = img.shape
    with open(path, "wb") as f:
        f.write(b"P6\n%d %d\n255\n" % (w, h))
        f.write(np.ascontiguousarray(img.transpose(1, 2, 0)).tobytes())


def load_mat_planar(path):
    """a .mat the reference's save_mat wrote (tools/halide_image_io.h:1916-2060: one array, x fastest) -> numpy, slowest axis first"""
    import scipy.io
    saved = scipy.io.loadmat(path)
    (name,) = [k for k in saved if not k.startswith("__")]
    a = saved[name]
    return np.ascontiguousarray(a.transpose(tuple(range(a.ndim))[::-1]))


def bgu_filter_cpp_low_res_pair(hi):
    """apps/bgu/filter.cpp:37-87 in float32, operator by operator: the 8x8 box downsample (sy outer, sx inner) and the
    "straw-man" sharpen + smoothstep + vignette the driver wants transferred to full resolution."""
    f32 = np.float32
    _, H, W = hi.shape
    lw, lh = W // 8, H // 8
    lo = np.zeros((3, lh, lw), f32)
    for sy in range(8):
        for sx in range(8):
            lo = lo + hi[:, sy::8, sx::8][:, :lh, :lw]
    lo = lo / f32(64)
    p = np.pad(lo, ((0, 0), (1, 1), (1, 1)), mode="edge")
    nb = ((p[:, 1:-1, :-2] + p[:, 1:-1, 2:]) + p[:, :-2, 1:-1]) + p[:, 2:, 1:-1]
    val = f32(2) * lo - nb / f32(4)
    yy, xx = np.mgrid[0:lh, 0:lw]
    edge = (xx == 0) | (xx == lw - 1) | (yy == 0) | (yy == lh - 1)
    val = np.where(edge[None], lo, val).astype(f32)
    boosted = (val * val) * (f32(3) - f32(2) * val)
    r = f32(min(W // 16, H // 16))
    mx, my = (xx - W // 16).astype(f32) / r, (yy - H // 16).astype(f32) / r
    mask = np.sqrt(mx * mx + my * my).astype(f32)[None]
    val = val * mask + boosted * (f32(1) - mask)
    val = val * ((f32(2) - mask) / f32(2))
    return lo, np.maximum(np.minimum(val, f32(1)), f32(0)).astype(f32)


def driver_cases():
    """The three pipelines RunGen cannot drive, through the apps' own drivers (each writes its output image; `.mat` keeps float32
    exact, tools/halide_image_io.h:1916-2060).  name -> dict(driver=<binary: the CMake target name of the app's driver, which is also
    what oracle/ref.mk calls this repository's build of the same unmodified source>, source=<reference file>, image=(file, u8 array),
    argv=[...] with {IN} / {OUT}, output_numpy_shape, oracle=callable on tests/oracle_lib -> array, note)."""
    c = {}
    h_img = scene8(200, 120, 8)
    c["harris_driver_200x120"] = dict(
        driver="harris_filter", source="apps/harris/filter.cpp:22-37", image=("harris_200x120.ppm", h_img), argv=["{IN}", "{OUT}"],
        output_numpy_shape=(114, 194),    # the driver's output is (W - 6) x (H - 6) at min (3, 3)
        oracle=lambda o: o.harris(h_img.astype(np.float32) / np.float32(255.0)), note="float32 response, region [3, W-3) x [3, H-3)")
    l_img = scene8(160, 100, 21)
    c["lens_blur_driver_160x100"] = dict(
        driver="lens_blur_process", source="apps/lens_blur/process.cpp:16-58", image=("lens_blur_160x100.ppm", l_img),
        argv=["{IN}", "32", "13", "0.5", "32", "1", "{OUT}"], output_numpy_shape=(3, 100, 160),
        oracle=lambda o: o.lens_blur(l_img, l_img, 32, 13, 0.5, 32),
        note="the same image as left and right view (process.cpp:26-27); depends on random_float()'s definition tag (scripts/lens_blur_tag.md)")
    b_img = scene8(256, 192, 31)
    b_hi = b_img.astype(np.float32) / np.float32(255.0)
    b_lo, b_lo_out = bgu_filter_cpp_low_res_pair(b_hi)
    c["bgu_driver_256x192"] = dict(
        driver="bgu_filter", source="apps/bgu/filter.cpp:17-108", image=("bgu_256x192.ppm", b_img), argv=["{IN}", "{OUT}"],
        output_numpy_shape=(3, 192, 256), oracle=lambda o, variant=0: o.bgu(1.0 / 8.0, 16, b_lo, b_lo_out, b_hi, variant=variant),
        note="the driver builds the low-res pair itself on the host; fast_inverse is target-dependent (x86: the rcpss estimate, oracle variant 1)")
    return c


def write_inputs():
    os.makedirs(os.path.join(GOLD, "inputs"), exist_ok=True)
    manifest = {}
    for name, c in cases().items():
        argv = []
        for an, v in c["args"]:
            if isinstance(v, np.ndarray):
                rel = os.path.join("inputs", f"{name}__{an}.npy")
                save_halide_npy(os.path.join(GOLD, rel), v)
                argv.append(f"{an}={{GOLD}}/{rel}")
            else:
                argv.append(f"{an}={float(np.float32(v))!r}" if isinstance(v, float) else f"{an}={v}")
        on, oshape, odt = c["output"]
        argv.append(f"{on}={{GOLD}}/{name}.npy")
        argv.append("--output_extents=[" + ",".join(str(e) for e in oshape[::-1]) + "]")
        manifest[name] = {"rungen": c["rungen"] + ".rungen", "argv": argv, "output": f"{name}.npy", "output_numpy_shape": list(oshape),
                          "output_dtype": np.dtype(odt).name, "exact_expected": c["exact"]}
    drivers = {}
    for name, c in driver_cases().items():
        fn, img = c["image"]
        write_ppm8(os.path.join(GOLD, "inputs", fn), img)
        drivers[name] = {"driver": c["driver"], "source": c["source"],
                         "argv": [a.replace("{IN}", "{GOLD}/inputs/" + fn).replace("{OUT}", "{GOLD}/" + name + ".mat") for a in c["argv"]],
                         "output": name + ".mat", "output_numpy_shape": list(c["output_numpy_shape"]), "output_dtype": "float32", "note": c["note"]}
    manifest["driver_cases"] = drivers   # (the one key of the manifest that is not a RunGen case)
    with open(os.path.join(GOLD, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1)
    return manifest


def commands(manifest, rungen_dir="$HALIDE_RUNGEN_DIR"):
    return [" ".join([f"{rungen_dir}/{m['rungen']}"] + [a.replace("{GOLD}", GOLD) for a in m["argv"]])
            for k, m in manifest.items() if k != "driver_cases"]


def driver_commands(manifest, driver_dir="$HALIDE_DRIVER_DIR"):
    return [" ".join([f"{driver_dir}/{m['driver']}"] + [a.replace("{GOLD}", GOLD) for a in m["argv"]]) for m in manifest["driver_cases"].values()]


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "inputs"
    m = write_inputs()
    if what == "commands":
        print("\n".join(commands(m)))
    elif what == "driver_commands":
        print("\n".join(driver_commands(m)))
    else:
        print(f"wrote {len(m) - 1} RunGen cases and {len(m['driver_cases'])} driver cases under {GOLD} (inputs/, manifest.json)")
