"""Drop-in proof: the reference's own app drivers (apps/<app>/process.cpp …), compiled UNMODIFIED from
/root/reference against libhlmi.so + include/aot/*.h (recipe: oracle/ref.mk, outputs in oracle/_ref/),
run on the GPU box and produce results identical to the oracle's.  The binaries are prebuilt in the dev
container by __graft_entry__.build() (the GPU box has no /root/reference) and travel with the snapshot: a missing
binary is a FAILURE — a box without them must not go green on skips."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_BIN = os.path.join(ROOT, "oracle", "_ref")


@pytest.fixture(autouse=True)
def _oracle_in_the_form_of_the_library_the_binaries_link(linked_library_canon):
    yield



def _exe(name):
    p = os.path.join(REF_BIN, name)
    assert os.path.exists(p), (f"oracle/_ref/{name} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` where "
                               "/root/reference is present (make -C oracle ref)")
    return p


def write_ppm8(path, img):  # img: (3, H, W) uint8
    c, h, w = img.shape
    with open(path, "wb") as f:
        f.write(b"P6\n%d %d\n255\n" % (w, h))
        f.write(np.ascontiguousarray(img.transpose(1, 2, 0)).tobytes())


def read_pnm16(path):
    with open(path, "rb") as f:
        data = f.read()
    toks, pos = [], 0
    while len(toks) < 4:
        while data[pos:pos + 1].isspace():
            pos += 1
        s = pos
        while not data[pos:pos + 1].isspace():
            pos += 1
        toks.append(data[s:pos])
    pos += 1
    magic, w, h, maxv = toks[0], int(toks[1]), int(toks[2]), int(toks[3])
    ch = 3 if magic == b"P6" else 1
    dt = np.dtype(">u2") if maxv > 255 else np.uint8
    arr = np.frombuffer(data, dt, count=w * h * ch, offset=pos).reshape(h, w, ch)
    return arr.transpose(2, 0, 1).astype(np.uint16 if maxv > 255 else np.uint8)


@pytest.mark.gpu
def test_local_laplacian_process_cpp(tmp_path, oracle):
    exe = _exe("local_laplacian_process")
    rng = np.random.default_rng(4)
    yy, xx = np.mgrid[0:240, 0:384]
    base = (np.sin(xx / 31.0) + np.cos(yy / 17.0) + 2.2) / 4.4
    img8 = np.clip(np.stack([base * 255, base * 200, base[::-1] * 180]) + rng.normal(0, 6, (3, 240, 384)), 0, 255)
    img8 = img8.astype(np.uint8)
    src, dst = str(tmp_path / "in.ppm"), str(tmp_path / "out.ppm")
    write_ppm8(src, img8)
    # process.cpp:31 — local_laplacian(input, levels, alpha / (levels - 1), beta, output); argv: levels alpha beta iters
    r = subprocess.run([exe, src, "8", "1", "1", "3", dst], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "Success!" in r.stdout, r.stdout + r.stderr
    got = read_pnm16(dst)
    inp16 = img8.astype(np.uint16) * 0x0101  # load_and_convert_image u8 -> u16 (tools/halide_image_io.h:194-196)
    want = oracle.local_laplacian(inp16, 8, float(np.float32(1.0) / np.float32(7)), 1.0)
    assert np.array_equal(got, want)


def write_pgm(path, img, maxval):  # img: (H, W)
    h, w = img.shape
    with open(path, "wb") as f:
        f.write(b"P5\n%d %d\n%d\n" % (w, h, maxval))
        f.write(np.ascontiguousarray(img.astype(">u2" if maxval > 255 else np.uint8)).tobytes())


def _quantise(out_f32, maxval):
    # convert_and_save_image: float -> uint: lround(v * maxval) (tools/halide_image_io.h:180-182, :232-234)
    return np.floor(out_f32.astype(np.float32) * np.float32(maxval) + np.float32(0.5)).astype(np.uint16)


def _scene8(w, h, seed, ch):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    base = (np.sin(xx / 19.0 + seed) + np.cos(yy / 13.0) + 2.2) / 4.4
    img = np.stack([base * (255 - 20 * c) for c in range(ch)]) + rng.normal(0, 7, (ch, h, w))
    return np.clip(img, 0, 255).astype(np.uint8)


def _maxval(path):
    with open(path, "rb") as f:
        head = f.read(64).split()
    return int(head[3])


@pytest.mark.gpu
def test_bilateral_grid_filter_cpp(tmp_path, oracle):
    exe = _exe("bilateral_grid_filter")
    img8 = _scene8(320, 200, 1, 1)[0]
    src, dst = str(tmp_path / "in.pgm"), str(tmp_path / "out.pgm")
    write_pgm(src, img8, 255)
    r = subprocess.run([exe, src, dst, "0.1", "3"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "Success!" in r.stdout, r.stdout + r.stderr
    got = read_pnm16(dst)[0]
    inp = img8.astype(np.float32) / np.float32(255.0)  # u8 -> float (tools/halide_image_io.h:610-612)
    want = oracle.bilateral_grid(inp, float(np.float32(0.1)))
    assert np.array_equal(got, _quantise(want, _maxval(dst)))


@pytest.mark.gpu
def test_nl_means_process_cpp(tmp_path, oracle):
    exe = _exe("nl_means_process")
    img8 = _scene8(192, 120, 2, 3)
    src, dst = str(tmp_path / "in.ppm"), str(tmp_path / "out.ppm")
    write_ppm8(src, img8)
    r = subprocess.run([exe, src, "7", "7", "0.12", "2", dst], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "Success!" in r.stdout, r.stdout + r.stderr
    got = read_pnm16(dst)
    inp = img8.astype(np.float32) / np.float32(255.0)
    want = oracle.nl_means(inp, 7, 7, float(np.float32(0.12)))
    assert np.array_equal(got, _quantise(want, _maxval(dst)))


@pytest.mark.gpu
def test_stencil_chain_process_cpp(tmp_path, oracle):
    exe = _exe("stencil_chain_process")
    img8 = _scene8(256, 160, 3, 3)
    src, dst = str(tmp_path / "in.ppm"), str(tmp_path / "out.pgm")
    write_ppm8(src, img8)
    r = subprocess.run([exe, src, "3", dst], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "Success!" in r.stdout, r.stdout + r.stderr
    got = read_pnm16(dst)[0]
    red16 = img8[0].astype(np.uint16) * 0x0101  # process.cpp:26 takes channel 0 of the u16 conversion
    assert np.array_equal(got, oracle.stencil_chain(red16))


@pytest.mark.gpu
def test_conv_layer_process_cpp():
    exe = _exe("conv_layer_process")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "Success!" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_depthwise_separable_conv_process_cpp():
    exe = _exe("depthwise_separable_conv_process")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "Success!" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_unsharp_filter_cpp(tmp_path, oracle):
    exe = _exe("unsharp_filter")
    img8 = _scene8(200, 120, 5, 3)
    src, dst = str(tmp_path / "in.ppm"), str(tmp_path / "out.ppm")
    write_ppm8(src, img8)
    r = subprocess.run([exe, src, dst], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "Success!" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_max_filter_filter_cpp(tmp_path, oracle):
    """apps/max_filter/filter.cpp unmodified: 8-bit in -> float -> max_filter (+ _auto_schedule) -> 16-bit out.  A max of
    values k/255 is one of them, so the saved image is exactly the footprint max of the 8-bit input (x 257)."""
    exe = _exe("max_filter_filter")
    img8 = _scene8(180, 140, 6, 3)
    src, dst = str(tmp_path / "in.ppm"), str(tmp_path / "out.ppm")
    write_ppm8(src, img8)
    r = subprocess.run([exe, src, dst], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "Success!" in r.stdout, r.stdout + r.stderr
    got = read_pnm16(dst)
    assert np.array_equal(got, oracle.max_filter(img8.astype(np.float32)).astype(np.uint16) * 257)


@pytest.mark.gpu
def test_hist_filter_cpp(tmp_path, oracle):
    exe = _exe("hist_filter")
    img8 = _scene8(256, 160, 7, 3)
    src, dst = str(tmp_path / "in.ppm"), str(tmp_path / "out.ppm")
    write_ppm8(src, img8)
    r = subprocess.run([exe, src, dst], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "Success!" in r.stdout, r.stdout + r.stderr
    got = read_pnm16(dst)
    assert np.array_equal(got, oracle.hist(img8))


@pytest.mark.gpu
def test_harris_filter_cpp(tmp_path):
    exe = _exe("harris_filter")
    img8 = _scene8(200, 120, 8, 3)
    src, dst = str(tmp_path / "in.ppm"), str(tmp_path / "out.pgm")
    write_ppm8(src, img8)
    r = subprocess.run([exe, src, dst], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "Success!" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_interpolate_filter_cpp(tmp_path, oracle):
    """apps/interpolate/filter.cpp wants an RGBA image; without libpng in this image it is fed (and writes) MATLAB level-5
    ".mat" files, the one format of tools/halide_image_io.h (:1759-1915 load_mat, :1916-2060 save_mat) that carries a
    3-dimensional float image without libpng — which also keeps the result exact: the saved output must be the oracle's
    bit for bit."""
    import scipy.io
    exe = _exe("interpolate_filter")
    rng = np.random.default_rng(12)
    rgba = rng.random((4, 96, 160), dtype=np.float32)
    rgba[3] = (rng.random((96, 160)) > 0.7).astype(np.float32) * rgba[3]     # sparse alpha: the pull-push has holes to fill
    src, dst = str(tmp_path / "rgba.mat"), str(tmp_path / "result.mat")
    # MATLAB arrays are column-major: an array of shape (W, H, C) stores x fastest, then y, then c — Halide's planar layout
    scipy.io.savemat(src, {"rgba": np.asfortranarray(rgba.transpose(2, 1, 0))}, format="5", do_compression=False)
    r = subprocess.run([exe, src, dst], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "Success!" in r.stdout, r.stdout + r.stderr
    saved = scipy.io.loadmat(dst)
    (name,) = [k for k in saved if not k.startswith("__")]
    got = np.ascontiguousarray(saved[name].transpose(2, 1, 0))
    want = oracle.interpolate(rgba)
    assert got.dtype == np.float32 and got.shape == want.shape and np.array_equal(got.view(np.uint32), want.view(np.uint32))


@pytest.mark.gpu
def test_lens_blur_process_cpp(tmp_path, oracle):
    """apps/lens_blur/process.cpp:16-58: the same image as left and right view (process.cpp:26-27), result saved as .mat
    (exact float32): must be the oracle's bit for bit (with the random tag both sides default to)."""
    import scipy.io
    exe = _exe("lens_blur_process")
    img8 = _scene8(160, 100, 21, 3)
    src, dst = str(tmp_path / "in.ppm"), str(tmp_path / "result.mat")
    write_ppm8(src, img8)
    r = subprocess.run([exe, src, "32", "13", "0.5", "32", "2", dst], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "Success!" in r.stdout, r.stdout + r.stderr
    saved = scipy.io.loadmat(dst)
    (name,) = [k for k in saved if not k.startswith("__")]
    got = np.ascontiguousarray(saved[name].transpose(2, 1, 0))
    want = oracle.lens_blur(img8, img8, 32, 13, 0.5, 32)
    assert got.dtype == np.float32 and np.array_equal(got.view(np.uint32), want.view(np.uint32))


def _bgu_filter_cpp_low_res_pair(hi):
    """apps/bgu/filter.cpp:37-87 in float32, operator by operator: the 8x8 box downsample (sy outer, sx inner) and the
    "straw-man" sharpen + smoothstep + vignette it wants transferred to full resolution."""
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


@pytest.mark.gpu
def test_bgu_filter_cpp(tmp_path, oracle):
    """apps/bgu/filter.cpp:17-108 unmodified: it builds the low-res pair itself on the host, calls bgu() and
    bgu_auto_schedule() under the benchmark harness and saves the full-res result — here as .mat (exact float32), which
    must be the oracle's result for the same pair, bit for bit."""
    import scipy.io
    exe = _exe("bgu_filter")
    img8 = _scene8(256, 192, 31, 3)
    src, dst = str(tmp_path / "in.ppm"), str(tmp_path / "result.mat")
    write_ppm8(src, img8)
    r = subprocess.run([exe, src, dst], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "Success!" in r.stdout, r.stdout + r.stderr
    assert "Manually-tuned time" in r.stdout and "Auto-scheduled time" in r.stdout
    saved = scipy.io.loadmat(dst)
    (name,) = [k for k in saved if not k.startswith("__")]
    got = np.ascontiguousarray(saved[name].transpose(2, 1, 0))
    hi = img8.astype(np.float32) / np.float32(255.0)              # tools/halide_image_io.h:610-612
    lo, lo_out = _bgu_filter_cpp_low_res_pair(hi)
    want = oracle.bgu(1.0 / 8.0, 16, lo, lo_out, hi)
    assert got.dtype == np.float32 and got.shape == want.shape
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), f"{np.count_nonzero(got != want)} of {got.size} differ"


# ---- the PNG path of tools/halide_image_io.h (load_png :856-940, save_png :952-1040), libpng calls over zlib (tests/cpp/png_shim)
def write_png(path, img, bit_depth=8):
    """img: (C, H, W) integer array, C in 1..4.  Scanline filters cycle through all five types (clause 9 of the PNG spec)."""
    import struct
    import zlib
    c, h, w = img.shape
    bps = bit_depth // 8
    rows = np.ascontiguousarray(img.transpose(1, 2, 0)).astype(">u2" if bps == 2 else np.uint8).reshape(h, -1).view(np.uint8)
    bpp, raw, prev = c * bps, bytearray(), np.zeros(rows.shape[1], np.int32)
    for y in range(h):
        cur = rows[y].astype(np.int32)
        left = np.concatenate([np.zeros(bpp, np.int32), cur[:-bpp]])
        upleft = np.concatenate([np.zeros(bpp, np.int32), prev[:-bpp]])
        ft = y % 5
        if ft == 0:
            pred = np.zeros_like(cur)
        elif ft == 1:
            pred = left
        elif ft == 2:
            pred = prev
        elif ft == 3:
            pred = (left + prev) >> 1
        else:
            pp = left + prev - upleft
            pa, pb, pc = np.abs(pp - left), np.abs(pp - prev), np.abs(pp - upleft)
            pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, prev, upleft))
        raw.append(ft)
        raw += bytes(((cur - pred) & 255).astype(np.uint8))
        prev = cur

    def chunk(t, b):
        return struct.pack(">I", len(b)) + t + b + struct.pack(">I", zlib.crc32(t + b))
    color = {1: 0, 2: 4, 3: 2, 4: 6}[c]
    z = zlib.compress(bytes(raw), 6)
    with open(path, "wb") as f:   # two IDAT chunks: the reader must concatenate them
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, bit_depth, color, 0, 0, 0)) +
                chunk(b"IDAT", z[:len(z) // 2]) + chunk(b"IDAT", z[len(z) // 2:]) + chunk(b"IEND", b""))


def read_png(path):
    """-> (C, H, W) uint8 / uint16."""
    import struct
    import zlib
    data = open(path, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, z = 8, b""
    while pos < len(data):
        n, t = struct.unpack(">I", data[pos:pos + 4])[0], data[pos + 4:pos + 8]
        body = data[pos + 8:pos + 8 + n]
        assert zlib.crc32(t + body) == struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])[0]
        if t == b"IHDR":
            w, h, depth, color = struct.unpack(">IIBB", body[:10])
        elif t == b"IDAT":
            z += body
        pos += 12 + n
    c, bps = {0: 1, 4: 2, 2: 3, 6: 4}[color], depth // 8
    raw = np.frombuffer(zlib.decompress(z), np.uint8).reshape(h, 1 + w * c * bps)
    assert not raw[:, 0].any()       # the shim writes filter type 0
    px = raw[:, 1:].copy().view(">u2" if bps == 2 else np.uint8).reshape(h, w, c)
    return px.transpose(2, 0, 1).astype(np.uint16 if bps == 2 else np.uint8)


@pytest.mark.gpu
def test_local_laplacian_process_cpp_through_png(tmp_path, oracle):
    """The reference's driver built WITHOUT -DHALIDE_NO_PNG: 8-bit RGB PNG in (all five scanline filters, split IDAT), 16-bit
    PNG out (process.cpp:22, :50), through the reference's own load_png / save_png."""
    exe = _exe("local_laplacian_process_png")
    img8 = _scene8(200, 120, 31, 3)
    src, dst = str(tmp_path / "in.png"), str(tmp_path / "out.png")
    write_png(src, img8)
    r = subprocess.run([exe, src, "8", "1", "1", "2", dst], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "Success!" in r.stdout, r.stdout + r.stderr
    got = read_png(dst)
    want = oracle.local_laplacian(img8.astype(np.uint16) * 0x0101, 8, float(np.float32(1.0) / np.float32(7)), 1.0)
    assert got.dtype == np.uint16 and np.array_equal(got, want)


@pytest.mark.gpu
def test_interpolate_filter_cpp_through_png(tmp_path, oracle):
    """apps/interpolate/filter.cpp with the RGBA PNG it was written for (16-bit here), result saved as PNG."""
    exe = _exe("interpolate_filter_png")
    rng = np.random.default_rng(13)
    rgba = rng.integers(0, 65536, (4, 72, 100)).astype(np.uint16)
    rgba[3][rng.random((72, 100)) < 0.6] = 0
    src, dst = str(tmp_path / "rgba.png"), str(tmp_path / "out.png")
    write_png(src, rgba, 16)
    r = subprocess.run([exe, src, dst], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "Success!" in r.stdout, r.stdout + r.stderr
    got = read_png(dst)
    inp = (rgba.astype(np.float32) / np.float32(65535))          # load_and_convert_image: u16 -> float (halide_image_io.h:186-189)
    want = oracle.interpolate(inp)
    maxval = 65535 if got.dtype == np.uint16 else 255
    assert np.array_equal(got, _quantise(want, maxval).astype(got.dtype))


@pytest.mark.gpu
def test_iir_blur_filter_cpp(tmp_path):
    exe = _exe("iir_blur_filter")
    img8 = _scene8(192, 130, 9, 3)
    src, dst = str(tmp_path / "in.ppm"), str(tmp_path / "out.ppm")
    write_ppm8(src, img8)
    r = subprocess.run([exe, src, dst], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "Success!" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_camera_pipe_process_cpp(tmp_path, oracle):
    exe = _exe("camera_pipe_process")
    from test_camera_pipe import M3200, M7000, _raw
    raw = _raw(352, 280, seed=6)
    src, dst = str(tmp_path / "raw.pgm"), str(tmp_path / "out.ppm")
    write_pgm(src, raw, 65535)
    r = subprocess.run([exe, src, "3700", "2.0", "50", "1.0", "2", dst], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "Success!" in r.stdout, r.stdout + r.stderr
    got = read_pnm16(dst)
    ow, oh = ((352 - 32) // 32) * 32, ((280 - 24) // 32) * 32  # process.cpp:34
    want = oracle.camera_pipe(raw, M3200, M7000, 3700.0, 2.0, 50.0, 1.0, 25, 1023, ow, oh)
    assert np.array_equal(got, want)
