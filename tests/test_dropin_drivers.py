"""Drop-in proof: the reference's own app drivers (apps/<app>/process.cpp …), compiled UNMODIFIED from
/root/reference against libhlmi.so + include/aot/*.h (recipe: oracle/ref.mk, outputs in oracle/_ref/),
run on the GPU box and produce results identical to the oracle's.  The binaries are prebuilt in the dev
container (the GPU box has no /root/reference); tests skip if they are absent."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_BIN = os.path.join(ROOT, "oracle", "_ref")


def _exe(name):
    p = os.path.join(REF_BIN, name)
    if not os.path.exists(p):
        pytest.skip(f"oracle/_ref/{name} not built")
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
