"""hlmi_rungen: the RunGen-compatible runner (reference: tools/RunGenMain.cpp usage :41-190, output format
tools/RunGen.h:1285-1298).  It reaches the pipelines only through `<name>_argv` / `<name>_metadata` and the
bounds-query protocol, like the reference's RunGen."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RUNGEN = os.path.join(ROOT, "halide_amd", "bin", "hlmi_rungen")


def _run(*args, check=True):
    p = subprocess.run([RUNGEN, *args], capture_output=True, text=True, timeout=300)
    if check:
        assert p.returncode == 0, p.stdout + p.stderr
    return p


def test_describe_lists_the_generator_arguments():
    out = _run("--name=local_laplacian", "--describe").stdout
    assert 'Input "input" is of type Buffer<uint16> with 3 dimensions' in out
    assert 'Input "levels" is of type int32' in out and 'Input "alpha" is of type float32' in out
    assert 'Output "output" is of type Buffer<uint16> with 3 dimensions' in out
    out = _run("--name=nl_means", "--describe").stdout
    assert 'Input "sigma" is of type float32' in out and 'Output "non_local_means"' in out


def test_boolean_flags_take_values_like_rungens():
    """--flag, --flag=true and --flag=false (tools/RunGenMain.cpp:421-500); --skip_bad_environment in both of the reference's spellings."""
    want = _run("--name=hist", "--describe").stdout
    assert "Filter name" in want
    assert _run("--name=hist", "--describe=true", "--quiet", "--track_memory=false", "--skip_bad_environment").stdout == want
    p = _run("--name=hist", "--describe=false", check=False)
    assert p.returncode != 0 and "no value for buffer" in p.stderr           # not describing: it wants its inputs
    p = _run("--name=hist", "--describe=maybe", check=False)
    assert p.returncode != 0 and "Invalid value for flag: describe" in p.stderr
    assert _run("--name=hist", "--describe", "--skip_bad_environement").stdout == want   # the spelling of RunGen's usage text (:179)
    p = _run("--name=hist", "--skip_bad_env", check=False)
    assert p.returncode != 0 and "unknown flag" in p.stderr


@pytest.mark.gpu
def test_conversion_warnings_use_rungens_wording(tmp_path):
    """RunGen warns when a file's type or shape is not the argument's (tools/RunGen.h:433-477, :1116-1121); same sentences here."""
    rng = np.random.default_rng(2)
    rgb8 = rng.integers(0, 256, (24, 40, 3), dtype=np.uint8)
    with open(tmp_path / "c.ppm", "wb") as f:
        f.write(b"P6\n40 24\n255\n" + rgb8.tobytes())
    p = _run("--name=bilateral_grid", f"input={tmp_path / 'c.ppm'}", "r_sigma=0.1", f"bilateral_grid={tmp_path / 'o.pgm'}")
    assert 'Warning: Image for Input "input" has 3 dimensions, but only the first 2 were used; data loss may have occurred.' in p.stderr
    assert 'Warning: Image loaded for argument "input" is type uint8 but this argument expects type float32; data loss may have occurred.' in p.stderr
    assert 'Warning: Image for argument "bilateral_grid" is of type float32 but is being saved as type uint16; data loss may have occurred.' in p.stderr
    p = _run("--name=bilateral_grid", f"input={tmp_path / 'c.ppm'}", "r_sigma=0.1", f"bilateral_grid={tmp_path / 'o.jpg'}")
    assert 'Warning: Image for argument "bilateral_grid" is of type float32 but is being saved as type uint8; data loss may have occurred.' in p.stderr
    # (no --output_extents above: the output assumed the shape of the first input, tools/RunGen.h:1077-1090)
    assert open(tmp_path / "o.pgm", "rb").read(12).split()[:3] == [b"P5", b"40", b"24"]
    gray8 = rgb8[..., 0]
    with open(tmp_path / "g.pgm", "wb") as f:
        f.write(b"P5\n40 24\n255\n" + gray8.tobytes())
    p = _run("--name=hist", f"input={tmp_path / 'g.pgm'}", f"output={tmp_path / 'h.npy'}", check=False)   # a gray file for an RGB argument
    assert 'Warning: Image for Input "input" has 2 dimensions, but this argument requires at least 3 dimensions: adding dummy dimensions of extent 1.' in p.stderr
    assert "is type uint8 but" not in p.stderr                                  # u8 file, u8 argument


def _fixture_lib(tmp_path):
    """tests/cpp/runner_fixture_lib.cpp: two CPU stand-in pipelines, so that the runner's whole path runs here without a GPU."""
    so = tmp_path / "librunner_fixture.so"
    if not so.exists():
        subprocess.run(["g++", "-shared", "-fPIC", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "tests", "cpp", "runner_fixture_lib.cpp"), "-o", str(so)], check=True)
    return str(so)


def _run_fixture(tmp_path, *args):
    env = dict(os.environ, HLMI_LIB=_fixture_lib(tmp_path))
    p = subprocess.run([RUNGEN, *args], capture_output=True, text=True, timeout=120, env=env)
    assert p.returncode == 0, p.stdout + p.stderr
    return p


def test_runner_end_to_end_on_the_cpu_fixture(tmp_path):
    """Load (PPM into a 2-D float argument), RunGen's three conversion warnings, the output assuming the input's shape, a JPEG and a
    16-bit PGM written, --track_memory's line and --success — the whole path of the runner, on tests/cpp/runner_fixture_lib.cpp."""
    rng = np.random.default_rng(2)
    rgb8 = rng.integers(0, 256, (24, 40, 3), dtype=np.uint8)
    (tmp_path / "c.ppm").write_bytes(b"P6\n40 24\n255\n" + rgb8.tobytes())
    p = _run_fixture(tmp_path, "--name=fixture_copy", f"input={tmp_path / 'c.ppm'}", f"output={tmp_path / 'o.pgm'}", "--track_memory", "--success")
    assert 'Warning: Image for Input "input" has 3 dimensions, but only the first 2 were used; data loss may have occurred.' in p.stderr
    assert 'Warning: Image loaded for argument "input" is type uint8 but this argument expects type float32; data loss may have occurred.' in p.stderr
    assert 'Warning: Image for argument "output" is of type float32 but is being saved as type uint16; data loss may have occurred.' in p.stderr
    assert p.stdout.splitlines() == ["Maximum Halide memory: 0 bytes for output of 0.000915527 mpix.", "Success!"]
    raw = (tmp_path / "o.pgm").read_bytes()
    assert raw.startswith(b"P5\n40 24\n65535\n")
    got = np.frombuffer(raw[len(b"P5\n40 24\n65535\n"):], ">u2").reshape(24, 40)
    assert np.array_equal(got, rgb8[..., 0].astype(np.uint16) * 257)            # u8 -> float (/255) -> u16 (x 65535, rounded)
    p = _run_fixture(tmp_path, "--name=fixture_copy", f"input={tmp_path / 'c.ppm'}", f"output={tmp_path / 'o.jpg'}")
    assert 'is of type float32 but is being saved as type uint8' in p.stderr
    tool = _jpeg_tool(tmp_path)
    (tmp_path / "g.bin").write_bytes(rgb8[..., 0].tobytes())
    subprocess.run([tool, "encode", str(tmp_path / "g.bin"), "40", "24", "1", str(tmp_path / "ref.jpg"), "99"], check=True)
    assert (tmp_path / "o.jpg").read_bytes() == (tmp_path / "ref.jpg").read_bytes()


def test_runner_grows_a_loaded_input_and_runs_on_the_cpu_fixture(tmp_path):
    """fixture_shift reads one pixel right of and below its output, two pixels of box more: the 40 x 24 file is re-allocated as a
    42 x 26 input with the samples at their coordinates (RunGen's adapt_input_buffer), the output assumes 40 x 24."""
    g = np.random.default_rng(1).integers(0, 256, (24, 40), dtype=np.uint8)
    (tmp_path / "g.pgm").write_bytes(b"P5\n40 24\n255\n" + g.tobytes())
    p = _run_fixture(tmp_path, "--name=fixture_shift", f"input={tmp_path / 'g.pgm'}", f"output={tmp_path / 'o.pgm'}", "--verbose")
    assert "Input input: grown to the region the bounds query asks for" in p.stdout
    out = np.frombuffer((tmp_path / "o.pgm").read_bytes()[len(b"P5\n40 24\n255\n"):], np.uint8).reshape(24, 40)
    want = np.zeros((24, 40), np.uint8)
    want[:23, :39] = g[1:, 1:]
    assert np.array_equal(out, want)


def test_loaded_input_is_grown_to_the_region_the_bounds_query_asks_for(tmp_path):
    """RunGen's adapt_input_buffer (tools/RunGen.h:774-817): an input that does not cover its region is re-allocated on the region,
    the loaded samples copied in.  halide_blur reads two pixels beyond its output; without --output_extents the output assumes
    the input's shape (:1077-1090), so the 40 x 24 file becomes a 42 x 26 input.  (The shapes are settled by bounds queries, which
    need no device: the run itself then fails here for want of one.)"""
    rng = np.random.default_rng(3)
    np.save(tmp_path / "u.npy", rng.integers(0, 65536, (24, 40), dtype=np.uint16).reshape(40, 24))
    p = _run("--name=halide_blur", f"input={tmp_path / 'u.npy'}", f"blur_y={tmp_path / 'b.npy'}", "--verbose", check=False)
    assert "Input input: grown to the region the bounds query asks for" in p.stdout
    assert "Argument input: [ (0,42,1) (0,26,42) ]" in p.stdout and "Argument blur_y: [ (0,40,1) (0,24,40) ]" in p.stdout
    p = _run("--name=halide_blur", f"input={tmp_path / 'u.npy'}", f"blur_y={tmp_path / 'b.npy'}", "--output_extents=[38,22]", "--verbose", check=False)
    assert "grown" not in p.stdout and "Argument input: [ (0,40,1) (0,24,40) ]" in p.stdout


def test_unknown_pipeline_and_argument_are_errors():
    assert _run("--name=no_such_filter", "--describe", check=False).returncode != 0
    p = _run("--name=halide_blur", "bogus=1", "--describe", check=False)
    assert p.returncode != 0 and "unknown argument" in p.stderr


def test_argv0_basename_selects_the_pipeline(tmp_path):
    link = tmp_path / "stencil_chain.rungen"
    os.symlink(RUNGEN, link)
    p = subprocess.run([str(link), "--describe"], capture_output=True, text=True, env={**os.environ, "HLMI_LIB": os.path.join(ROOT, "halide_amd", "lib", "libhlmi.so")})
    assert p.returncode == 0 and 'Filter name: "stencil_chain"' in p.stdout


@pytest.mark.gpu
def test_estimate_then_auto_inside_a_pseudo_file(tmp_path):
    """`random:0:estimate_then_auto` — what --estimate_all stands for (tools/RunGenMain.cpp:479-486) — written out by hand."""
    a = _run("--name=hist", "--estimate_all", f"output={tmp_path / 'a.npy'}")
    b = _run("--name=hist", "--default_input_buffers=random:0:estimate_then_auto", "--output_extents=estimate", f"output={tmp_path / 'b.npy'}")
    assert a.returncode == 0 and b.returncode == 0
    assert np.array_equal(np.load(tmp_path / "a.npy"), np.load(tmp_path / "b.npy"))


@pytest.mark.gpu
def test_estimate_all_benchmark_parsable_output():
    out = _run("--name=local_laplacian", "--estimate_all", "--benchmarks=all", "--parsable_output", "--success").stdout
    keys = {l.split()[1] for l in out.splitlines() if l.startswith("local_laplacian ")}
    assert {"BEST_TIME_MSEC_PER_ITER", "SAMPLES", "ITERATIONS", "TIMING_ACCURACY", "THROUGHPUT_MPIX_PER_SEC", "HALIDE_TARGET"} <= keys
    assert "Success!" in out


@pytest.mark.gpu
def test_npy_round_trip_matches_the_direct_call(hl, oracle, tmp_path):
    rng = np.random.default_rng(3)
    inp = rng.integers(0, 65536, (3, 120, 200), dtype=np.uint16)
    # the reference's .npy convention (tools/halide_image_io.h:1414-1470): the shape tuple lists the HALIDE extents, x first, over
    # a payload with x innermost — the bytes of the (c, y, x) array under the header (W, H, C)
    np.save(tmp_path / "in.npy", inp.reshape(200, 120, 3))
    _run("--name=local_laplacian", f"input={tmp_path / 'in.npy'}", "levels=8", "alpha=0.14285714285714285", "beta=1",
         f"output={tmp_path / 'out.npy'}", "--output_extents=[200,120,3]")
    got = np.load(tmp_path / "out.npy")
    assert got.shape == (200, 120, 3)
    got = got.reshape(3, 120, 200)
    want = oracle.local_laplacian(inp, 8, np.float32(0.14285714285714285), 1.0)
    assert got.shape == want.shape and np.array_equal(got, want)


@pytest.mark.gpu
def test_bounds_query_sizes_auto_inputs_and_ppm_output(tmp_path):
    # blur reads (W+2) x (H+2): `auto` must come back from the bounds query as 66 x 50 for a 64 x 48 output
    p = _run("--name=halide_blur", "input=random:1:auto", f"blur_y={tmp_path / 'o.pgm'}", "--output_extents=[64,48]", "--verbose")
    assert "Argument input: [ (0,66,1) (0,50,66) ]" in p.stdout
    with open(tmp_path / "o.pgm", "rb") as f:
        assert f.read(2) == b"P5"


# ---- PNG files (halide_amd/tools/hlmi_png.h: the runner's own codec over zlib; reference: tools/halide_image_io.h:856-1040)
def _png_tool(tmp_path):
    exe = tmp_path / "png_codec_test"
    subprocess.run(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "halide_amd", "tools"),
                    os.path.join(ROOT, "tests", "cpp", "png_codec_test.cpp"), "-o", str(exe), "-lz"], check=True)
    return str(exe)


@pytest.mark.parametrize("channels,depth", [(1, 8), (2, 8), (3, 8), (4, 8), (1, 16), (3, 16), (4, 16)])
def test_png_codec_round_trip(tmp_path, channels, depth):
    """A file with all five scanline filters and a split IDAT stream (written by the independent Python encoder of
    tests/test_dropin_drivers.py) decodes to the same samples, and what the codec writes the Python decoder reads back."""
    from test_dropin_drivers import read_png, write_png
    rng = np.random.default_rng(channels * 100 + depth)
    img = rng.integers(0, 1 << depth, (channels, 37, 53)).astype(np.uint16 if depth == 16 else np.uint8)
    src, dst = str(tmp_path / "in.png"), str(tmp_path / "out.png")
    write_png(src, img, depth)
    p = subprocess.run([_png_tool(tmp_path), src, dst], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    assert p.stdout.split() == ["53", "37", str(channels), str(depth)]
    assert np.array_equal(read_png(dst), img)


def test_png_codec_rejects_damaged_files(tmp_path):
    from test_dropin_drivers import write_png
    tool = _png_tool(tmp_path)
    src = str(tmp_path / "in.png")
    write_png(src, np.zeros((3, 8, 8), np.uint8))
    data = bytearray(open(src, "rb").read())
    data[40] ^= 0x55                                   # inside the first IDAT chunk: its CRC no longer matches
    open(src, "wb").write(bytes(data))
    p = subprocess.run([tool, src, str(tmp_path / "o.png")], capture_output=True, text=True)
    assert p.returncode == 1 and "CRC" in p.stderr
    open(src, "wb").write(b"not a png at all")
    p = subprocess.run([tool, src, str(tmp_path / "o.png")], capture_output=True, text=True)
    assert p.returncode == 1 and "not a PNG" in p.stderr


@pytest.mark.gpu
def test_png_in_png_out_matches_the_oracle(oracle, tmp_path):
    """`input=foo.png output=bar.png` — the command the reference's RunGen accepts (tools/RunGenMain.cpp:100-112): 16-bit RGB
    in, 16-bit RGB out, bit-exact with the oracle on the decoded samples."""
    from test_dropin_drivers import read_png, write_png
    rng = np.random.default_rng(11)
    inp = rng.integers(0, 65536, (3, 90, 160), dtype=np.uint16)
    write_png(str(tmp_path / "in.png"), inp, 16)
    _run("--name=local_laplacian", f"input={tmp_path / 'in.png'}", "levels=8", "alpha=0.14285714285714285", "beta=1",
         f"output={tmp_path / 'out.png'}", "--output_extents=[160,90,3]")
    got = read_png(str(tmp_path / "out.png"))
    want = oracle.local_laplacian(inp, 8, np.float32(0.14285714285714285), 1.0)
    assert got.dtype == np.uint16 and np.array_equal(got, want)
    # an 8-bit file feeding the u16 input is rescaled x257 like the reference's image I/O (halide_image_io.h:79-240)
    inp8 = rng.integers(0, 256, (3, 90, 160), dtype=np.uint8)
    write_png(str(tmp_path / "in8.png"), inp8, 8)
    _run("--name=local_laplacian", f"input={tmp_path / 'in8.png'}", "levels=8", "alpha=0.14285714285714285", "beta=1",
         f"output={tmp_path / 'out8.png'}", "--output_extents=[160,90,3]")
    want8 = oracle.local_laplacian(inp8.astype(np.uint16) * 257, 8, np.float32(0.14285714285714285), 1.0)
    assert np.array_equal(read_png(str(tmp_path / "out8.png")), want8)


def _write_mat_like_the_reference(path, arr, name="m"):
    """A level-5 .mat file laid out the way tools/halide_image_io.h:1916-2100 writes one: 128-byte text header, one miMATRIX
    element = array flags, dimensions (dimension 0 first), name, real part; `arr` is given in numpy order (slowest first)."""
    import struct
    mi, mx = {np.dtype("float32"): (7, 7), np.dtype("uint16"): (4, 11), np.dtype("uint8"): (2, 9), np.dtype("float64"): (9, 6)}[arr.dtype]
    ext = list(arr.shape[::-1])
    nd = max(2, len(ext))
    ext_file = ext + [1] * (nd - len(ext))
    ext_file += [0] * (len(ext_file) & 1)
    nm = name.encode() + b"\0" * (-len(name) % 8)
    payload = arr.tobytes()
    pad = 7 - ((len(payload) - 1) & 7)
    text = b"MATLAB 5.0 MAT-file, produced by a test".ljust(124) + struct.pack("<H", 0x0100) + b"IM"
    body = struct.pack("<4I", 6, 8, mx, 1) + struct.pack("<2I", 5, 4 * len(ext)) + struct.pack(f"<{len(ext_file)}i", *ext_file)
    body += struct.pack("<2I", 1, len(name)) + nm + struct.pack("<2I", mi, len(payload)) + payload + b"\0" * pad
    with open(path, "wb") as f:
        f.write(text + struct.pack("<2I", 14, len(body)) + body)


@pytest.mark.gpu
def test_mat_and_tmp_files_round_trip(hl, oracle, tmp_path):
    """The reference's two other raw array formats (tools/halide_image_io.h:1632-1722 .tmp, :1760-2100 .mat): a .mat written in the
    reference's layout comes in, results go out as .mat and .tmp, come back in as inputs, and every path equals the direct call."""
    import struct
    rng = np.random.default_rng(5)
    inp = rng.integers(0, 65536, (3, 96, 160), dtype=np.uint16)
    want = oracle.local_laplacian(inp, 8, np.float32(0.14285714285714285), 1.0)
    _write_mat_like_the_reference(tmp_path / "in.mat", inp)
    args = ["--name=local_laplacian", "levels=8", "alpha=0.14285714285714285", "beta=1", "--output_extents=[160,96,3]"]
    _run(*args, f"input={tmp_path / 'in.mat'}", f"output={tmp_path / 'out.tmp'}")
    raw = open(tmp_path / "out.tmp", "rb").read()
    assert struct.unpack("<5i", raw[:20]) == (160, 96, 3, 1, 4)            # extents padded to four, type code 4 = uint16
    got = np.frombuffer(raw[20:], np.uint16).reshape(3, 96, 160)
    assert np.array_equal(got, want)
    # the .tmp just written as the input, .mat as the output
    with open(tmp_path / "in.tmp", "wb") as f:
        f.write(struct.pack("<5i", 160, 96, 3, 1, 4) + inp.tobytes())
    _run(*args, f"input={tmp_path / 'in.tmp'}", f"output={tmp_path / 'out.mat'}")
    raw = open(tmp_path / "out.mat", "rb").read()
    assert raw[:10] == b"MATLAB 5.0" and raw[126:128] == b"IM"
    assert struct.unpack("<2I", raw[128:136])[0] == 14 and struct.unpack("<4I", raw[136:152]) == (6, 8, 11, 1)
    assert struct.unpack("<2I", raw[152:160]) == (5, 12) and struct.unpack("<4i", raw[160:176]) == (160, 96, 3, 0)
    assert struct.unpack("<2I", raw[176:184]) == (1, 3) and raw[184:187] == b"out"
    assert struct.unpack("<2I", raw[192:200]) == (4, inp.nbytes)
    assert np.array_equal(np.frombuffer(raw[200:200 + inp.nbytes], np.uint16).reshape(3, 96, 160), want)
    # samples of another type convert like every other image file: a float .mat in [0, 1] feeds the u16 input
    _write_mat_like_the_reference(tmp_path / "inf.mat", (inp.astype(np.float64) / 65535.0).astype(np.float32))
    _run(*args, f"input={tmp_path / 'inf.mat'}", f"output={tmp_path / 'outf.npy'}")
    back = np.floor(np.clip((inp.astype(np.float64) / 65535.0).astype(np.float32).astype(np.float64), 0, 1) * 65535.0 + 0.5).astype(np.uint16)
    assert np.array_equal(np.load(tmp_path / "outf.npy").reshape(3, 96, 160), oracle.local_laplacian(back, 8, np.float32(0.14285714285714285), 1.0))


def test_damaged_mat_and_tmp_files_are_rejected(tmp_path):
    import struct
    (tmp_path / "short.tmp").write_bytes(struct.pack("<5i", 16, 16, 3, 1, 4) + b"\0" * 100)
    (tmp_path / "code.tmp").write_bytes(struct.pack("<5i", 4, 4, 1, 1, 12) + b"\0" * 64)
    (tmp_path / "huge.tmp").write_bytes(struct.pack("<5i", 1 << 30, 1 << 30, 4, 4, 2))
    (tmp_path / "junk.mat").write_bytes(b"MATLAB 5.0".ljust(128) + struct.pack("<2I", 15, 64) + b"\0" * 64)   # a compressed element
    # headers that promise more samples than the file holds must be refused before anything of that size is allocated
    (tmp_path / "big.tmp").write_bytes(struct.pack("<5i", 40000, 40000, 4, 1, 1) + b"\0" * 64)          # 51 GB of doubles
    body = struct.pack("<4I", 6, 8, 6, 1) + struct.pack("<2I", 5, 8) + struct.pack("<2i", 50000, 50000) + struct.pack("<2I", 1, 1) + b"m".ljust(8, b"\0")
    body += struct.pack("<2I", 9, 64) + b"\0" * 64
    (tmp_path / "big.mat").write_bytes(b"MATLAB 5.0".ljust(124) + b"\0\x01IM" + struct.pack("<2I", 14, len(body)) + body)
    (tmp_path / "big.npy").write_bytes(b"\x93NUMPY\x01\x00" + struct.pack("<H", 118) + b"{'descr': '<u2', 'fortran_order': False, 'shape': (60000, 60000), }".ljust(117) + b"\n" + b"\0" * 64)
    (tmp_path / "junk.npy").write_bytes(b"NUMPY" + b"\0" * 200)
    (tmp_path / "big.pgm").write_bytes(b"P5\n50000 50000\n65535\n" + b"\0" * 64)
    (tmp_path / "junk.ppm").write_bytes(b"P6\nabc def\n255\n" + b"\0" * 64)
    for name in ("short.tmp", "code.tmp", "huge.tmp", "junk.mat", "big.tmp", "big.mat", "big.npy", "junk.npy", "big.pgm", "junk.ppm"):
        p = _run("--name=stencil_chain", f"input={tmp_path / name}", "output=/dev/null", check=False)
        assert p.returncode > 0, (name, p.returncode, p.stderr[-200:])    # an error exit, not a signal (bad_alloc, SIGSEGV)


def _parse_tiff(raw):
    """A baseline-TIFF reader for the test: the IFD as {tag: values} and the concatenated strips."""
    import struct
    assert raw[:4] == b"II*\0"
    ifd, = struct.unpack("<I", raw[4:8])
    n, = struct.unpack("<H", raw[ifd:ifd + 2])
    fields = {}
    for i in range(n):
        tag, typ, cnt, val = struct.unpack("<HHII", raw[ifd + 2 + 12 * i:ifd + 14 + 12 * i])
        size = {3: 2, 4: 4, 5: 8}[typ]
        if typ == 5:
            fields[tag] = struct.unpack("<2I", raw[val:val + 8])
        elif size * cnt <= 4:
            fields[tag] = struct.unpack("<" + "HI"[typ - 3] * cnt, raw[ifd + 10 + 12 * i:ifd + 10 + 12 * i + size * cnt])
        else:
            fields[tag] = struct.unpack("<" + "HI"[typ - 3] * cnt, raw[val:val + size * cnt])
    assert struct.unpack("<I", raw[ifd + 2 + 12 * n:ifd + 6 + 12 * n]) == (0,)
    data = b"".join(raw[o:o + c] for o, c in zip(fields[273], fields[279]))
    return fields, data


@pytest.mark.gpu
def test_tiff_out_and_back_in(hl, oracle, tmp_path):
    """TIFF the way the reference writes it (tools/halide_image_io.h:2230-2384): uncompressed, little endian, one strip per
    channel in planar configuration, SampleFormat from the element type.  The reference declines to read TIFF (:2109-2113);
    this runner reads the family it writes, plus chunky and big-endian files, so outputs can be fed back."""
    import struct
    rng = np.random.default_rng(6)
    inp = rng.integers(0, 65536, (3, 96, 160), dtype=np.uint16)
    want = oracle.local_laplacian(inp, 8, np.float32(0.14285714285714285), 1.0)
    np.save(tmp_path / "in.npy", inp.reshape(160, 96, 3))              # the reference's .npy convention: extents x-first
    args = ["--name=local_laplacian", "levels=8", "alpha=0.14285714285714285", "beta=1", "--output_extents=[160,96,3]"]
    _run(*args, f"input={tmp_path / 'in.npy'}", f"output={tmp_path / 'out.tiff'}")
    fields, data = _parse_tiff(open(tmp_path / "out.tiff", "rb").read())
    assert fields[256] == (160,) and fields[257] == (96,) and fields[258] == (16,) and fields[259] == (1,)
    assert fields[262] == (2,) and fields[277] == (3,) and fields[278] == (96,) and fields[284] == (2,)
    assert fields[339] == (1,) and fields[32997] == (1,) and fields[282] == (1, 1) and fields[283] == (1, 1)
    assert fields[279] == (160 * 96 * 2,) * 3 and len(fields[273]) == 3
    assert sorted(fields) == [256, 257, 258, 259, 262, 273, 277, 278, 279, 282, 283, 284, 296, 339, 32997]
    assert np.array_equal(np.frombuffer(data, np.uint16).reshape(3, 96, 160), want)
    # that file as the input
    _run(*args, f"input={tmp_path / 'out.tiff'}", f"output={tmp_path / 'twice.npy'}")
    assert np.array_equal(np.load(tmp_path / "twice.npy").reshape(3, 96, 160), oracle.local_laplacian(want, 8, np.float32(0.14285714285714285), 1.0))
    # a chunky big-endian file in two strips, as other writers produce
    chunky = np.ascontiguousarray(inp.transpose(1, 2, 0)).astype(">u2").tobytes()
    half = 48 * 160 * 3 * 2
    tags = [(256, 4, 1, 160), (257, 4, 1, 96), (258, 3, 3, None), (259, 3, 1, 1 << 16), (262, 3, 1, 2 << 16), (273, 4, 2, None),
            (277, 3, 1, 3 << 16), (278, 4, 1, 48), (279, 4, 2, None), (284, 3, 1, 1 << 16), (339, 3, 1, 1 << 16)]
    extra_at = 8 + 2 + 12 * len(tags) + 4
    extra = struct.pack(">3H", 16, 16, 16) + b"\0\0"
    data_at = extra_at + len(extra) + 16
    extra += struct.pack(">2I", data_at, data_at + half) + struct.pack(">2I", half, half)
    offs = {258: extra_at, 273: extra_at + 8, 279: extra_at + 16}
    ifd = struct.pack(">H", len(tags)) + b"".join(struct.pack(">HHII", t, ty, c, offs.get(t, v)) for t, ty, c, v in tags) + struct.pack(">I", 0)
    (tmp_path / "be.tif").write_bytes(b"MM\0*" + struct.pack(">I", 8) + ifd + extra + chunky)
    _run(*args, f"input={tmp_path / 'be.tif'}", f"output={tmp_path / 'be.npy'}")
    assert np.array_equal(np.load(tmp_path / "be.npy").reshape(3, 96, 160), want)
    # float output: SampleFormat 3, one sample per pixel -> contiguous configuration with the byte count inline
    src = rng.random((3, 64, 96), dtype=np.float32)
    np.save(tmp_path / "f.npy", src.reshape(96, 64, 3))
    _run("--name=iir_blur", f"input={tmp_path / 'f.npy'}", "alpha=0.5", f"output={tmp_path / 'f.tiff'}")
    fields, data = _parse_tiff(open(tmp_path / "f.tiff", "rb").read())
    assert fields[258] == (32,) and fields[339] == (3,) and fields[277] == (3,)
    assert np.array_equal(np.frombuffer(data, np.float32).reshape(3, 64, 96), oracle.iir_blur(src, np.float32(0.5)))
    _run("--name=iir_blur", f"input={tmp_path / 'f.tiff'}", "alpha=0.5", f"output={tmp_path / 'f2.npy'}")
    assert np.array_equal(np.load(tmp_path / "f2.npy").reshape(3, 64, 96), oracle.iir_blur(oracle.iir_blur(src, np.float32(0.5)), np.float32(0.5)))


def test_damaged_tiff_files_are_rejected(tmp_path):
    import struct
    def tiff(tags, tail=b""):
        return b"II*\0" + struct.pack("<IH", 8, len(tags)) + b"".join(struct.pack("<HHII", *t) for t in tags) + struct.pack("<I", 0) + tail
    base = [(256, 4, 1, 4), (257, 4, 1, 4), (258, 3, 1, 8), (259, 3, 1, 1), (273, 4, 1, 200), (277, 3, 1, 1), (279, 4, 1, 16)]
    cases = {
        "magic.tiff": b"II+\0" + b"\0" * 64,
        "lzw.tiff": tiff([t if t[0] != 259 else (259, 3, 1, 5) for t in base], b"\0" * 300),
        "short.tiff": tiff(base, b"\0" * 20),                                  # the strip lies beyond the end of the file
        "huge.tiff": tiff([t if t[0] != 256 else (256, 4, 1, 1 << 30) for t in base], b"\0" * 300),
        "bits.tiff": tiff([t if t[0] != 258 else (258, 3, 1, 12) for t in base], b"\0" * 300),
        "few.tiff": tiff([t if t[0] != 279 else (279, 4, 1, 8) for t in base], b"\0" * 300),   # strips hold half the image
        "ifd.tiff": b"II*\0" + struct.pack("<I", 4000),
    }
    for name, blob in cases.items():
        (tmp_path / name).write_bytes(blob)
        p = _run("--name=stencil_chain", f"input={tmp_path / name}", "output=/dev/null", check=False)
        assert p.returncode != 0 and "Segmentation" not in p.stderr, name
        assert p.returncode > 0, (name, p.returncode)                         # an error exit, not a signal


JPEG_DIR = os.path.join(ROOT, "tests", "golden", "jpeg")
JPEG_CASES = ["rgb444_q95", "rgb422_q75", "rgb420_q75", "rgb420_q30_opt", "rgb420_q90_rst", "rgb420_narrow", "gray_q85", "gray_q100_rst"]


def _jpeg_tool(tmp_path):
    exe = tmp_path / "jpeg_codec_test"
    if not exe.exists():
        subprocess.run(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "halide_amd", "tools"),
                        os.path.join(ROOT, "tests", "cpp", "jpeg_codec_test.cpp"), "-o", str(exe)], check=True)
    return str(exe)


def test_jpeg_decoder_returns_libjpegs_samples(tmp_path):
    """The runner's JPEG decoder against libjpeg-turbo (through Pillow, scripts/make_jpeg_golden.py): 4:4:4 / 4:2:2 / 4:2:0, gray,
    optimised Huffman tables, restart intervals, a component too narrow for the triangle filter — every sample identical
    (the reference reads JPEG through libjpeg's defaults, tools/halide_image_io.h:1506-1548)."""
    tool = _jpeg_tool(tmp_path)
    want = np.load(os.path.join(JPEG_DIR, "expected.npz"))
    assert sorted(n for n in want.files if not n.startswith("src_")) == sorted(JPEG_CASES)
    for name in JPEG_CASES:
        out = str(tmp_path / (name + ".bin"))
        p = subprocess.run([tool, "decode", os.path.join(JPEG_DIR, name + ".jpg"), out], capture_output=True, text=True)
        assert p.returncode == 0, (name, p.stderr)
        w, h, c = map(int, p.stdout.split())
        got = np.fromfile(out, np.uint8).reshape(h, w, c)
        assert got.shape == want[name].shape and np.array_equal(got, want[name]), name


def test_jpeg_decoder_refuses_what_it_does_not_decode(tmp_path):
    tool = _jpeg_tool(tmp_path)
    good = open(os.path.join(JPEG_DIR, "rgb420_q75.jpg"), "rb").read()
    sof = good.index(b"\xff\xc0")
    cases = {
        "progressive.jpg": good[:sof] + b"\xff\xc2" + good[sof + 2:],          # SOF2
        "twelve_bit.jpg": good[:sof + 4] + b"\x0c" + good[sof + 5:],           # sample precision 12
        "cut_header.jpg": good[:sof + 6],
        "cut_scan.jpg": good[:len(good) * 2 // 3],                              # ends inside the entropy-coded data
        "not_jpeg.jpg": b"\x89PNG\r\n\x1a\n" + good[8:],
        "no_tables.jpg": good[:2] + good[good.index(b"\xff\xc0"):],             # frame and scan without DQT / DHT
    }
    for name, blob in cases.items():
        (tmp_path / name).write_bytes(blob)
        p = subprocess.run([tool, "decode", str(tmp_path / name), str(tmp_path / "x.bin")], capture_output=True, text=True)
        assert p.returncode == 1 and p.stderr.strip(), (name, p.returncode, p.stderr)
    # random damage: an answer or a refusal, never a crash
    rng = np.random.default_rng(11)
    for trial in range(60):
        d = bytearray(good)
        for _ in range(int(rng.integers(1, 6))):
            d[int(rng.integers(2, len(d)))] = int(rng.integers(0, 256))
        (tmp_path / "fz.jpg").write_bytes(bytes(d))
        p = subprocess.run([tool, "decode", str(tmp_path / "fz.jpg"), str(tmp_path / "x.bin")], capture_output=True, text=True)
        assert p.returncode in (0, 1), (trial, p.returncode, p.stderr[-200:])


def test_jpeg_encoder_writes_libjpegs_file(tmp_path):
    """The runner's JPEG encoder against libjpeg-turbo (through Pillow): the same FILE, byte for byte — JFIF header, scaled
    Annex K quantization tables, 4:2:0 chroma with libjpeg's edge padding and dummy blocks, accurate integer DCT, Annex K Huffman
    tables (the reference writes JPEG through libjpeg's defaults at quality 99, tools/halide_image_io.h:1558-1610)."""
    tool = _jpeg_tool(tmp_path)
    gold = np.load(os.path.join(JPEG_DIR, "expected.npz"))
    for name, q in (("enc_rgb_q99", 99), ("enc_rgb_q75_odd", 75), ("enc_gray_q99", 99), ("enc_rgb_q30_tiny", 30)):
        src = gold["src_" + name]
        h, w, c = src.shape
        (tmp_path / "src.bin").write_bytes(src.tobytes())
        out = tmp_path / (name + ".jpg")
        p = subprocess.run([tool, "encode", str(tmp_path / "src.bin"), str(w), str(h), str(c), str(out), str(q)], capture_output=True, text=True)
        assert p.returncode == 0, (name, p.stderr)
        assert out.read_bytes() == open(os.path.join(JPEG_DIR, name + ".jpg"), "rb").read(), name
        # and what was written reads back through the decoder as libjpeg reads it (checked against Pillow when the fixtures were made)
        p = subprocess.run([tool, "decode", str(out), str(tmp_path / "back.bin")], capture_output=True, text=True)
        assert p.returncode == 0 and p.stdout.split() == [str(w), str(h), str(c)], (name, p.stderr)


@pytest.mark.gpu
def test_jpeg_input_through_the_runner(oracle, tmp_path):
    """A JPEG file as a pipeline input: decoded, converted like every other 8-bit image file (u8 -> u16: x 257) and run."""
    want_rgb = np.load(os.path.join(JPEG_DIR, "expected.npz"))["rgb420_q75"]                  # (37, 61, 3) u8
    inp = np.ascontiguousarray(want_rgb.transpose(2, 0, 1)).astype(np.uint16) * 257
    _run("--name=local_laplacian", f"input={os.path.join(JPEG_DIR, 'rgb420_q75.jpg')}", "levels=4", "alpha=0.3333333333333333", "beta=1",
         "--output_extents=[61,37,3]", f"output={tmp_path / 'out.npy'}")
    got = np.load(tmp_path / "out.npy").reshape(3, 37, 61)
    want = oracle.local_laplacian(inp, 4, np.float32(0.3333333333333333), 1.0)
    assert np.array_equal(got, want)
    # ... and a JPEG file as the output: the u16 result narrowed the way the reference narrows it, then libjpeg's file
    _run("--name=local_laplacian", f"input={os.path.join(JPEG_DIR, 'rgb420_q75.jpg')}", "levels=4", "alpha=0.3333333333333333", "beta=1",
         "--output_extents=[61,37,3]", f"output={tmp_path / 'out.jpg'}")
    narrowed = (((want.astype(np.uint32) + 0x80) * 255 + 255) >> 16).astype(np.uint8).transpose(1, 2, 0)
    tool = _jpeg_tool(tmp_path)
    (tmp_path / "n.bin").write_bytes(np.ascontiguousarray(narrowed).tobytes())
    subprocess.run([tool, "encode", str(tmp_path / "n.bin"), "61", "37", "3", str(tmp_path / "ref.jpg"), "99"], check=True)
    assert (tmp_path / "out.jpg").read_bytes() == (tmp_path / "ref.jpg").read_bytes()
