"""The reference's own consumers of the boundary, linked against libhlmi.so (recipe: oracle/ref.mk -> oracle/_ref/):

* `<name>.rungen` = the reference's tools/RunGenMain.cpp + tools/RunGen.h, compiled UNMODIFIED, plus the three-line
  registration unit (tests/cpp/rungen_registration.cpp = what `-e registration` emits, src/Module.cpp:171-201).  RunGen
  reads `<name>_metadata()` (static-data layout of src/CodeGen_C.cpp:760-912: buffer_estimates / scalar_estimate
  pointers), sizes its buffers with the bounds-query protocol (tools/RunGen.h:1212-1250, 1384-1430), calls
  `<name>_argv`, syncs and copies back through the device interface (RunGen.h:1129-1150) and replaces the allocator
  (RunGenMain.cpp:220-259: halide_default_malloc / halide_set_custom_malloc).
* `entry_protocol_ref` (tests/cpp): the cases of test/generator/error_codes_aottest.cpp:27-120 with the reference's
  own HalideRuntime.h enumerators.
* `device_interface_test` (tests/cpp): device_crop / device_slice / release_crop / buffer_copy /
  device_and_host_malloc / wrap_native through the reference's Halide::Runtime::Buffer, plus cross-stream ordering.

A missing binary is a failure: they are built by __graft_entry__.build() and travel to the GPU box.
"""
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
    assert os.path.exists(p), f"oracle/_ref/{name} is missing (make -C oracle ref, where /root/reference is present)"
    return p


def _run(name, *args, timeout=600):
    return subprocess.run([_exe(name), *args], capture_output=True, text=True, timeout=timeout)


def test_error_codes_replay_with_the_reference_header():
    """No GPU needed: every case fails in the prologue."""
    r = _run("entry_protocol_ref", "errors")
    assert r.returncode == 0 and "Success!" in r.stdout, r.stdout + r.stderr


def test_reference_rungen_describe_reads_our_metadata():
    r = _run("local_laplacian.rungen", "--describe")
    assert r.returncode == 0, r.stdout + r.stderr
    assert 'Filter name: "local_laplacian"' in r.stdout
    assert 'Input "input" is of type Buffer<uint16> with 3 dimensions' in r.stdout
    assert 'Input "levels" is of type int32' in r.stdout and 'Output "output" is of type Buffer<uint16> with 3 dimensions' in r.stdout
    r = _run("camera_pipe.rungen", "--describe")
    assert r.returncode == 0 and 'Input "matrix_3200" is of type Buffer<float32> with 2 dimensions' in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_error_codes_replay_with_a_device():
    r = _run("entry_protocol_ref", "all")
    assert r.returncode == 0 and "Success!" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_device_interface_through_halide_runtime_buffer():
    r = _run("device_interface_test")
    assert r.returncode == 0 and "Success!" in r.stdout, r.stdout + r.stderr


RUNGEN_KEYS = {"BEST_TIME_MSEC_PER_ITER", "SAMPLES", "ITERATIONS", "TIMING_ACCURACY", "THROUGHPUT_MPIX_PER_SEC", "HALIDE_TARGET"}


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["local_laplacian", "bilateral_grid", "halide_blur", "nl_means", "stencil_chain", "conv_layer",
                                  "camera_pipe", "depthwise_separable_conv", "unsharp", "max_filter", "hist", "harris",
                                  "interpolate", "iir_blur", "lens_blur", "bgu"])
def test_reference_rungen_estimate_all_benchmark(name):
    """tools/RunGen.h:1285-1298: the parsable benchmark lines; inputs and extents from the metadata's estimates."""
    r = _run(f"{name}.rungen", "--estimate_all", "--benchmarks=all", "--parsable_output", "--benchmark_min_time=0.05")
    assert r.returncode == 0, r.stdout + r.stderr
    keys = {l.split()[1] for l in r.stdout.splitlines() if l.startswith(name + " ") and len(l.split()) >= 3}
    assert RUNGEN_KEYS <= keys, r.stdout + r.stderr
    target = [l.split()[2] for l in r.stdout.splitlines() if l.startswith(name + " ") and l.split()[1] == "HALIDE_TARGET"]
    assert target and "hip" in target[0]


@pytest.mark.gpu
def test_reference_rungen_output_equals_the_oracle(oracle, tmp_path):
    """Image in, image out through the reference's RunGen == oracle.  The reference's .npy files list the HALIDE
    extents (dimension 0 first) under 'fortran_order': False while the payload has dimension 0 fastest
    (tools/halide_image_io.h:1433-1445, :1376-1387): the bytes of a planar (3, H, W) C-order array under the shape
    label (W, H, 3)."""
    rng = np.random.default_rng(11)
    inp = rng.integers(0, 65536, (3, 96, 160), dtype=np.uint16)
    np.save(tmp_path / "in.npy", inp.reshape(160, 96, 3))
    r = _run("local_laplacian.rungen", f"input={tmp_path / 'in.npy'}", "levels=8", "alpha=0.14285714285714285", "beta=1",
             f"output={tmp_path / 'out.npy'}", "--output_extents=[160,96,3]")
    assert r.returncode == 0, r.stdout + r.stderr
    got = np.load(tmp_path / "out.npy")
    assert got.shape == (160, 96, 3) and got.dtype == np.uint16
    want = oracle.local_laplacian(inp, 8, np.float32(0.14285714285714285), 1.0)
    assert np.array_equal(got.reshape(3, 96, 160), want)


@pytest.mark.gpu
def test_reference_rungen_tracks_allocations_through_our_malloc_hooks():
    """RunGenMain.cpp:220-259 installs a tracking allocator around halide_default_malloc; --track_memory reports it."""
    r = _run("halide_blur.rungen", "--estimate_all", "--track_memory", "--benchmarks=all", "--benchmark_min_time=0.02")
    assert r.returncode == 0, r.stdout + r.stderr
