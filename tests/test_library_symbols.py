"""The C-ABI library must load without a GPU and export every symbol include/*.h declares."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    names = set()
    for hdr in ("hlmi_runtime.h", "hlmi_pipelines.h"):
        src = open(os.path.join(ROOT, "include", hdr)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        src = re.sub(r"#define HLMI_DECLARE_AUX.*?\n\n", "\n", src, flags=re.S)
        for m in re.finditer(r"HLMI_DECLARE_AUX\((\w+)\)", src):
            names.update({m.group(1) + "_argv", m.group(1) + "_metadata"})
        for m in re.finditer(r"^\s*(?:const\s+)?[\w\s\*]+?\b(\w+)\s*\([^;{]*\)\s*;", src, flags=re.M):
            if not m.group(0).lstrip().startswith("typedef"):
                names.add(m.group(1))
    return names


def test_every_declared_symbol_is_exported(hl):
    lib = ctypes.CDLL(hl.LIB_PATH)
    names = _declared()
    assert {"local_laplacian", "camera_pipe_argv", "halide_hip_device_interface", "halide_copy_to_host",
            "conv_layer_auto_schedule"} <= names
    missing = [n for n in sorted(names) if not hasattr(lib, n)]
    assert not missing, missing


def test_metadata_of_every_pipeline(hl):
    expect = {"local_laplacian": 5, "bilateral_grid": 3, "halide_blur": 2, "nl_means": 5, "stencil_chain": 2,
              "conv_layer": 4, "conv_layer_bf16": 4, "depthwise_separable_conv": 5, "unsharp": 2, "max_filter": 2, "hist": 2, "harris": 2, "interpolate": 2, "iir_blur": 3, "camera_pipe": 10, "lens_blur": 7, "bgu": 6}
    for name, n in expect.items():
        md = hl.metadata(name)
        assert md.version == 1 and md.num_arguments == n and md.name.decode() == name
        assert b"hip" in md.target
        kinds = [md.arguments[i].kind for i in range(n)]
        assert kinds[-1] == 2 and kinds[0] == (0 if name == "bgu" else 1)  # inputs in declaration order (bgu: its two scalars), the output buffer last


def test_no_gpu_means_loud_failure_not_fallback(hl):
    """Without a usable gfx950 device the product refuses to run (-29); it never computes on the CPU."""
    import numpy as np
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("a GPU is present")
    a, o = hl.Buffer(np.zeros((4, 4), np.uint16)), hl.Buffer(np.zeros((2, 2), np.uint16))
    try:
        hl.halide_blur(a, o)
        raise AssertionError("expected HalideError")
    except hl.HalideError as e:
        assert e.code == -29


def test_one_hip_runtime_per_process_whatever_the_import_order():
    """The torch wheel bundles its own libamdhip64.so; libhlmi.so loaded first used to bind to the system copy, torch
    then brought its own, and a torch stream handed to the library belonged to the other runtime (a crash inside HIP that
    depended on test order).  halide_amd loads torch's copy first when torch is installed: one runtime either way."""
    import subprocess
    import sys
    import importlib.util
    if importlib.util.find_spec("torch") is None:
        import pytest
        pytest.skip("torch is not installed")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r)\n%s\n"
            "libs = sorted({l.split()[-1] for l in open('/proc/self/maps') if 'amdhip64' in l and 'r-xp' in l})\n"
            "print('RUNTIMES', len(libs), libs)\n"
            "import halide_amd as hl\nassert hl.hip_runtime().hipGetDeviceCount\n")
    for order in ("import halide_amd, torch", "import torch, halide_amd"):
        r = subprocess.run([sys.executable, "-c", code % (root, order)], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout + r.stderr
        line = [l for l in r.stdout.splitlines() if l.startswith("RUNTIMES")][0]
        assert line.split()[1] == "1", f"{order}: {line}"
