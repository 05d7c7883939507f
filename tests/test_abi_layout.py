"""The re-declared ABI (include/hlmi_abi.h) must be layout-identical to the reference's
src/runtime/HalideRuntime.h; checked with static_asserts by compiling against the reference header
where that tree exists (dev container), and against the numbers recorded from it otherwise."""
import os
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/src/runtime"

# sizes/offsets measured from the reference header
EXPECT = dict(sz_buf=56, sz_dim=16, sz_type=4, off_device=0, off_iface=8, off_host=16, off_flags=24, off_type=32,
              off_dims=36, off_dim=40, off_pad=48, sz_iface=128, sz_arg=64, sz_md=32)

PROBE = r"""
#include <cstddef>
#include <cstdio>
#include INCLUDE_HEADER
int main() {
  printf("sz_buf=%zu sz_dim=%zu sz_type=%zu off_device=%zu off_iface=%zu off_host=%zu off_flags=%zu off_type=%zu "
         "off_dims=%zu off_dim=%zu off_pad=%zu sz_iface=%zu sz_arg=%zu sz_md=%zu\n",
         sizeof(halide_buffer_t), sizeof(halide_dimension_t), sizeof(halide_type_t),
         offsetof(halide_buffer_t, device), offsetof(halide_buffer_t, device_interface),
         offsetof(halide_buffer_t, host), offsetof(halide_buffer_t, flags), offsetof(halide_buffer_t, type),
         offsetof(halide_buffer_t, dimensions), offsetof(halide_buffer_t, dim), offsetof(halide_buffer_t, padding),
         sizeof(halide_device_interface_t), sizeof(halide_filter_argument_t), sizeof(halide_filter_metadata_t));
  printf("codes %d %d %d %d %d %d %d\n", (int)halide_error_code_bad_type, (int)halide_error_code_access_out_of_bounds,
         (int)halide_error_code_constraint_violated, (int)halide_error_code_buffer_argument_is_null,
         (int)halide_error_code_host_is_null, (int)halide_error_code_bad_dimensions,
         (int)halide_error_code_device_dirty_with_no_device_support);
  return 0;
}
"""


def _probe(header, incdir):
    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, "p.cpp")
        with open(src, "w") as f:
            f.write(PROBE.replace("INCLUDE_HEADER", f'"{header}"'))
        exe = os.path.join(td, "p")
        subprocess.run(["g++", "-std=c++17", "-I", incdir, src, "-o", exe], check=True)
        out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout.splitlines()
    vals = dict(kv.split("=") for kv in out[0].split())
    return {k: int(v) for k, v in vals.items()}, out[1]


def test_own_header_matches_recorded_layout():
    vals, codes = _probe("hlmi_abi.h", os.path.join(ROOT, "include"))
    assert vals == EXPECT
    assert codes == "codes -3 -4 -8 -12 -34 -43 -44"


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "HalideRuntime.h")), reason="reference tree absent")
def test_reference_header_has_same_layout():
    ours, codes_ours = _probe("hlmi_abi.h", os.path.join(ROOT, "include"))
    ref, codes_ref = _probe("HalideRuntime.h", REF)
    assert ours == ref
    assert codes_ours == codes_ref


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "HalideRuntime.h")), reason="reference tree absent")
def test_headers_compile_on_top_of_reference_runtime_header():
    """A caller that includes the reference's runtime header first must be able to include ours
    (pipelines + runtime API) without redefinition errors."""
    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, "c.cpp")
        with open(src, "w") as f:
            f.write('#include "HalideRuntime.h"\n#include "hlmi_pipelines.h"\n'
                    'int main(){ halide_buffer_t b = {0}; (void)b; return 0; }\n')
        subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-I", REF, "-I", os.path.join(ROOT, "include"), src],
                       check=True)
