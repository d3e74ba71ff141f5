"""Static regression guard on the generated gfx950 code (no GPU needed: hipcc cross-compiles).

The f32 matrix pipe loses ~5 cycles per vector-ALU instruction issued on its SIMD ("VALU diet"), and twice
during development hipcc quietly moved address arithmetic back into the MFMA loop.  This test compiles the device code
to assembly and checks, for every instantiation of the occupancy conv kernel, that
  * the 3x3 conv's MFMA blocks contain no vector-ALU instruction at all (MODE 1/2/3 may keep a handful),
  * nothing spills to scratch,
and that no 16-byte buffer store is directly followed by an instruction that overwrites its data registers: hipcc does not
insert the wait state gfx950 needs there when the store uses an SGPR soffset, and the store then writes garbage
(profiles/r02_epilogue_fence_audit.md -- the root cause of round 1's non-repeatable wrong lanes).
"""
import os
import re
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_conv_main_loops_have_no_vector_alu_and_no_scratch(tmp_path):
    asm = tmp_path / "og_api.s"
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                    os.path.join(ROOT, "openglottal_amd", "csrc", "og_api.hip"), "-o", str(asm)],
                   check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_audit.py"), str(asm)],
                         check=True, capture_output=True, text=True).stdout
    hz = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_store_hazard.py"), str(asm)], capture_output=True, text=True)
    assert hz.returncode == 0 and "overwritten by the next instruction: 0" in hz.stdout, hz.stdout
    rows = [l for l in out.splitlines() if l.startswith("k_conv_mfma_o<")]
    assert len(rows) >= 10, out
    for l in rows:
        mode = int(re.search(r"MODE=(\d)", l).group(1))
        in_loop = int(re.search(r"in MFMA blocks\s+(\d+)", l).group(1))
        scratch = int(re.search(r"private_segment_fixed_size (\d+)", l).group(1))
        assert scratch == 0, l
        assert in_loop == 0 if mode == 0 else in_loop <= 16, l
