"""Resource checks on the built gfx950 code object (no GPU needed).

k_bp_emit2 issues its pipeline loads as inline assembly and waits for them itself (csrc/k_bp_emit2.inc): the compiler does not
know that their destination registers are in flight.  A build in which the register allocator spills - stores such a register
to scratch right after the load that is still filling it - would compute on garbage addresses.  So: no scratch in that kernel."""
import os
import re
import shutil
import subprocess

import pytest

from conftest import ROOT

LLVM = "/opt/rocm/lib/llvm/bin"


def _kernel_notes(tmp_path):
    so = os.path.join(ROOT, "smcounter_amd", "libsmcounter_hip.so")
    tools = [os.path.join(LLVM, t) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-readelf")]
    if not os.path.exists(so) or not all(os.path.exists(t) for t in tools):
        pytest.skip("library or LLVM tools missing")
    fat, co = str(tmp_path / "fat.bin"), str(tmp_path / "dev.co")
    subprocess.check_call([tools[0], "--dump-section", ".hip_fatbin=" + fat, so, str(tmp_path / "copy.so")])
    subprocess.check_call([tools[1], "--unbundle", "--type=o", "--input=" + fat, "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                           "--output=" + co])
    txt = subprocess.check_output([tools[2], "--notes", co], text=True)
    out = {}
    for blk in re.split(r"\n\s*- \.agpr_count", txt):
        m = re.search(r"\.name:\s+(\S+)", blk)
        if m:
            f = {k: int(v) for k, v in re.findall(r"\.(private_segment_fixed_size|vgpr_count|sgpr_count|group_segment_fixed_size):\s+(\d+)", blk)}
            out[m.group(1)] = f
    return out


def test_the_plane_walk_uses_no_scratch(tmp_path):
    notes = _kernel_notes(tmp_path)
    walks = {k: v for k, v in notes.items() if "k_bp_emit2" in k}
    assert len(walks) == 3, sorted(notes)          # raw-field planes too, 32-bit words, 16-bit words
    for name, f in walks.items():
        assert f["private_segment_fixed_size"] == 0, (name, f)
        assert f["vgpr_count"] <= 96, (name, f)            # five wavefronts per SIMD
