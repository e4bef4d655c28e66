"""Registers / scratch / LDS of kernels in a built library.  usage: kres.py LIB.so [name-substring]"""
import re, subprocess, sys, tempfile, os
LLVM = "/opt/rocm/lib/llvm/bin"
so, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
with tempfile.TemporaryDirectory() as d:
    fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "dev.co")
    subprocess.check_call([LLVM + "/llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, so, os.path.join(d, "copy.so")])
    subprocess.check_call([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + fat,
                           "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
    txt = subprocess.check_output([LLVM + "/llvm-readelf", "--notes", co], text=True)
for blk in re.split(r"\n\s*- \.agpr_count", txt):
    m = re.search(r"\.name:\s+(\S+)", blk)
    if m and pat in m.group(1):
        f = dict(re.findall(r"\.(private_segment_fixed_size|vgpr_count|sgpr_count|group_segment_fixed_size|sgpr_spill_count|vgpr_spill_count):\s+(\d+)", blk))
        print(m.group(1), f)
