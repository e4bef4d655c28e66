"""Does k_bp_emit's time depend on WHERE its arrays lie?  (Same code, same box, same data: 4.2 or 4.9 ms per process.)  Several
independently allocated copies of one C3-shaped run inside ONE process, the walk's own time for each, twice round (dev tool).
usage: bp_addr_probe.py [n_loci] [copies]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from smcounter_amd import _lib, abi, synth, engine
import bench_fa as fa_leg

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
copies = int(sys.argv[2]) if len(sys.argv) > 2 else 3
eng = engine.Engine(0)
L = eng.L
cfg = synth.CONFIGS["C3"]
runs = [fa_leg.AlignmentRun(eng, cfg, synth.params_for(cfg), n, 8) for _ in range(copies)]
for i, r in enumerate(runs):
    print("copy %d: words 0x%x umi_start 0x%x bq 0x%x aln 0x%x" % (i, r.words.data_ptr(), r.uaux[0].data_ptr(), r.d_in[2].data_ptr(), r.d_in[0].data_ptr()))


def timed(run, reps=6):
    for _ in range(2):
        run.step()
    L.smc_device_sync(eng.ctx)
    L.smc_build_set_timing(eng.ctx, reps)
    for _ in range(reps):
        run.step()
    L.smc_device_sync(eng.ctx)
    k_ms, k_n = ctypes.c_float(), ctypes.c_int32()
    L.smc_build_kernel_ms(eng.ctx, ctypes.byref(k_ms), ctypes.byref(k_n))
    L.smc_build_set_timing(eng.ctx, 0)
    return k_ms.value


for rnd in range(2):
    print("round %d: k_bp_emit " % rnd + "  ".join("copy %d %.3f ms" % (i, timed(r)) for i, r in enumerate(runs)))
