"""Which array's PLACEMENT moves the walk's time?  One C3-shaped run; k alternative copies of the output arrays (words, umi_start ...)
and k of the input pool, every combination timed in one process (dev tool).  usage: bp_place_probe.py [n_loci] [k]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from smcounter_amd import _lib, abi, synth, engine
from smcounter_amd.engine import DevBuf
import bench_fa

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 3
eng = engine.Engine(0)
L = eng.L
cfg = synth.CONFIGS["C3"]
run = bench_fa.AlignmentRun(eng, cfg, synth.params_for(cfg), n, 8, slots=k)
pools = [run.d_in[2]] + [DevBuf(eng, run.A["bq"].nbytes + 256).upload(run.A["bq"].view(np.uint8).reshape(-1)) for _ in range(k - 1)]
print("words at", ["0x%x" % S["words"].data_ptr() for S in run.slots])
print("pools at", ["0x%x" % p.data_ptr() for p in pools])


def timed(slot, reps=5):
    for _ in range(2):
        run.step(slot=slot)
    L.smc_device_sync(eng.ctx)
    L.smc_build_set_timing(eng.ctx, reps)
    for _ in range(reps):
        run.step(slot=slot)
    L.smc_device_sync(eng.ctx)
    k_ms, k_n = ctypes.c_float(), ctypes.c_int32()
    L.smc_build_kernel_ms(eng.ctx, ctypes.byref(k_ms), ctypes.byref(k_n))
    L.smc_build_set_timing(eng.ctx, 0)
    return k_ms.value


for pi, p in enumerate(pools):
    run.bi.bq = p.data_ptr()
    print("pool %d: " % pi + "  ".join("outputs %d: %.3f ms" % (s, timed(s)) for s in range(k)))
