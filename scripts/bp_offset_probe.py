"""The walk's time against the OFFSET of the read-words array inside one allocation (dev tool).  usage: bp_offset_probe.py [n_loci]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from smcounter_amd import _lib, abi, synth, engine
from smcounter_amd.engine import DevBuf
import bench_fa

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
eng = engine.Engine(0)
L = eng.L
cfg = synth.CONFIGS["C3"]
run = bench_fa.AlignmentRun(eng, cfg, synth.params_for(cfg), n, 8, slots=1)
SLACK = 80 << 20
big = DevBuf(eng, 4 * (run.ns + 64) + SLACK)


class View(object):
    def __init__(self, base, off): self.p = base + off
    def data_ptr(self): return self.p
    def free(self): pass


def timed(reps=4):
    run.step(slot=0)
    L.smc_device_sync(eng.ctx)
    L.smc_build_set_timing(eng.ctx, reps)
    for _ in range(reps):
        run.step(slot=0)
    L.smc_device_sync(eng.ctx)
    k_ms, k_n = ctypes.c_float(), ctypes.c_int32()
    L.smc_build_kernel_ms(eng.ctx, ctypes.byref(k_ms), ctypes.byref(k_n))
    L.smc_build_set_timing(eng.ctx, 0)
    return k_ms.value


print("base 0x%x" % big.data_ptr())
for off in [0, 256, 4096, 65536, 1 << 18, 1 << 20, 2 << 20, 3 << 20, 4 << 20, 6 << 20, 8 << 20, 12 << 20, 16 << 20, 24 << 20, 32 << 20, 48 << 20, 64 << 20, 0]:
    run.slots[0]["words"] = View(big.data_ptr(), off)
    print("offset %9d: %.3f ms" % (off, timed()))
