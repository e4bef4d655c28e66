"""Does a spacer allocation before it change how good an allocation of the read words is for the walk?  (dev tool)"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from smcounter_amd import _lib, abi, synth, engine
from smcounter_amd.engine import DevBuf
import bench_fa

eng = engine.Engine(0)
L = eng.L
cfg = synth.CONFIGS["C3"]
run = bench_fa.AlignmentRun(eng, cfg, synth.params_for(cfg), 200000, 8, slots=1)


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


print("first allocation: %.3f ms" % timed())
keep = []
for i, mb in enumerate([0, 37, 301, 611, 1024, 1777, 2500, 3333, 150, 4097]):
    sp = DevBuf(eng, (mb << 20) + 4096) if mb else None
    w = DevBuf(eng, 4 * (run.ns + 64))
    keep += [sp, w]
    run.slots[0]["words"] = w
    print("spacer %5d MB: words at 0x%x: %.3f ms" % (mb, w.data_ptr(), timed()))
