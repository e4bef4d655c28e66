"""Does the library's write-pattern probe (smc_mem_write_probe) rank allocations of the read words as the walk itself does?  (dev tool)"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["SMC_VMM_CHUNK_MB"] = "0"
import numpy as np
from smcounter_amd import synth, engine
from smcounter_amd.engine import DevBuf
import bench_fa
eng = engine.Engine(0)
L = eng.L
L.smc_mem_write_probe.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.POINTER(ctypes.c_float)]
cfg = synth.CONFIGS[os.environ.get("SWEEP_CFG", "C3")]
run = bench_fa.AlignmentRun(eng, cfg, synth.params_for(cfg), cfg.n_loci, 48, slots=1, place=0)
need = 4 * (run.ns + 64)
cands, spacers = [run.slots[0]["words"]], []
for i in range(13):
    spacers.append(DevBuf(eng, ((37, 301, 1024, 2500, 150, 4097, 611, 1777)[i % 8] << 20) + 4096))
    cands.append(DevBuf(eng, need))
spins = [0, 8, 24, 64]
walk, probe = [], {s: [] for s in spins}
for c in cands:
    run.slots[0]["words"] = c
    walk.append(run._walk_ms(reps=4))
    for s in spins:
        os.environ["SMC_PROBE_SPIN"] = str(s)
        ms = ctypes.c_float()
        assert L.smc_mem_write_probe(eng.ctx, ctypes.c_void_p(c.data_ptr()), need, ctypes.byref(ms)) == 0
        probe[s].append(ms.value)
print("allocation   walk ms   probe ms at spin " + " / ".join(str(s) for s in spins))
for k in range(len(cands)):
    print("#%-2d        %8.3f   %s" % (k, walk[k], "  ".join("%7.3f" % probe[s][k] for s in spins)))
for s in spins:
    print("spin %3d: correlation with the walk %.3f; the probe's best is the walk's #%d of %d" % (
        s, float(np.corrcoef(walk, probe[s])[0, 1]), sorted(walk).index(walk[int(np.argmin(probe[s]))]) + 1, len(walk)))
