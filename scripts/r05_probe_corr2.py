"""smc_mem_write_probe against the walk itself over 40 virtual-memory-backed allocations (64 / 32 / 128 MB handles): does the probe
single out the rare top class (walk ~ 1.19 ms)?  And what does a candidate cost (allocate + probe)?  (dev tool)"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from smcounter_amd import synth, engine
from smcounter_amd.engine import DevBuf
import bench_fa
eng = engine.Engine(0)
eng.alloc_tries = 1
L = eng.L
cfg = synth.CONFIGS["C3"]
run = bench_fa.AlignmentRun(eng, cfg, synth.params_for(cfg), cfg.n_loci, 48, slots=1, place=0)
need = 4 * (run.ns + 64)
cands, t_alloc = [], []
for i in range(40):
    os.environ["SMC_VMM_CHUNK_MB"] = str((64, 32, 128, 64)[i % 4])
    t0 = time.perf_counter()
    cands.append(DevBuf(eng, need))
    t_alloc.append(time.perf_counter() - t0)
spins = [0, 4, 8]
walk, probe, t_probe = [], {s: [] for s in spins}, []
for c in cands:
    run.slots[0]["words"] = c
    walk.append(run._walk_ms(reps=3))
    for s in spins:
        os.environ["SMC_PROBE_SPIN"] = str(s)
        ms = ctypes.c_float()
        t0 = time.perf_counter()
        assert L.smc_mem_write_probe(eng.ctx, ctypes.c_void_p(c.data_ptr()), need, ctypes.byref(ms)) == 0
        t_probe.append(time.perf_counter() - t0)
        probe[s].append(ms.value)
order = np.argsort(walk)
print("allocation (by walk time)   walk ms   probe ms at spin " + " / ".join(str(s) for s in spins))
for k in order:
    print("#%-2d (%3s MB handles)     %8.3f   %s" % (k, (64, 32, 128, 64)[k % 4], walk[k], "  ".join("%7.3f" % probe[s][k] for s in spins)))
for s in spins:
    print("spin %3d: correlation %.3f; the probe's best is the walk's #%d of %d (%.3f ms; the walk's best %.3f)" % (
        s, float(np.corrcoef(walk, probe[s])[0, 1]), sorted(walk).index(walk[int(np.argmin(probe[s]))]) + 1, len(walk),
        walk[int(np.argmin(probe[s]))], min(walk)))
print("a candidate costs: allocate %.1f ms (median), probe %.1f ms" % (1e3 * float(np.median(t_alloc)), 1e3 * float(np.median(t_probe))))
