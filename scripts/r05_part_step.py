"""The step (and its walk) at parts of 64 ... 256 rows, on three allocations (dev tool)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["SMC_VMM_CHUNK_MB"] = "0"
from smcounter_amd import synth, engine
from smcounter_amd.engine import DevBuf
import bench_fa
eng = engine.Engine(0)
L = eng.L
cfg = synth.CONFIGS["C3"]
run = bench_fa.AlignmentRun(eng, cfg, synth.params_for(cfg), 200000, 48, slots=1, place=0)
cands = [run.slots[0]["words"]] + [DevBuf(eng, 4 * (run.ns + 64)) for _ in range(3)]
print("%-14s %s" % ("allocation", "".join("%22s" % ("part %d: walk / step" % p) for p in (128, 256, 512, 1024))))
for k, c in enumerate(cands):
    row = []
    for p in (128, 256, 512, 1024):
        os.environ["SMC_BP_PART"] = str(p)
        run.slots[0]["words"] = c
        w = run._walk_ms(reps=4)
        for _ in range(3):
            run.step(slot=0)
        L.smc_device_sync(eng.ctx)
        t0 = time.perf_counter()
        for _ in range(15):
            run.step(slot=0)
        L.smc_device_sync(eng.ctx)
        row.append("%10.3f / %.3f" % (w, (time.perf_counter() - t0) / 15 * 1e3))
    print("hipMalloc #%-3d %s" % (k, "".join("%22s" % x for x in row)), flush=True)
