"""Quick single-GPU timing of the hot path on a slice of a synthetic config (dev tool)."""
import argparse, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from smcounter_amd import synth, engine, abi

ap = argparse.ArgumentParser()
ap.add_argument("--cfg", default="C3"); ap.add_argument("--loci", type=int, default=20000)
ap.add_argument("--iters", type=int, default=5)
a = ap.parse_args()
cfg = synth.CONFIGS[a.cfg]; P = synth.params_for(cfg)
t = time.time(); db = synth.generate_native(cfg, 0, a.loci); print("gen %.1fs" % (time.time() - t))
eng = engine.Engine(0)
planes = eng.upload(db); plan = eng.make_plan(db.loci); rows = plan.alloc_rows()
print("plan", plan.info())
plan.run(planes, P, rows); torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(a.iters + 1)]
ev[0].record()
for i in range(a.iters):
    plan.run(planes, P, rows); ev[i + 1].record()
torch.cuda.synchronize()
ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(a.iters)]
best = min(ms)
bytes_per_locus = 16 * cfg.depth + 360
print("ms/iter", ["%.3f" % m for m in ms])
print("%s: %d loci, %.3f ms -> %.3f M loci/s, algorithmic %.1f GB/s (%.2f%% of 8 TB/s)" % (
    a.cfg, a.loci, best, a.loci / best / 1e3, a.loci * bytes_per_locus / best / 1e6,
    a.loci * bytes_per_locus / best / 1e6 / 8000 * 100))
R = plan.download(rows)
print("status", np.bincount(R["status"] & 0xff), "mean PI_ref", R["pi"].max(axis=1).mean())
