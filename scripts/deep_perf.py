"""Timing on a locus shape like the reference's example run (58k reads, ~4000 barcodes x 14 reads)."""
import sys, os, dataclasses, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from smcounter_amd import synth, engine
for name, cfg in (("deep 4000x14", synth.SynthConfig("deep", 512, 4000, 14, 7)), ("mid 500x20", synth.SynthConfig("mid", 4000, 500, 20, 8))):
    P = synth.params_for(cfg)
    db = synth.generate_native(cfg, 0, cfg.n_loci)
    eng = engine.Engine(0); planes = eng.upload(db); plan = eng.make_plan(db.loci); rows = plan.alloc_rows()
    plan.run(planes, P, rows); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); plan.run(planes, P, rows); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    reads = cfg.n_loci * cfg.depth
    print("%-14s SMC_KERNEL=%-7s %.3f ms  %.1f G reads/s  %.0f loci/s  (%.1f%% of 8 TB/s)" % (
        name, os.environ.get("SMC_KERNEL", "sorted"), min(ts), reads / min(ts) / 1e6, cfg.n_loci / min(ts) * 1e3, reads * 16 / min(ts) / 1e6 / 80))
