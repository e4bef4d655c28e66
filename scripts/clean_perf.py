import sys, os, dataclasses
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from smcounter_amd import synth, engine
base = synth.CONFIGS["C3"]
for name, kw in (("C3 as is", {}), ("C3 no events (all barcodes single-allele)", dict(p_err=0.0, p_gap=0.0, p_ins=0.0, p_delstart=0.0, p_n=0.0))):
    cfg = dataclasses.replace(base, **kw); P = synth.params_for(cfg)
    db = synth.generate_native(cfg, 0, 40000)
    eng = engine.Engine(0); planes = eng.upload(db); plan = eng.make_plan(db.loci); rows = plan.alloc_rows()
    plan.run(planes, P, rows); torch.cuda.synchronize()
    ts = []
    for _ in range(6):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); plan.run(planes, P, rows); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    print("%-50s %.3f ms" % (name, min(ts)))
