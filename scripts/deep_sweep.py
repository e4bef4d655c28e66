"""Dev tool: sweep the deep class's geometry (parts, chunk sizes; read by smc_plan_create from the environment) on the
deep shapes.  usage: deep_sweep.py [CFG:N ...]"""
import os, sys, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from smcounter_amd import synth, engine
specs = sys.argv[1:] or ["X9:1000", "X10:300", "X1:2000"]
eng = engine.Engine(0)
for spec in specs:
    name, n = spec.split(":"); n = int(n)
    cfg = synth.CONFIGS[name]; P = synth.params_for(cfg)
    db = synth.generate_native(cfg, 0, n)
    planes = eng.upload(db)
    bpl = 16 * cfg.depth + 360
    geos = [tuple(int(x) for x in g.split(",")) for g in os.environ.get("GEOS", "16384,4352,2048;18432,4864,2304;24576,6400,3072;32768,8448,4096;16384,4352,1792;15360,4096,2048").split(";")]
    for part, fcap, ucap in geos:
        os.environ.update(SMC_DEEP_PART_READS=str(part), SMC_DEEP_FCAP=str(fcap), SMC_DEEP_UCAP=str(ucap))
        plan = eng.make_plan(db.loci); rows = plan.alloc_rows()
        plan.run(planes, P, rows); torch.cuda.synchronize()
        ms = []
        for _ in range(7):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); plan.run(planes, P, rows); b.record(); torch.cuda.synchronize(); ms.append(a.elapsed_time(b))
        m = float(np.median(ms))
        print("%s part_reads %6d fcap %5d ucap %5d: %.3f ms  %.1f%% of 8 TB/s" % (name, part, fcap, ucap, m, n * bpl / m / 1e6 / 80))
        plan.close()
