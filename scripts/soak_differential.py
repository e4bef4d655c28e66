"""More seeds of tests/test_gpu_parity.py::test_randomised_differential in one process (dev tool, GPU box): random stress batches
under random parameters, every field of every row against the CPU restatement.  usage: soak_differential.py FIRST LAST"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle_lib
oracle_lib.build()
from smcounter_amd import engine
import test_gpu_parity as T
eng = engine.Engine(0)
a, b = int(sys.argv[1]), int(sys.argv[2])
bad = []
for seed in range(a, b):
    try:
        T.test_randomised_differential(eng, seed)
    except AssertionError as e:
        bad.append(seed); print("seed", seed, "FAILED:", str(e)[:300], flush=True)
print("seeds %d..%d: %d failed %s" % (a, b - 1, len(bad), bad))
# ---- loci with a variant (barcodes whose fragments all show one other allele, mixed barcodes), at several depths and shares
import numpy as np
from smcounter_amd import synth, abi
n_bad = 0
k = 0
for af in (0.02, 0.1, 0.5, 0.95):
    for umi, rpb in ((20, 3), (50, 14), (50, 60), (400, 9), (3000, 9)):
        k += 1
        n = 400 if umi * rpb < 5000 else 40
        cfg = synth.SynthConfig("soak%d" % k, n, umi, rpb, 20171000 + k, alt_locus_frac=0.5, alt_af=af)
        P = synth.params_for(cfg)
        db = synth.generate_native(cfg, 0, n, P)
        plan = eng.make_plan(db.loci)
        got = plan.download(plan.run(eng.upload(db), P))
        plan.close()
        want, fragile = oracle_lib.call_batch(db, abi.c_params(P), abi.ROW_DTYPE, return_fragile=True)
        d = abi.compare_rows(got, want, 1e-6, 1e-6, fragile)
        n_bad += len(d)
        print("variant share %.2f, %d barcodes x %d reads, %d loci (%d to the filters): %d differences, PI max-abs-diff %.2e" % (
            af, umi, rpb, n, int((got["cand"][:, 0]["flt_applied"] != 0).sum()), len(d), float(np.abs(got["pi"] - want["pi"]).max())), flush=True)
print("variant shapes: %d differences in all" % n_bad)
