"""Soak: random locus shapes from the native generator (barcodes x reads per barcode, every launch class), GPU rows
against the CPU restatement (dev tool).  usage: shape_soak.py first_seed n_seeds"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
from smcounter_amd import abi, engine, synth
from smcounter_amd.params import VcParams
import oracle_lib

first, n = int(sys.argv[1]), int(sys.argv[2])
eng = engine.Engine(0)
bad_total = 0
for seed in range(first, first + n):
    rng = np.random.RandomState(seed)
    n_umi = int(rng.choice([1, 2, 7, 30, 64, 65, 129, 400, 1500, 5000]))
    rpb = int(rng.choice([1, 2, 3, 9, 20, 60, 150]))
    while n_umi * rpb >= (1 << 18):
        rpb = max(1, rpb // 2)
    n_loci = int(max(1, min(64, 400000 // (n_umi * rpb))))
    cfg = synth.SynthConfig("s%d" % seed, n_loci, n_umi, rpb, seed, p_err=float(rng.choice([1e-3, 2e-2])),
                            p_gap=float(rng.choice([2e-3, 3e-2])), alt_locus_frac=float(rng.choice([0.0, 0.5])),
                            alt_af=float(rng.choice([0.01, 0.3])))
    P = VcParams(minBQ=int(rng.choice([20, 30])), minMQ=30, mtDepth=int(rng.choice([max(1, n_umi // 3), n_umi])),
                 rpb=float(rpb), hpLen=8, mismatchThr=6.0, mtDrop=int(rng.choice([0, 1])), maxMT=0,
                 primerDist=int(rng.choice([2, 30])))
    db = synth.generate_native(cfg, 0, n_loci, P)
    got = eng.call_batch_host(db, P)
    want, fragile, pi_all = oracle_lib.call_batch(db, abi.c_params(P), abi.ROW_DTYPE, return_fragile=True, return_pi_all=True)
    bad = abi.compare_rows(got, want, 1e-6, 1e-6, fragile, pi_all)
    # the general path's shortcut for barcodes with many reference fragments against the full walk: integer columns equal, PI
    # far inside the tolerance (a few 1e-12 per barcode: 1e-8 holds for thousands of barcodes)
    os.environ["SMC_NO_LITE"] = "1"
    full = eng.call_batch_host(db, P)
    del os.environ["SMC_NO_LITE"]
    bad += ["lite vs full walk: " + x for x in abi.compare_rows(got, full, 1e-8, 1e-12, fragile, pi_all)]
    if bad:
        bad_total += 1
        print("seed", seed, "shape", n_umi, "x", rpb, "loci", n_loci, "MISMATCH", bad[:3], flush=True)
print("shape soak: %d seeds, %d with mismatches" % (n, bad_total))
