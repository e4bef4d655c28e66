"""Soak: many random stress batches under random parameters, GPU rows against the CPU restatement (dev tool).
usage: fuzz_soak.py first_seed n_seeds"""
import dataclasses, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
from smcounter_amd import abi, engine, features, synth
from smcounter_amd.params import VcParams
import oracle_lib

first, n = int(sys.argv[1]), int(sys.argv[2])
eng = engine.Engine(0)
bad_total = 0
for seed in range(first, first + n):
    rng = np.random.RandomState(seed)
    deep = bool(seed % 5 == 0)
    pb, _ = synth.generate_stress(int(rng.randint(10, 120)) if not deep else 4, seed, deep=deep,
                                  max_umi=int(rng.choice([2, 8, 30, 90, 200])))
    P = VcParams(mtDepth=int(rng.choice([2, 9, 40, 5000])), rpb=float(rng.choice([1.2, 2.5, 3.0, 8.6])), hpLen=8,
                 mtDrop=int(rng.choice([0, 0, 1, 2, 3])), minBQ=int(rng.choice([0, 2, 20, 25, 30])),
                 minMQ=int(rng.choice([0, 20, 30, 60])), mismatchThr=float(rng.choice([0.5, 2.0, 6.0, 100.0])),
                 maxMT=int(rng.choice([0, 0, 0, 3, 11])), primerDist=int(rng.choice([0, 2, 10, 50])))
    if seed % 2:
        names = [["B%05d_%d" % (u, l) for u in range(int(pb.umi[pb.locus_slice(l)].max()) + 1
                                                         if pb.read_off[l + 1] > pb.read_off[l] else 0)]
                 for l in range(pb.n_loci)]
        pb = dataclasses.replace(pb, umi_names=names)
    db = features.extract_features(pb, P)
    got = eng.call_batch_host(db, P)
    want, fragile, pi_all = oracle_lib.call_batch(db, abi.c_params(P), abi.ROW_DTYPE, return_fragile=True, return_pi_all=True)
    bad = abi.compare_rows(got, want, 1e-6, 1e-6, fragile, pi_all)
    if bad:
        bad_total += 1
        print("seed", seed, "MISMATCH", bad[:3], flush=True)
print("soak: %d seeds, %d with mismatches" % (n, bad_total))
