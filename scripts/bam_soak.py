"""Soak through the BAM path: random BAMs (random CIGARs with S/M/I/D/N, two references, with / without .bai),
native decoder -> device batch -> GPU rows against the CPU restatement, the native batch against the Python
decoder's, and the planes built on the device (k_build_planes.inc) against the host-built ones up to barcode / fragment
numbering (planecheck.py) (dev tool).  usage: bam_soak.py first_seed n_seeds"""
import os, pathlib, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from smcounter_amd import abi, bamio, engine, fasta, features, pileup
from smcounter_amd.params import VcParams
import oracle_lib
import test_bamio

first, n = int(sys.argv[1]), int(sys.argv[2])
eng = engine.Engine(0)
bad_total = 0
for seed in range(first, first + n):
    rng = np.random.RandomState(seed)
    tmp = pathlib.Path(tempfile.mkdtemp())
    bam, fa_path, loci = test_bamio._random_bam(tmp, seed, bool(seed % 2))
    fa = fasta.FastaFile(fa_path)
    P = VcParams(mtDepth=int(rng.choice([4, 12, 100])), rpb=float(rng.choice([1.5, 3.0])), hpLen=8,
                 minBQ=int(rng.choice([10, 20, 30])), minMQ=int(rng.choice([0, 30])), mismatchThr=float(rng.choice([4.0, 100.0])),
                 mtDrop=int(rng.choice([0, 1])), primerDist=int(rng.choice([2, 20])))
    mr = int(rng.choice([3000, 2_000_000]))
    want_pb = pileup.concat([b for _, b in bamio.iter_pileup_batches(bamio.BamFile(bam), fa, loci, max_reads=mr)])
    dbw = features.extract_features(want_pb, P)
    problems = []
    off = 0
    for _, db in bamio.iter_device_batches_native(bam, fa, loci, P, max_reads=mr, nthreads=int(rng.choice([1, 4]))):
        got = eng.call_batch_host(db, P)
        want, fragile, pi_all = oracle_lib.call_batch(db, abi.c_params(P), abi.ROW_DTYPE, return_fragile=True, return_pi_all=True)
        problems += abi.compare_rows(got, want, 1e-6, 1e-6, fragile, pi_all)
        for l in range(db.n_loci):
            o, k = db.read_off(l), int(db.loci["n_reads"][l])
            ow = dbw.read_off(off + l)
            for name in ("meta", "umi", "frag", "dist"):
                if not np.array_equal(getattr(db, name)[o:o + k], getattr(dbw, name)[ow:ow + k]):
                    problems.append("decoders differ at locus %d plane %s" % (off + l, name))
        off += db.n_loci
    # the device plane builder (smc_bam_alignments + k_build_planes) against the host builder: the same bytes, batch by batch
    from smcounter_amd import devplanes
    nt = int(rng.choice([1, 4]))
    host = list(bamio.iter_device_batches_native(bam, fa, loci, P, max_reads=mr, nthreads=nt))
    dev = list(devplanes.iter_resident_batches(bam, fa, loci, P, eng, max_reads=mr, nthreads=nt))
    if len(host) != len(dev):
        problems.append("device planes: %d batches, host %d" % (len(dev), len(host)))
    for (f1, hb), (f2, rb) in zip(host, dev):
        d = rb.to_host()
        if f1 != f2 or d.n_loci != hb.n_loci:
            problems.append("device planes: batch boundaries differ"); break
        from smcounter_amd import planecheck
        problems += ["device planes: " + x for x in planecheck.differences(d, hb)]
    if problems:
        bad_total += 1
        print("seed", seed, "PROBLEM", problems[:3], flush=True)
print("bam soak: %d seeds, %d with problems" % (n, bad_total))
