"""Soak of the device path FROM ALIGNMENTS: random shapes (barcodes x reads per barcode from a handful to past both sort thresholds,
indel / clip / variant rates, run lengths from less than a tile to dozens) through smc_build_planes -> smc_plan_create_dev ->
smc_plan_run_words against oracle/aln_planes.c + oracle/smc_oracle.c, row by row (dev tool).  usage: fa_soak.py first_seed n_seeds"""
import dataclasses, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
from smcounter_amd import abi, devplanes, engine, synth
from smcounter_amd.params import VcParams
import oracle_lib

first, n = int(sys.argv[1]), int(sys.argv[2])
eng = engine.Engine(0)
cores = min(64, len(os.sched_getaffinity(0)))
bad_total = 0
n_misfit = 0
for seed in range(first, first + n):
    rng = np.random.RandomState(seed)
    kind = seed % 6
    if kind == 0:
        n_umi, rpb, nl = int(rng.randint(1, 12)), int(rng.randint(1, 8)), int(rng.randint(1, 700))
    elif kind == 1:
        n_umi, rpb, nl = int(rng.randint(20, 120)), int(rng.randint(2, 70)), int(rng.randint(60, 2500))
    elif kind == 2:                                   # a tile's window around the small sort's capacity (6,144 entries)
        n_umi, rpb, nl = int(rng.randint(60, 75)), 60, int(rng.randint(64, 900))
    elif kind == 3:                                   # ... around the one-workgroup sort's (16,384)
        n_umi, rpb, nl = int(rng.randint(170, 190)), 60, int(rng.randint(64, 400))
    elif kind == 4:                                   # beyond it: segments of eight tiles, the multi-launch sort, the deep class
        n_umi, rpb, nl = int(rng.randint(1500, 5000)), int(rng.randint(6, 16)), int(rng.randint(30, 140))
    else:
        n_umi, rpb, nl = int(rng.randint(300, 900)), int(rng.randint(8, 30)), int(rng.randint(64, 600))
    cfg = synth.SynthConfig("S%d" % seed, nl, n_umi, rpb, 770000 + seed, start_pos=int(rng.randint(1000, 2_000_000)),
                            p_overlap=float(rng.choice([0.0, 0.43, 0.9])), p_err=float(rng.choice([0.0, 1e-3, 2e-2])),
                            alt_locus_frac=float(rng.choice([0.0, 0.05, 0.5])), alt_af=float(rng.choice([0.02, 0.1, 0.5])))
    P = VcParams(minBQ=int(rng.choice([0, 13, 20, 30])), minMQ=int(rng.choice([0, 30])), mtDepth=max(1, int(n_umi * rng.choice([1.0, 1.0, 2.0]))),
                 rpb=float(rpb), hpLen=8, mismatchThr=float(rng.choice([3.0, 6.0, 100.0])), mtDrop=int(rng.choice([0, 0, 1])), maxMT=0,
                 primerDist=int(rng.choice([2, 20])))
    A = synth.generate_alignments(cfg, nl, P, nthreads=8, p_ins_aln=float(rng.choice([0.0, 0.02, 0.2])), p_del_aln=float(rng.choice([0.0, 0.02, 0.2])),
                                  p_clip=float(rng.choice([0.0, 0.05, 0.4])))
    try:
        rb = devplanes.resident_from_alignments(A, eng, P, all_planes=bool(seed % 2))
    except RuntimeError as e:
        print("seed %d (%d barcodes x %d, %d loci): not taken by the device builder (%s)" % (seed, n_umi, rpb, nl, e))
        continue
    if os.environ.get("FA_SOAK_DUMP") and int(os.environ["FA_SOAK_DUMP"].split(":")[0]) == seed:      # (debugging: SEED:path.npz)
        np.savez(os.environ["FA_SOAK_DUMP"].split(":")[1].replace("alone", "aln").replace("loop", "aln"), aln=A["aln"], loc=A["loc"], cig=A["cig"], start0=A["start0"])
        np.savez(os.environ["FA_SOAK_DUMP"].split(":")[1], words=rb.words.download(np.uint32, rb.n_slots),
                 ustart=rb.planes[4].download(np.uint32, rb.n_ustart), loci=rb.loci)
    d_loci = devplanes.DevLoci(eng, rb.loci)
    # two seeds of three make their plan WITHOUT the host (smc_plan_create_dev_spec: sized from whatever shape came before - most do
    # not fit, say so, and are made again the exact way; the ones that fit must give the same rows)
    spec = P if seed % 3 else None
    plan = eng.make_plan_dev(d_loci, rb.n_loci, spec_params=spec)
    got = plan.run_devbuf([rb.words, rb.planes[4]], P).copy()
    if not plan.ok():
        n_misfit += 1
        plan.close()
        plan = eng.make_plan_dev(d_loci, rb.n_loci, spec_params=spec)
        got = plan.run_devbuf([rb.words, rb.planes[4]], P).copy()
        assert plan.ok()
    plan.close()
    # ... and once more, sized from this very batch's record: it fits, and gives the same bytes
    for attempt in range(2):                  # (sized from an earlier seed's batch it may not fit: then it says so and is made again)
        plan = eng.make_plan_dev(d_loci, rb.n_loci, spec_params=P)
        again = plan.run_devbuf([rb.words, rb.planes[4]], P).copy()
        fits = plan.ok()
        plan.close()
        if fits:
            break
        n_misfit += 1
    assert fits and again.tobytes() == got.tobytes(), "seed %d: the plan made without the host differs (fits: %s)" % (seed, fits)
    d_loci.free()
    db = oracle_lib.aln_planes(A, P, 0, nl, n_threads=cores)
    want, fragile, pi_all = oracle_lib.call_batch_mt(db, abi.c_params(P), abi.ROW_DTYPE, cores, return_fragile=True, return_pi_all=True)
    problems = abi.compare_rows(got, want, 1e-6, 1e-6, fragile, pi_all)
    # (no locus over the barcode cap: without barcode texts both sides keep the ds lowest barcode NUMBERS, and the two builders number differently)
    assert int(db.loci["n_umi"].max()) <= P.ds, "shape puts loci over the barcode cap"
    if not (rb.loci["n_alleles"] == db.loci["n_alleles"]).all():
        problems.append("allele counts differ at loci %r" % np.flatnonzero(rb.loci["n_alleles"] != db.loci["n_alleles"])[:5].tolist())
    for b in [rb.words] + [p for p in rb.planes if p is not None]:
        b.free()
    if problems:
        bad_total += 1
        print("seed %d (%d barcodes x %d, %d loci, depth %.0f): %d problems, first: %s" % (seed, n_umi, rpb, nl, A["reads"] / nl, len(problems), problems[0]), flush=True)
print("from-alignments soak: %d seeds, %d with problems; plans without the host: %s, %d said they did not fit" % (n, bad_total, eng.spec_counts(), n_misfit))
