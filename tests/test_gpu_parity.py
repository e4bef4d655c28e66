"""Parity of the HIP path (through the C ABI) against the CPU restatement and the reference's golden
vectors.  Integer columns bit-exact; PI and Fisher p-values within 1e-6 (BASELINE.json)."""
import os

import numpy as np
import pytest

from conftest import golden_files, load_golden
from smcounter_amd import abi, features, pileup, rows, synth
from smcounter_amd.params import VcParams

import oracle_lib

pytestmark = pytest.mark.gpu
PI_TOL = 1e-6
P_TOL = 1e-6


@pytest.fixture(autouse=True)
def _experiment_switches(monkeypatch):
    """The environment switches these tests flip (chunk geometries, poisoned scratch, forced code paths) are experiment knobs: the
    libraries read them only under SMC_EXPERIMENTAL."""
    monkeypatch.setenv("SMC_EXPERIMENTAL", "1")


@pytest.mark.parametrize("path", golden_files(), ids=os.path.basename)
def test_golden_rows_vs_oracle_and_reference(engine0, path):
    pb, db, P, refp, expected = load_golden(path)
    got = engine0.call_batch_host(db, P)
    want, fragile = oracle_lib.call_batch(db, abi.c_params(P), abi.ROW_DTYPE, return_fragile=True)
    assert abi.compare_rows(got, want, PI_TOL, P_TOL, fragile) == []
    text = rows.format_rows(got, db, P, refp)
    ties = abi.near_tie_loci(got, want)
    flagged = ((got["status"] | want["status"]) & abi.ST_UNDERFLOW) != 0
    if os.path.basename(path) == "stress_giant.npz":
        # barcodes of 4,200 / 6,600 / 6,800 / 7,400 unpaired fragments: the first is inside the double range (and must
        # match the reference's strings; 0.9^6600 = 1e-302 is where the 1e-6 PCR terms go denormal), the others are flagged by BOTH
        assert flagged.tolist() == [False, True, True, True]
        assert ((got["status"] & abi.ST_UNDERFLOW) != 0).tolist() == ((want["status"] & abi.ST_UNDERFLOW) != 0).tolist()
    for l, (t, e) in enumerate(zip(text, expected)):
        if e["tie_ambiguous"] or l in ties or fragile[l] or flagged[l]:
            continue        # unpinned by the reference algorithm itself (see abi.compare_rows)
        assert t == e["row"], "locus %d differs from the reference's own output" % l
        if e["pi_raw"]:
            d = max(max(abs(e["pi_raw"][k] - got["pi"][l][k]) for k in range(4)),
                    abs(e["pi_raw"][4] - got["cand"][l][0]["pi"]))
            assert d <= PI_TOL


# C2 / C3 / X6 / X4 / X5 / X1: one shape per workgroup-size class of the launch plan (64 / 128 / 256 / 512 / 512 /
# 1024 threads, host_abi.inc), C5 and X2 sit right above the 128 -> 256 threshold
@pytest.mark.parametrize("name,n", [("C2", 3000), ("C3", 1500), ("C5", 300), ("X6", 300), ("X2", 200), ("X4", 100),
                                    ("X5", 80), ("X1", 40)])
def test_synthetic_configs_vs_oracle(engine0, name, n):
    cfg = synth.CONFIGS[name]
    P = synth.params_for(cfg)
    db = synth.generate_native(cfg, 0, n, P)
    planes = engine0.upload(db)
    plan = engine0.make_plan(db.loci)
    got = plan.download(plan.run(planes, P))
    want, fragile = oracle_lib.call_batch(db, abi.c_params(P), abi.ROW_DTYPE, return_fragile=True)
    assert (fragile > 0).mean() <= 0.01          # realistic error rates: rounding-decided barcodes are rare
    assert abi.compare_rows(got, want, PI_TOL, P_TOL, fragile) == []
    assert float(np.abs(got["pi"] - want["pi"]).max()) <= PI_TOL
    # run-to-run bit reproducibility (fixed-point PI accumulation, no float atomics)
    again = plan.download(plan.run(planes, P))
    assert again.tobytes() == got.tobytes()
    plan.close()


def test_a_plan_run_again_and_again_with_every_locus_on_the_filter_worklist(engine0):
    """k_filter_loci's last workgroup leaves the worklist empty for the plan's next run (no memset per run): a batch whose
    every locus has a candidate, so that a count left standing would run the list past its end on the second run; the
    host-made plan (zeroed before its first run) and the device-made one (zeroed by k_plan_classify)."""
    import torch
    cfg = synth.SynthConfig("allvar", 700, 40, 30, 20170501, alt_locus_frac=1.0, alt_af=0.3)
    P = synth.params_for(cfg)
    db = synth.generate_native(cfg, 0, cfg.n_loci, P)
    want, fragile = oracle_lib.call_batch(db, abi.c_params(P), abi.ROW_DTYPE, return_fragile=True)
    assert int((want["cand"][:, 0]["flt_applied"] != 0).sum()) > 600
    planes = engine0.upload(db)
    d_loci = torch.from_numpy(np.ascontiguousarray(db.loci).view(np.uint8)).to("cuda:0")
    for plan in (engine0.make_plan(db.loci), engine0.make_plan_dev(d_loci, cfg.n_loci)):
        first = None
        for _ in range(4):
            got = plan.download(plan.run(planes, P))
            assert abi.compare_rows(got, want, PI_TOL, P_TOL, fragile) == []
            first = got.tobytes() if first is None else first
            assert got.tobytes() == first
        plan.close()


def test_stress_mix_of_sizes_and_bins(engine0):
    """Loci of very different sizes in one batch: exercises every launch bin, including the
    global-scratch one, and the zero-coverage / tiny loci."""
    parts = []
    a, _ = synth.generate_stress(120, 21)
    b, _ = synth.generate_stress(6, 22, deep=True, scenarios=("snp", "discord", "multi", "biallelic"))
    c, _ = synth.generate_stress(2, 23, deep=True, max_umi=60, scenarios=("snp",))
    pb = pileup.concat([a, b, c])
    P = VcParams(mtDepth=5000, rpb=3.0, hpLen=8, mtDrop=1)
    db = features.extract_features(pb, P)
    got = engine0.call_batch_host(db, P)
    want, fragile = oracle_lib.call_batch(db, abi.c_params(P), abi.ROW_DTYPE, return_fragile=True)
    assert abi.compare_rows(got, want, PI_TOL, P_TOL, fragile) == []


def test_deep_locus_takes_the_parts_path(engine0):
    """Loci beyond the whole-locus classes (> 24 k reads) are split into parts (shares of the barcodes, one workgroup each)
    that add their headers into a global accumulator; the plan reports that scratch."""
    cfg = synth.SynthConfig("big", 2, 7000, 12, 99)
    P = synth.params_for(cfg)
    db = synth.generate_native(cfg, 0, 2, P)
    assert int(db.loci["n_reads"][0]) > 24576
    plan = engine0.make_plan(db.loci)
    assert plan.info()[1] > 0                       # bytes of accumulators + flag scratch
    got = engine0.call_batch_host(db, P)
    want, fragile = oracle_lib.call_batch(db, abi.c_params(P), abi.ROW_DTYPE, return_fragile=True)
    assert abi.compare_rows(got, want, PI_TOL, P_TOL, fragile) == []


def test_parts_and_chunks_give_identical_rows(monkeypatch):
    """Deep loci: several workgroups per locus ("parts") x chunks of whole barcodes per part.  Everything
    a chunk produces is additive in integers, so the rows must be BIT-IDENTICAL however the locus is cut: (a) default
    geometry, one workgroup for the whole locus, many small parts with tiny chunks, and the whole-locus class (no parts at
    all); (b) one barcode larger than a chunk (its flag bytes go to global scratch); (c) more barcodes than the cap
    without the host's sampling marks (the stand-in needs the whole locus: one workgroup takes it, chunk after chunk);
    (d) a contract violation inside a later part; (e) host sampling marks across parts."""
    import dataclasses
    from smcounter_amd import engine
    cfg = synth.SynthConfig("big", 3, 7000, 12, 99)
    P = synth.params_for(cfg)
    db = synth.generate_native(cfg, 0, 3, P)
    assert int(db.loci["n_reads"][0]) > 24576
    want, fragile = oracle_lib.call_batch(db, abi.c_params(P), abi.ROW_DTYPE, return_fragile=True)
    eng = engine.Engine(0)
    base = eng.call_batch_host(db, P)
    assert abi.compare_rows(base, want, PI_TOL, P_TOL, fragile) == []
    geometries = [dict(SMC_DEEP_PART_READS="100000000"),                                         # one part, chunks of 16 k reads
                  dict(SMC_DEEP_PART_READS="3000", SMC_DEEP_FCAP="512", SMC_DEEP_UCAP="97"),    # 28 parts, small chunks
                  dict(SMC_DEEP_PART_READS="100000000", SMC_DEEP_FCAP="40", SMC_DEEP_UCAP="5"), # one part, ~600 chunks
                  dict(SMC_DEEP_FROM_READS="100")]                                              # everything deep, incl. tiny loci
    for env in geometries:
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        got = eng.call_batch_host(db, P)
        for k in env:
            monkeypatch.delenv(k)
        assert got.tobytes() == base.tobytes(), env
    # the small shapes through the deep class as well (parts of a 300-read locus)
    small = synth.generate_native(synth.CONFIGS["C2"], 0, 500, synth.params_for(synth.CONFIGS["C2"]))
    Ps = synth.params_for(synth.CONFIGS["C2"])
    ref_rows = eng.call_batch_host(small, Ps)
    monkeypatch.setenv("SMC_DEEP_FROM_READS", "1"); monkeypatch.setenv("SMC_DEEP_PART_READS", "64"); monkeypatch.setenv("SMC_DEEP_FCAP", "16")
    assert eng.call_batch_host(small, Ps).tobytes() == ref_rows.tobytes()
    for k in ("SMC_DEEP_FROM_READS", "SMC_DEEP_PART_READS", "SMC_DEEP_FCAP"):
        monkeypatch.delenv(k)
    # (b) 5 barcodes x 12,400 reads each: one barcode is more than a 16 k-read chunk holds only with smaller chunks - force
    # them: 8,600 fragments per barcode.  (Not deeper: from ~6,700 unpaired fragments in ONE barcode on, the likelihood
    # products of calProb leave the double range and what is left depends on the multiplication order - sequential in the
    # reference, a tree here; no real barcode is near.)
    cfg_b = synth.SynthConfig("giant2", 2, 5, 12400, 4322)
    Pb = synth.params_for(cfg_b)
    dbb = synth.generate_native(cfg_b, 0, 2, Pb)
    wb, fb = oracle_lib.call_batch(dbb, abi.c_params(Pb), abi.ROW_DTYPE, return_fragile=True)
    got = eng.call_batch_host(dbb, Pb)
    assert abi.compare_rows(got, wb, PI_TOL, P_TOL, fb) == []
    monkeypatch.setenv("SMC_DEEP_FCAP", "1024")               # 4 k reads per chunk < 12.4 k reads of one barcode
    got2 = eng.call_batch_host(dbb, Pb)
    monkeypatch.delenv("SMC_DEEP_FCAP")
    assert got2.tobytes() == got.tobytes()
    # (c) ds = 2 * mtDepth below the barcode count, no marks
    Pc = dataclasses.replace(P, mtDepth=1000)
    got = eng.call_batch_host(db, Pc)
    wc = oracle_lib.call_batch(db, abi.c_params(Pc), abi.ROW_DTYPE)
    assert (got["status"] & abi.ST_DOWNSAMPLED).all() and (got["used_mt"] == 2000).all()
    assert abi.compare_rows(got, wc, PI_TOL, P_TOL) == []
    # (d) a fragment slot out of range in the last third of locus 1: that row is flagged, the others are untouched
    bad = dataclasses.replace(db, frag=db.frag.copy())
    o = 4 * int(db.loci["read_off4"][1]) + int(db.loci["n_reads"][1]) * 5 // 6
    bad.frag[o] = (bad.frag[o] & ~np.uint32(features.FRAG_SLOT_MASK)) | np.uint32(int(db.loci["n_frag"][1]) + 7)
    got = eng.call_batch_host(bad, P)
    assert got["status"][1] & abi.ST_BAD_INPUT
    assert got[[0, 2]].tobytes() == base[[0, 2]].tobytes()
    # (e) marks: drop every third barcode of locus 0 down to exactly ds
    Pe = dataclasses.replace(P, mtDepth=2500)                   # ds = 5000 < 7000
    us = db.umi_start.copy()
    o, nu = int(db.loci["umi_off"][0]), int(db.loci["n_umi"][0])
    drop = np.arange(nu)[::3][:nu - 5000]
    us[o + drop] |= np.uint32(features.USTART_DROPPED)
    loci = db.loci.copy()
    loci["flags"][0] |= features.LF_SAMPLED
    dbe = dataclasses.replace(db, umi_start=us, loci=loci)
    dbe = dataclasses.replace(dbe, loci=dbe.loci[:1].copy())     # (only the marked locus: the others would need marks too)
    got = eng.call_batch_host(dbe, Pe)
    we, fe = oracle_lib.call_batch(dbe, abi.c_params(Pe), abi.ROW_DTYPE, return_fragile=True)
    assert int(got["used_mt"][0]) == 5000 and not (got["status"][0] & abi.ST_BAD_INPUT)
    assert abi.compare_rows(got, we, PI_TOL, P_TOL, fe) == []
    eng.close()


def test_general_path_with_and_without_the_reference_fragment_shortcut(monkeypatch):
    """A queued barcode with P.lite_from or more reference-allele fragments is scored from the odds of its OTHER fragments
    only (calProb's likelihoods all carry rightP, smCounter.py:62-91; host_abi.inc lite_from_for): the rows must equal the
    full walk's (SMC_NO_LITE=1) in every integer column, PI far inside the tolerance; below the bound (C2's ~7 fragments
    per barcode; minBQ = 0, where the bound does not exist) nothing may change at all."""
    import dataclasses
    from smcounter_amd import engine
    eng = engine.Engine(0)
    for name, n, same_bytes in (("C3", 600, False), ("X3", 400, False), ("C5", 200, False), ("C2", 2000, True)):
        cfg = synth.CONFIGS[name]
        P = synth.params_for(cfg)
        db = synth.generate_native(cfg, 0, n, P)
        lite = eng.call_batch_host(db, P)
        monkeypatch.setenv("SMC_NO_LITE", "1")
        full = eng.call_batch_host(db, P)
        monkeypatch.delenv("SMC_NO_LITE")
        if same_bytes:
            assert lite.tobytes() == full.tobytes(), name
            continue
        assert lite.tobytes() != full.tobytes(), name           # (the shortcut was taken)
        assert abi.compare_rows(lite, full, 1e-9, 1e-12) == [], name
        want, fragile = oracle_lib.call_batch(db, abi.c_params(P), abi.ROW_DTYPE, return_fragile=True)
        assert abi.compare_rows(lite, want, PI_TOL, P_TOL, fragile) == [], name
    cfg = synth.CONFIGS["C3"]
    P0 = dataclasses.replace(synth.params_for(cfg), minBQ=0)
    db = synth.generate_native(cfg, 0, 300, P0)
    lite = eng.call_batch_host(db, P0)
    monkeypatch.setenv("SMC_NO_LITE", "1")
    assert eng.call_batch_host(db, P0).tobytes() == lite.tobytes()
    monkeypatch.delenv("SMC_NO_LITE")
    eng.close()


def test_one_allele_barcodes_of_the_major_other_allele(monkeypatch):
    """Deep loci (> 24576 reads) guess their major other allele from 64 sampled reads and score the barcodes whose live fragments
    all show it from the per-count table, like the reference allele's (calProb's posteriors of a one-allele barcode depend on the
    fragment count only).  With the shortcut (default) and without (SMC_NO_ALT=1): every integer column equal, PI far inside the
    tolerance - on the example run's shape (58 k reads, 30 % of the loci with a variant: the shortcut is taken) and on a deep
    shape without variants; both against the CPU restatement.  Shallow loci are not touched by it at all."""
    from smcounter_amd import engine
    eng = engine.Engine(0)
    for name, n, taken in (("EX", 40, True), ("X9", 24, None), ("C3", 300, False)):
        cfg = synth.CONFIGS[name]
        P = synth.params_for(cfg)
        db = synth.generate_native(cfg, 0, n, P)
        alt = eng.call_batch_host(db, P)
        monkeypatch.setenv("SMC_NO_ALT", "1")
        full = eng.call_batch_host(db, P)
        monkeypatch.delenv("SMC_NO_ALT")
        if taken is True:
            assert alt.tobytes() != full.tobytes(), name          # (the shortcut was taken)
        elif taken is False:
            assert alt.tobytes() == full.tobytes(), name
        assert abi.compare_rows(alt, full, 1e-9, 1e-12) == [], name
        want, fragile = oracle_lib.call_batch(db, abi.c_params(P), abi.ROW_DTYPE, return_fragile=True)
        assert abi.compare_rows(alt, want, PI_TOL, P_TOL, fragile) == [], name
    eng.close()


def test_locus_above_2_to_18_reads(engine0):
    """312,000 reads on one locus (the kernel takes up to 2^24; pysam's max_depth in the reference is 10^6): 22 parts,
    rows equal to the CPU restatement."""
    cfg = synth.SynthConfig("deep", 2, 5200, 60, 777)
    P = synth.params_for(cfg)
    db = synth.generate_native(cfg, 0, 2, P)
    assert int(db.loci["n_reads"].min()) > (1 << 18)
    got = engine0.call_batch_host(db, P)
    want, fragile = oracle_lib.call_batch(db, abi.c_params(P), abi.ROW_DTYPE, return_fragile=True)
    assert abi.compare_rows(got, want, PI_TOL, P_TOL, fragile) == []
    assert (got["cvg"] == 312000).all() and (got["used_mt"] == 5200).all()


def test_long_scan_and_giant_barcode(engine0):
    """(a) a locus scanned in more than 7 steps per wavefront: the packed 5-bit tally accumulators are widened on the
    way; (b) a barcode with more fragments than the per-count posterior table holds (>= 4096): scored by the general
    path although it shows one allele."""
    cfg = synth.SynthConfig("steps8", 40, 70, 60, 1234)
    P = synth.params_for(cfg)
    db = synth.generate_native(cfg, 0, 40, P)
    assert (db.loci["n_reads"] > 7 * 512).all()
    got = engine0.call_batch_host(db, P)
    want, fragile = oracle_lib.call_batch(db, abi.c_params(P), abi.ROW_DTYPE, return_fragile=True)
    assert abi.compare_rows(got, want, PI_TOL, P_TOL, fragile) == []
    cfg = synth.SynthConfig("giant", 3, 2, 6500, 4321)
    P = synth.params_for(cfg)
    db = synth.generate_native(cfg, 0, 3, P)
    assert (db.loci["n_frag"] > 2 * 4096).all()
    got = engine0.call_batch_host(db, P)
    want, fragile = oracle_lib.call_batch(db, abi.c_params(P), abi.ROW_DTYPE, return_fragile=True)
    assert abi.compare_rows(got, want, PI_TOL, P_TOL, fragile) == []


@pytest.mark.parametrize("seed", range(40, 52))
def test_randomised_differential(engine0, seed):
    """Random stress batches under random parameters (quality / mapping-quality cut-offs, mtDrop, barcode cap with and
    without barcode texts, deep and shallow mixes): every field of every row against the CPU restatement."""
    rng = np.random.RandomState(seed)
    deep = bool(seed % 3 == 0)
    pb, _ = synth.generate_stress(int(rng.randint(20, 90)) if not deep else 5, seed, deep=deep,
                                  max_umi=int(rng.choice([4, 20, 60, 150])))
    P = VcParams(mtDepth=int(rng.choice([3, 40, 5000])), rpb=float(rng.choice([1.5, 3.0, 8.6])), hpLen=8,
                 mtDrop=int(rng.choice([0, 0, 1, 2])), minBQ=int(rng.choice([2, 20, 30])), minMQ=int(rng.choice([0, 30])),
                 mismatchThr=float(rng.choice([2.0, 6.0, 100.0])), maxMT=int(rng.choice([0, 0, 7])),
                 primerDist=int(rng.choice([0, 2, 10])))
    if seed % 2:
        import dataclasses
        names = [["BC%05d_%d" % (u, l) for u in range(int(pb.umi[pb.locus_slice(l)].max()) + 1 if pb.read_off[l + 1] > pb.read_off[l] else 0)]
                 for l in range(pb.n_loci)]
        pb = dataclasses.replace(pb, umi_names=names)     # host-side py2 down-sampling where over the cap
    db = features.extract_features(pb, P)
    got = engine0.call_batch_host(db, P)
    want, fragile, pi_all = oracle_lib.call_batch(db, abi.c_params(P), abi.ROW_DTYPE, return_fragile=True,
                                                  return_pi_all=True)
    assert abi.compare_rows(got, want, PI_TOL, P_TOL, fragile, pi_all) == []


def test_triple_alignment_fragments(engine0):
    """Read names with >= 3 included alignments on one locus (re-created fragments,
    smCounter.py:468-479): the sequential replay path."""
    pb, chroms = synth.generate_stress(40, 31, scenarios=("multi", "discord", "snp"))
    P = VcParams(mtDepth=500, rpb=4.0, hpLen=8, minBQ=2, minMQ=0, mismatchThr=100.0)
    db = features.extract_features(pb, P)
    got = engine0.call_batch_host(db, P)
    want, fragile = oracle_lib.call_batch(db, abi.c_params(P), abi.ROW_DTYPE, return_fragile=True)
    assert abi.compare_rows(got, want, PI_TOL, P_TOL, fragile) == []


def test_downsample_flag_and_empty_batch(engine0):
    cfg = synth.CONFIGS["C2"]
    P = VcParams(minBQ=20, minMQ=30, mtDepth=5, rpb=10.0, hpLen=8)       # ds = 10 < 30 barcodes
    db = synth.generate_native(cfg, 0, 16, P)
    got = engine0.call_batch_host(db, P)
    want = oracle_lib.call_batch(db, abi.c_params(P), abi.ROW_DTYPE)
    assert (got["status"] & abi.ST_DOWNSAMPLED).all()
    assert abi.compare_rows(got, want, PI_TOL, P_TOL) == []
    empty = synth.generate_native(cfg, 0, 0, P)
    assert len(engine0.call_batch_host(empty, P)) == 0


def test_bad_ids_are_reported_not_crashed(engine0):
    """Violations of the batch contract flag the locus (SMC_ST_BAD_INPUT), leave the others alone and never
    fault: fragment slot or allele id out of the declared range, barcode read ranges (`umi_start`) that are not
    ascending / do not cover the locus.  (The `umi` plane itself is redundant next to `umi_start` for
    barcode-major reads and is not read by the table kernel.)"""
    cfg = synth.CONFIGS["C2"]
    P = synth.params_for(cfg)
    others = [0, 1, 2, 4, 5, 6, 7]

    def run(mutate):
        db = synth.generate_native(cfg, 0, 8, P)
        mutate(db)
        got = engine0.call_batch_host(db, P)
        assert got["status"][3] & abi.ST_BAD_INPUT
        assert (got["status"][others] == 0).all()
        return db, got

    def bad_frag(db):
        db.frag[db.read_off(3) + 5] = 100000
    def bad_allele(db):
        db.meta[db.read_off(3) + 7] |= 0xff
    def bad_ustart_order(db):
        o = int(db.loci["umi_off"][3])
        db.umi_start[o + 4], db.umi_start[o + 5] = db.umi_start[o + 5], db.umi_start[o + 4]
    def bad_ustart_cover(db):
        o = int(db.loci["umi_off"][3])
        db.umi_start[o + int(db.loci["n_umi"][3])] -= 1
    def bad_ustart_range(db):
        o = int(db.loci["umi_off"][3])
        db.umi_start[o + 2] = 0x7fffffff
    for m in (bad_frag, bad_allele, bad_ustart_order, bad_ustart_cover, bad_ustart_range):
        db, got = run(m)
    with pytest.raises(rows.RowError):
        rows.format_rows(got, db, P, synth.CyclicRef())


def test_pack_words_matches_the_numpy_mirror_and_checks_the_slots(engine0):
    """smc_pack_words (meta + frag planes -> one word per read) against devplanes.pack_words_host, the rows of smc_plan_run_words on those words against smc_plan_run on the planes - and the slot contract,
    which only the packing step can see: a slot that steps by two, a first slot that is not 0, a last slot that is not
    n_frag - 1 each flag their locus (class 31 in the word) and nobody else's."""
    import torch
    from smcounter_amd import devplanes
    cfg = synth.CONFIGS["C2"]
    P = synth.params_for(cfg)
    db = synth.generate_native(cfg, 0, 64, P)
    dev = torch.device("cuda", 0)
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int32)).to(dev)

    def words_of(db):
        plan = engine0.make_plan(db.loci)
        meta, frag, ustart = up(db.meta), up(db.frag), up(db.umi_start)
        words = torch.zeros_like(meta)
        plan.pack_words(meta, frag, words)
        r_words = plan.download(plan.run([words, ustart], P))
        r_planes = plan.download(plan.run([meta, meta, frag, meta, ustart], P))
        plan.close()
        return words.cpu().numpy().view(np.uint32), r_words, r_planes

    w, r_words, r_planes = words_of(db)
    assert np.array_equal(w, devplanes.pack_words_host(db.meta, db.frag, db.loci))
    assert r_words.tobytes() == r_planes.tobytes() and (r_words["status"] == 0).all()
    for case in range(3):
        bad = synth.generate_native(cfg, 0, 64, P)
        o, n = bad.read_off(5), int(bad.loci["n_reads"][5])
        slot = bad.frag[o:o + n] & np.uint32(features.FRAG_SLOT_MASK)
        cls = bad.frag[o:o + n] & ~np.uint32(features.FRAG_SLOT_MASK)
        if case == 0:
            slot[n // 2:] += 1                                   # a step of two in the middle
        elif case == 1:
            slot += 1                                            # first slot 1
        else:
            bad.loci["n_frag"][5] += 1                           # last slot != n_frag - 1
        bad.frag[o:o + n] = cls | slot
        w, r_words, r_planes = words_of(bad)
        assert r_words["status"][5] & abi.ST_BAD_INPUT and r_planes["status"][5] & abi.ST_BAD_INPUT
        assert (np.delete(r_words["status"], 5) == 0).all()
        assert ((w[o:o + n] >> 27) == 31).sum() >= 1
        assert np.array_equal(w, devplanes.pack_words_host(bad.meta, bad.frag, bad.loci))   # the mirror flags the same reads


def test_full_size_properties(engine0):
    """BASELINE.json's C2 at full size (10k loci) through size-independent properties: every locus
    reports DP = depth, UMT = barcodes, sum of per-allele depths <= DP, PI_ref ~ barcodes * PI of a
    clean barcode, and a checksum identical across two different launch shapes (whole batch vs two
    halves)."""
    cfg = synth.CONFIGS["C2"]
    P = synth.params_for(cfg)
    db = synth.generate_native(cfg, 0, cfg.n_loci, P)
    got = engine0.call_batch_host(db, P)
    assert (got["status"] == 0).all()
    assert (got["cvg"] == cfg.depth).all() and (got["all_mt"] == cfg.n_umi).all()
    assert (got["dp"].sum(axis=1) <= got["cvg"]).all()
    assert (got["used_mt"] <= cfg.n_umi).all() and (got["used_mt"] >= cfg.n_umi - 2).all()
    ref_ids = np.array([{"A": 0, "T": 1, "G": 2, "C": 3}[r] for r in db.ref])
    pi_ref = got["pi"][np.arange(len(got)), ref_ids]
    assert pi_ref.min() > 2.0 * cfg.n_umi and pi_ref.max() < 6.0 * cfg.n_umi
    half = cfg.n_loci // 2
    lo = synth.generate_native(cfg, 0, half, P)
    hi = synth.generate_native(cfg, half, cfg.n_loci, P)
    both = np.concatenate([engine0.call_batch_host(lo, P), engine0.call_batch_host(hi, P)])
    assert both.tobytes() == got.tobytes()


def test_planes_built_for_other_parameters_are_refused(engine0):
    """VERDICT r1 (weak 1b): the planes bake minBQ / minMQ / mismatchThr / primerDist in (read class, mismatch flag,
    in-deletion quality); running them under another set used to give silently wrong rows.  The batch now carries a
    fingerprint of those four (smc_locus.flags bits 1-15) and smc_plan_run answers SMC_E_INPUT."""
    import dataclasses
    from smcounter_amd import _lib
    cfg = synth.CONFIGS["C2"]
    P = synth.params_for(cfg)                                        # minBQ 20
    db = synth.generate_native(cfg, 0, 64, P)
    assert (engine0.call_batch_host(db, P)["status"] == 0).all()
    for other in (dataclasses.replace(P, minBQ=25), dataclasses.replace(P, minMQ=20), dataclasses.replace(P, primerDist=3),
                  dataclasses.replace(P, mismatchThr=5.0)):
        with pytest.raises(_lib.SmcError, match="built with other parameters"):
            engine0.call_batch_host(db, other)
        planes = engine0.upload(db)
        plan = engine0.make_plan(db.loci)
        with pytest.raises(_lib.SmcError, match="built with other parameters"):
            plan.run(planes, other)
        plan.close()
    # parameters that are not folded into the planes may change freely
    for same in (dataclasses.replace(P, mtDrop=1), dataclasses.replace(P, mtDepth=500), dataclasses.replace(P, rpb=2.0)):
        got = engine0.call_batch_host(db, same)
        want = oracle_lib.call_batch(db, abi.c_params(same), abi.ROW_DTYPE)
        assert abi.compare_rows(got, want, PI_TOL, P_TOL) == []
    # a batch that states no parameter set, or two, is refused when the plan is made
    for mutate in (lambda L: L["flags"].__setitem__(slice(None), 0), lambda L: L["flags"].__setitem__(5, L["flags"][5] ^ 0x10)):
        bad = dataclasses.replace(db, loci=db.loci.copy())
        mutate(bad.loci)
        with pytest.raises(_lib.SmcError, match="fingerprint"):
            engine0.call_batch_host(bad, P)


def _full_size_parity(engine0, cfg, n_loci, chunk, want_cut_vcf=None):
    """Every locus of [0, n_loci) of `cfg`: GPU rows (resident path) against the CPU restatement on all host cores, chunk
    by chunk (host RAM stays bounded).  Returns the parity report summed over chunks (+ the row strings when asked)."""
    P = synth.params_for(cfg)
    cores = len(os.sched_getaffinity(0))
    tot = {"loci": 0, "mismatches": 0, "fragile_skipped": 0, "near_tie_skipped": 0, "pi_max_abs_diff": 0.0, "detail": []}
    texts = ([], [])
    for lo in range(0, n_loci, chunk):
        db = synth.generate_native(cfg, lo, min(n_loci, lo + chunk), P)
        planes = engine0.upload(db)
        plan = engine0.make_plan(db.loci)
        got = plan.download(plan.run(planes, P))
        plan.close()
        del planes
        want, fragile, pi_all = oracle_lib.call_batch_mt(db, abi.c_params(P), abi.ROW_DTYPE, cores, return_fragile=True,
                                                         return_pi_all=True)
        rep = abi.parity_report(got, want, fragile, pi_all, PI_TOL, P_TOL)
        for k in ("loci", "mismatches", "fragile_skipped", "near_tie_skipped"):
            tot[k] += rep[k]
        tot["pi_max_abs_diff"] = max(tot["pi_max_abs_diff"], rep["pi_max_abs_diff"])
        tot["detail"] += ["[%d+] %s" % (lo, d) for d in rep["detail"]]
        if want_cut_vcf is not None:
            ref = synth.CyclicRef()
            firm = (fragile == 0)
            firm[list(abi.near_tie_loci(got, want, pi_all=pi_all))] = False      # (order of two PI-tied alleles: unpinned)
            for R, acc in ((got, texts[0]), (want, texts[1])):
                t = rows.format_rows(R, db, P, ref)
                acc.extend(x for x, f in zip(t, firm) if f)
    return tot, texts


@pytest.mark.timeout(900)
def test_full_size_C5_every_row_and_cut_vcf(engine0, tmp_path):
    """BASELINE configs[4] at its real size and spike rate on one GPU: 100,000 loci x 8000 reads, 0.5 % AF at 1 % of the
    loci.  EVERY row against the CPU restatement, and the .cut.vcf written from the GPU rows equal to the one written
    from the restatement's rows."""
    from smcounter_amd import postfilter, writers
    cfg = synth.CONFIGS["C5"]
    P = synth.params_for(cfg)
    rep, (t_gpu, t_cpu) = _full_size_parity(engine0, cfg, cfg.n_loci, 10000, want_cut_vcf=True)
    print("C5 full-size parity:", {k: v for k, v in rep.items() if k != "detail"})
    assert rep["loci"] == 100000 and rep["mismatches"] == 0, rep["detail"]
    assert rep["fragile_skipped"] + rep["near_tie_skipped"] <= 0.01 * rep["loci"]
    assert rep["pi_max_abs_diff"] <= PI_TOL
    thr = writers.pi_threshold(P.mtDepth, 0)
    bodies = []
    for tag, text in (("gpu", t_gpu), ("cpu", t_cpu)):
        prefix = str(tmp_path / tag)
        writers.write_outputs(prefix, postfilter.apply_repeat_filters(text, {}, {}), thr)
        bodies.append([l for l in open(prefix + ".smCounter.cut.vcf") if not l.startswith("#")])   # (header lines name the prefix)
    assert bodies[0] == bodies[1]
    n_calls = sum(1 for l in bodies[0] if l and not l.startswith("#"))
    assert n_calls >= 5      # 1000 spiked loci, 0.5 % AF of 133 barcodes = 0.67 alt barcodes on average: few reach PI 16
    print("C5 .cut.vcf: %d called variants, threshold %d" % (n_calls, thr))


@pytest.mark.timeout(900)
def test_full_size_C4_one_rank_share(engine0):
    """BASELINE configs[3] (1 M loci x 3000x over 8 GPUs): one rank's share, the first 125,000 loci of C4's seed, every
    row against the CPU restatement.  (The other seven shares are the same code on other loci: dist.shard_range.)"""
    cfg = synth.CONFIGS["C4"]
    rep, _ = _full_size_parity(engine0, cfg, 125000, 25000)
    print("C4 share parity:", {k: v for k, v in rep.items() if k != "detail"})
    assert rep["loci"] == 125000 and rep["mismatches"] == 0, rep["detail"]
    assert rep["fragile_skipped"] + rep["near_tie_skipped"] <= 0.01 * rep["loci"]
    assert rep["pi_max_abs_diff"] <= PI_TOL


@pytest.mark.gpu
def test_host_sampling_marks_are_checked(engine0):
    """SMC_LF_SAMPLED loci (reference down-sampling applied by the host): the kernel keeps exactly the marked
    subset (golden rows, above) and flags marks that do not keep min(#keys, ds) barcodes."""
    import dataclasses
    path = [p for p in golden_files() if "stress_ds1" in p][0]
    pb, db, P, refp, expected = load_golden(path)
    sampled = np.nonzero(db.loci["flags"] & 1)[0]
    assert len(sampled) > 10
    got = engine0.call_batch_host(db, P)
    assert (got["status"][sampled] & abi.ST_DOWNSAMPLED).all() and (got["used_mt"][sampled] == P.ds).all()
    bad = dataclasses.replace(db, umi_start=db.umi_start.copy())
    l = int(sampled[0])
    o = int(db.loci["umi_off"][l])
    k = int(np.nonzero(bad.umi_start[o:o + int(db.loci["n_umi"][l])] >> 31)[0][0])
    bad.umi_start[o + k] &= 0x7fffffff                   # one key too many
    got = engine0.call_batch_host(bad, P)
    assert got["status"][l] & abi.ST_BAD_INPUT
    assert not (np.delete(got["status"], l) & abi.ST_BAD_INPUT).any()


def test_spiked_variants_cut_vcf_concordance(engine0, tmp_path):
    """BASELINE configs[4] in small: spiked-in low-AF variants at 8000x depth (C5's shape, spike rate raised so a
    slice holds dozens); the .cut.vcf written from the GPU rows equals the one written from the CPU restatement's
    rows, and it holds called variants."""
    import dataclasses
    from smcounter_amd import postfilter, writers
    cfg = dataclasses.replace(synth.CONFIGS["C5"], name="C5s", alt_locus_frac=0.15, alt_af=0.03)
    P = synth.params_for(cfg)
    db = synth.generate_native(cfg, 0, 400, P)
    got = engine0.call_batch_host(db, P)
    want, fragile, pi_all = oracle_lib.call_batch(db, abi.c_params(P), abi.ROW_DTYPE, return_fragile=True,
                                                  return_pi_all=True)
    assert abi.compare_rows(got, want, PI_TOL, P_TOL, fragile, pi_all) == []
    ref = synth.CyclicRef()
    thr = writers.pi_threshold(P.mtDepth, 0)
    files = []
    for tag, R in (("gpu", got), ("cpu", want)):
        text = postfilter.apply_repeat_filters(rows.format_rows(R, db, P, ref), {}, {})
        prefix = str(tmp_path / tag)
        writers.write_outputs(prefix, text, thr)
        body = [l for l in open(prefix + ".smCounter.cut.vcf") if not l.startswith("#")]
        files.append(body)
    keep = [i for i in range(len(files[0]))]
    assert len(files[0]) == len(files[1]) >= 20
    # (loci with a rounding-decided barcode may differ by one in a count column; none is expected here)
    assert (fragile > 0).sum() == 0 and files[0] == files[1]


def test_pack_kernel_matches_the_numpy_mirror_and_unpacks_to_the_same_strings(engine0):
    """k_pack_rows (the 168-byte wire rows the multi-GPU gather moves) against abi.pack_wire byte for byte, on rows with
    filters / bi-allelic loci / zero coverage (golden stress vectors) and on a C3 slice; unpacked, they print the same
    strings as the full rows."""
    path = [p for p in golden_files() if "stress2" in p][0]
    pb, db, P, refp, expected = load_golden(path)
    for db_, P_, ref_ in ((db, P, refp), (synth.generate_native(synth.CONFIGS["C3"], 0, 700, synth.params_for(synth.CONFIGS["C3"])),
                                          synth.params_for(synth.CONFIGS["C3"]), synth.CyclicRef())):
        planes = engine0.upload(db_)
        plan = engine0.make_plan(db_.loci)
        rows_d = plan.run(planes, P_)
        wire = plan.download_wire(plan.pack(rows_d))
        full = plan.download(rows_d)
        assert wire.tobytes() == abi.pack_wire(full).tobytes()
        assert rows.format_rows(abi.unpack_wire(wire), db_, P_, ref_) == rows.format_rows(full, db_, P_, ref_)
        plan.close()


def test_descriptors_outside_the_buffers_are_refused(engine0):
    from smcounter_amd import _lib
    cfg = synth.CONFIGS["C2"]
    P = synth.params_for(cfg)
    db = synth.generate_native(cfg, 0, 4, P)
    for field, val in (("read_off4", 10 ** 6), ("umi_off", 10 ** 6), ("n_reads", -1)):
        bad = synth.generate_native(cfg, 0, 4, P)
        bad.loci[field][2] = val
        with pytest.raises(_lib.SmcError, match="points outside|layout contract"):
            engine0.call_batch_host(bad, P)
    assert (engine0.call_batch_host(db, P)["status"] == 0).all()


def test_integration_md_stub_runs_as_written(engine0):
    """The ctypes stub INTEGRATION.md shows for smCounter.py is executed (library path pointed at the in-tree build):
    same strings as the package's own path."""
    import re, types
    from conftest import ROOT
    from smcounter_amd import _lib
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    code = re.search(r"```python\n(.*?)```", text, re.S).group(1)
    code = code.replace('ctypes.CDLL("libsmcounter_hip.so")', "ctypes.CDLL(%r)" % _lib.LIB_PATH)
    code = "\n".join(l for l in code.split("\n") if not l.startswith("output = vc_batch_gpu("))
    ns = {}
    exec(compile(code, "INTEGRATION.md", "exec"), ns)
    path = [p for p in golden_files() if "stress2" in p][0]
    pb, db, P, refp, expected = load_golden(path)
    args = types.SimpleNamespace(minBQ=P.minBQ, minMQ=P.minMQ, mtDepth=P.mtDepth, rpb=P.rpb, hpLen=P.hpLen,
                                 mismatchThr=P.mismatchThr, mtDrop=P.mtDrop, maxMT=P.maxMT, primerDist=P.primerDist)
    got = ns["vc_batch_gpu"](pb, args, refp)
    want = rows.format_rows(engine0.call_batch_host(db, P), db, P, refp)
    assert got == want and len(got) == pb.n_loci
