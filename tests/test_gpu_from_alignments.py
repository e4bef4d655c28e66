"""Loci WITH a variant through the whole device path: synthetic ALIGNMENTS with variants under the reads
(smc_synth_alignments' alt_locus_frac / alt_af) -> smc_build_planes -> smc_plan_create_dev -> smc_plan_run_words, every row
against the same alignments taken through oracle/aln_planes.c (the reference's per-pileup-read logic, smCounter.py:316-471) +
oracle/smc_oracle.c (:26-98, :182-269, :482-600).  Integer columns and FILTER bits bit-exact, PI and Fisher p within 1e-6."""
import os

import numpy as np
import pytest

from smcounter_amd import abi, synth

import oracle_lib

pytestmark = pytest.mark.gpu
PI_TOL = 1e-6
P_TOL = 1e-6


@pytest.fixture(autouse=True)
def _experiment_switches(monkeypatch):
    """The environment switches these tests flip (chunk geometries, poisoned scratch, forced code paths) are experiment knobs: the
    libraries read them only under SMC_EXPERIMENTAL."""
    monkeypatch.setenv("SMC_EXPERIMENTAL", "1")


def _gpu_rows(engine0, A, P):
    """The product path on one run of alignments: words built on the device, plan made on the device, rows back."""
    from smcounter_amd import devplanes
    rb = devplanes.resident_from_alignments(A, engine0, P, all_planes=False)
    d_loci = devplanes.DevLoci(engine0, rb.loci)
    plan = engine0.make_plan_dev(d_loci, rb.n_loci)
    got = plan.run_devbuf([rb.words, rb.planes[4]], P).copy()
    plan.close()
    d_loci.free()
    return rb, got


def oracle_rows(A, P, chunk, cores):
    """-> (rows, fragile, pi_all, alleles per locus) of every locus of the run, the oracle's way, chunk by chunk."""
    want, fragile, pi_all, n_al = [], [], [], []
    for c0 in range(0, A["nl"], chunk):
        c1 = min(A["nl"], c0 + chunk)
        db = oracle_lib.aln_planes(A, P, c0, c1, n_threads=cores)
        w, f, p = oracle_lib.call_batch_mt(db, abi.c_params(P), abi.ROW_DTYPE, cores, return_fragile=True, return_pi_all=True)
        want.append(w); fragile.append(f); pi_all.append(p); n_al.append(db.loci["n_alleles"].copy())
    return np.concatenate(want), np.concatenate(fragile), np.concatenate(pi_all), np.concatenate(n_al)


@pytest.mark.timeout(900)
def test_x3_shaped_loci_from_alignments_reach_the_filters(engine0):
    """24,000 loci of C3's depth shape with a 10 % variant at 30 % of the positions: a third of the rows goes through
    filterVariants and its Fisher tests - from alignments, not from pre-built read words."""
    cfg = synth.CONFIGS["X3"]
    P = synth.params_for(cfg)
    cores = len(os.sched_getaffinity(0))
    A = synth.generate_alignments(cfg, 24000, P)
    rb, got = _gpu_rows(engine0, A, P)
    want, fragile, pi_all, n_al = oracle_rows(A, P, 8000, cores)
    assert (rb.loci["n_alleles"] == n_al).all()
    rep = abi.parity_report(got, want, fragile, pi_all, PI_TOL, P_TOL)
    print("X3 from alignments:", {k: v for k, v in rep.items() if k != "detail"})
    assert rep["loci"] == 24000 and rep["mismatches"] == 0, rep["detail"]
    assert rep["loci_filtered"] >= 5000 and rep["fisher_tests_run"] >= 3 * rep["loci_filtered"]
    assert rep["pi_max_abs_diff"] <= PI_TOL and rep["p_max_abs_diff"] <= P_TOL
    assert rep["fragile_skipped"] + rep["near_tie_skipped"] <= 0.01 * rep["loci"]
    # FILTER bits of every candidate the two agree on having
    both = (got["cand"]["flt_applied"] != 0) & (want["cand"]["flt_applied"] != 0)
    firm = np.ones(len(got), bool)
    firm[list(abi.near_tie_loci(got, want, pi_all=pi_all))] = False
    firm &= fragile == 0
    assert (got["cand"]["flt"][both & firm[:, None]] == want["cand"]["flt"][both & firm[:, None]]).all()


@pytest.mark.timeout(1500)
def test_C5_from_alignments_every_row_and_cut_vcf(engine0, tmp_path):
    """BASELINE configs[4] from ALIGNMENTS at its real size on one GPU: 100,000 loci at ~ 8000x (133 barcodes x 60 reads), a
    0.5 % AF spike at 1 % of the positions.  Every row against the oracle, and the .cut.vcf written from the GPU rows equal to
    the one written from the oracle's rows."""
    from smcounter_amd import postfilter, rows, writers
    cfg = synth.CONFIGS["C5"]
    P = synth.params_for(cfg)
    cores = len(os.sched_getaffinity(0))
    A = synth.generate_alignments(cfg, cfg.n_loci, P)
    rb, got = _gpu_rows(engine0, A, P)
    want, fragile, pi_all, n_al = oracle_rows(A, P, 6000, cores)
    assert (rb.loci["n_alleles"] == n_al).all()
    rep = abi.parity_report(got, want, fragile, pi_all, PI_TOL, P_TOL)
    print("C5 from alignments:", {k: v for k, v in rep.items() if k != "detail"})
    assert rep["loci"] == cfg.n_loci and rep["mismatches"] == 0, rep["detail"]
    assert rep["loci_filtered"] >= 100 and rep["fisher_tests_run"] > 0
    assert rep["pi_max_abs_diff"] <= PI_TOL and rep["p_max_abs_diff"] <= P_TOL
    firm = fragile == 0
    firm[list(abi.near_tie_loci(got, want, pi_all=pi_all))] = False       # (order of two PI-tied alleles: unpinned)
    ref = synth.CyclicRef()
    thr = writers.pi_threshold(P.mtDepth, 0)
    bodies = []
    for tag, R in (("gpu", got), ("cpu", want)):
        text = [t for t, f in zip(rows.format_rows(R, rb, P, ref), firm) if f]
        prefix = str(tmp_path / tag)
        writers.write_outputs(prefix, postfilter.apply_repeat_filters(text, {}, {}), thr)
        bodies.append([l for l in open(prefix + ".smCounter.cut.vcf") if not l.startswith("#")])
    assert bodies[0] == bodies[1]
    print("C5 from alignments .cut.vcf: %d called variants, threshold %d" % (len(bodies[0]), thr))
    assert len(bodies[0]) >= 5


def test_example_depth_from_alignments(engine0):
    """EX's statistics (the reference's own example run: ~ 58 k reads and ~ 4,900 barcodes per locus, a 10 % variant at 30 % of
    the positions) from alignments: the deep class's parts and chunks behind the device builder, the Fisher tests on supports of
    thousands."""
    cfg = synth.CONFIGS["EX"]
    P = synth.params_for(cfg)
    cores = len(os.sched_getaffinity(0))
    A = synth.generate_alignments(cfg, 400, P)
    rb, got = _gpu_rows(engine0, A, P)
    want, fragile, pi_all, n_al = oracle_rows(A, P, 200, cores)
    assert (rb.loci["n_alleles"] == n_al).all() and int(rb.loci["n_reads"].min()) > 50000
    rep = abi.parity_report(got, want, fragile, pi_all, PI_TOL, P_TOL)
    print("EX from alignments:", {k: v for k, v in rep.items() if k != "detail"})
    assert rep["mismatches"] == 0, rep["detail"]
    assert rep["loci_filtered"] >= 80 and rep["fisher_tests_run"] >= 3 * rep["loci_filtered"]
    assert rep["pi_max_abs_diff"] <= PI_TOL and rep["p_max_abs_diff"] <= P_TOL


def test_large_blocks_and_the_write_pattern_probe(engine0):
    """smc_mem_alloc_best picks one of several candidate blocks by the walk's write pattern: it must behave like any device
    block - copies in and out, kernels writing it, freeing.  (Until the end of round 5 blocks of 256 MB and more were virtual-memory
    ranges over 64 MB handles: see test_a_fresh_large_block_keeps_what_is_written_to_it.)"""
    import ctypes
    from smcounter_amd import _lib
    from smcounter_amd.engine import DevBuf
    L = engine0.L
    n = (300 << 20) // 4
    rng = np.random.default_rng(5)
    src = rng.integers(0, 1 << 32, size=n, dtype=np.uint32)
    plain = DevBuf(engine0, 4 * n)
    tuned = DevBuf(engine0, 4 * n, walk_output=True)
    assert tuned.tuned and engine0.alloc_log and engine0.alloc_log[-1]["candidates"] >= 1
    for b in (plain, tuned):
        b.upload(src)
        assert (b.download(np.uint32, n) == src).all()
        b.upload(src[::-1].copy(), 0)
        assert (b.download(np.uint32, 1 << 20, 4 * (n - (1 << 20))) == src[::-1][n - (1 << 20):]).all()
    ms = ctypes.c_float()
    _lib.check(L.smc_mem_write_probe(engine0.ctx, ctypes.c_void_p(tuned.data_ptr()), 4 * n, ctypes.byref(ms)), "smc_mem_write_probe")
    assert 0.0 < ms.value < 1000.0
    plain.free(); tuned.free()
    engine0.trim()
    again = DevBuf(engine0, 4 * n, walk_output=True)              # (after the trim: a fresh choice)
    again.upload(src)
    assert (again.download(np.uint32, n) == src).all()
    again.free()


def test_a_fresh_large_block_keeps_what_is_written_to_it(engine0):
    """Blocks of hundreds of megabytes allocated, written, read back after a while with kernels running in between, freed, again:
    with the virtual-memory backing smc_mem_alloc had for such blocks (hipMemCreate + hipMemMap) every second block read back as
    zeros some tens of milliseconds after the write on this ROCm - the wipe of the pages an earlier release gave back lands on
    the new owner (scripts/vmm_stress.py: 123-159 bad read-backs in 40 blocks; in the product a batch's umi_start came back
    zero).  The library allocates with hipMalloc again; this is that script's loop on what smc_mem_alloc gives now."""
    import ctypes, time
    from smcounter_amd import _lib
    L, ctx = engine0.L, engine0.ctx
    pat = (np.arange(1 << 18, dtype=np.uint32) * np.uint32(2654435761) + np.uint32(12345)).astype(np.uint32)
    mb = 576
    for it in range(6):
        p, other = ctypes.c_void_p(), ctypes.c_void_p()
        _lib.check(L.smc_mem_alloc(ctx, mb << 20, ctypes.byref(p)), "smc_mem_alloc")
        offs = [0, (mb << 20) // 2, (mb << 20) - pat.nbytes]
        for o in offs:
            _lib.check(L.smc_mem_h2d(ctx, p.value + o, pat.ctypes.data, pat.nbytes), "h2d")
        _lib.check(L.smc_mem_alloc(ctx, 300 << 20, ctypes.byref(other)), "smc_mem_alloc")
        ms = ctypes.c_float()
        for delay in (0.0, 0.02, 0.06):
            time.sleep(delay)
            _lib.check(L.smc_mem_write_probe(ctx, other, 300 << 20, ctypes.byref(ms)), "probe")
            for o in offs:
                got = np.empty_like(pat)
                _lib.check(L.smc_mem_d2h(ctx, got.ctypes.data, p.value + o, got.nbytes), "d2h")
                assert (got == pat).all(), (it, o, delay, int((got != pat).sum()))
        L.smc_mem_free(ctx, other)
        L.smc_mem_free(ctx, p)


@pytest.mark.parametrize("poison", [0xA5, 0xFF, 0x5A])
def test_nothing_reads_scratch_this_run_has_not_written(engine0, monkeypatch, poison):
    """The plane builder's scratch is recycled from run to run (a fresh allocation is zero, a recycled one is not).  Round 5's soak
    (scripts/fa_soak.py) found a read with bq = minBQ - 1 at the first position of an alignment with a deletion INCLUDED when the
    byte next to its quality - a pair of the second pool no copy covers - held 0x8D or more: the byte-lane add carried.  Here
    the scratch is filled with a pattern before every run (SMC_BP_POISON_SCRATCH) on shapes either side of the sort thresholds,
    with many general CIGARs and minBQ one above a quality that occurs: every row must still equal the oracle's.  The blocks the
    device-made launch plan takes (lists, the deep class's descriptors) get the same treatment (SMC_PLAN_POISON)."""
    import dataclasses
    from smcounter_amd.params import VcParams
    monkeypatch.setenv("SMC_BP_POISON_SCRATCH", str(poison))
    monkeypatch.setenv("SMC_PLAN_POISON", str(poison ^ 0x7E))
    cores = len(os.sched_getaffinity(0))
    for n_umi, rpb, nl, pdel, pins in ((73, 60, 420, 0.05, 0.0), (1500, 15, 100, 0.0, 0.2), (30, 10, 700, 0.1, 0.1)):
        cfg = synth.SynthConfig("P", nl, n_umi, rpb, 4242 + n_umi, p_overlap=0.9, alt_locus_frac=0.5, alt_af=0.1)
        P = VcParams(minBQ=13, minMQ=0, mtDepth=n_umi, rpb=float(rpb), hpLen=8, mismatchThr=100.0, mtDrop=0, maxMT=0, primerDist=20)
        A = synth.generate_alignments(cfg, nl, P, p_del_aln=pdel, p_ins_aln=pins, p_clip=0.0)
        rb, got = _gpu_rows(engine0, A, P)
        want, fragile, pi_all, n_al = oracle_rows(A, P, nl, cores)
        assert (rb.loci["n_alleles"] == n_al).all()
        assert abi.compare_rows(got, want, PI_TOL, P_TOL, fragile, pi_all) == []


def _with_word_bits(eng, bits, fn):
    old = eng.word_bits
    eng.word_bits = bits
    try:
        return fn()
    finally:
        eng.word_bits = old


def test_the_16_bit_read_words_are_the_32_bit_ones(engine0):
    """smc_build_planes_w16 against smc_build_planes on the same run (general CIGARs, insertions, deletions, clips, a variant
    under the reads, a locus window either side of the sort thresholds): every 16-bit word is smc_read_word16 of the 32-bit one,
    umi_start and the descriptors are the same bytes, and so are the rows of smc_plan_run_words16 and smc_plan_run_words."""
    from smcounter_amd import devplanes
    from smcounter_amd.params import VcParams
    for n_umi, rpb, nl in ((60, 40, 700), (900, 12, 150)):
        cfg = synth.SynthConfig("W", nl, n_umi, rpb, 977 + n_umi, p_overlap=0.8, alt_locus_frac=0.4, alt_af=0.2)
        P = VcParams(minBQ=20, minMQ=0, mtDepth=n_umi, rpb=float(rpb), hpLen=8, mismatchThr=100.0, mtDrop=0, maxMT=0, primerDist=20)
        A = synth.generate_alignments(cfg, nl, P, p_del_aln=0.08, p_ins_aln=0.05, p_clip=0.2)
        out = {}
        for bits in (32, 16):
            rb, rows = _with_word_bits(engine0, bits, lambda: _gpu_rows(engine0, A, P))
            assert rb.words.word_bits == bits
            out[bits] = (rb.words.download(np.uint16 if bits == 16 else np.uint32, rb.n_slots), rb.planes[4].download(np.uint32, rb.n_ustart),
                         rb.loci.copy(), rows)
        w32, w16 = out[32][0], out[16][0]
        assert (devplanes.words16_from_32(w32) == w16).all()
        real = w16 != 0
        assert (devplanes.words32_from_16(w16)[real] == w32[real]).all() and not w32[~real].any()
        assert out[32][2].tobytes() == out[16][2].tobytes()
        for L in out[32][2]:                     # (the entries between two loci's ranges are nobody's)
            o, nu = int(L["umi_off"]), int(L["n_umi"])
            assert (out[32][1][o:o + nu + 1] == out[16][1][o:o + nu + 1]).all()
        assert out[32][3].tobytes() == out[16][3].tobytes()


def test_a_run_without_room_in_16_bit_words_is_built_with_32(engine0):
    """A base quality beyond 63 under a covered locus: smc_build_planes_w16 reports it (status bit 32), the caller builds the run
    with 32-bit words, the rows are the oracle's."""
    from smcounter_amd import devplanes
    from smcounter_amd.params import VcParams
    cores = len(os.sched_getaffinity(0))
    P = VcParams(minBQ=20, minMQ=0, mtDepth=80, rpb=30.0, hpLen=8, mismatchThr=100.0, mtDrop=0, maxMT=0, primerDist=20)
    cfg = synth.SynthConfig("N", 300, 80, 30, 4711, p_overlap=0.8, alt_locus_frac=0.3, alt_af=0.2)
    # (a) one quality of 70 somewhere in the middle of the pool
    A = synth.generate_alignments(cfg, 300, P, p_del_aln=0.05, p_ins_aln=0.0, p_clip=0.0)
    a = A["aln"][len(A["aln"]) // 2]
    A["bq"][2 * (int(a["seq_off"]) + int(a["l_seq"]) // 2) + 1] = 70
    rb, got = _with_word_bits(engine0, 16, lambda: _gpu_rows(engine0, A, P))
    assert rb.words.word_bits == 32
    want, fragile, pi_all, n_al = oracle_rows(A, P, 300, cores)
    assert abi.compare_rows(got, want, PI_TOL, P_TOL, fragile, pi_all) == []
    # (more than sixteen alleles at a locus: tests/test_bam_golden.py's bam_deep case, through the decoder - its batch is built again)
    # and a run that does fit stays in 16 bits
    A = synth.generate_alignments(cfg, 300, P, p_del_aln=0.05, p_ins_aln=0.0, p_clip=0.0)
    rb, got = _with_word_bits(engine0, 16, lambda: _gpu_rows(engine0, A, P))
    assert rb.words.word_bits == 16
    # (b) a quality of exactly 63 - the largest the 16-bit word holds - is NOT "beyond 63" (the walk's byte-lane test
    # looked at quality + 1 and sent every file with a Q63 to the 32-bit words; the exact path and the host's words16_from_32 took it)
    a = A["aln"][len(A["aln"]) // 3]
    A["bq"][2 * (int(a["seq_off"]) + int(a["l_seq"]) // 2) + 1] = 63
    rb, got = _with_word_bits(engine0, 16, lambda: _gpu_rows(engine0, A, P))
    assert rb.words.word_bits == 16
    want, fragile, pi_all, n_al = oracle_rows(A, P, 300, cores)
    assert abi.compare_rows(got, want, PI_TOL, P_TOL, fragile, pi_all) == []
    # ... and 64 is
    A["bq"][2 * (int(a["seq_off"]) + int(a["l_seq"]) // 2) + 1] = 64
    rb, got = _with_word_bits(engine0, 16, lambda: _gpu_rows(engine0, A, P))
    assert rb.words.word_bits == 32


def test_filter_tallies_by_their_own_kernel_give_the_same_rows(engine0, monkeypatch):
    """The eight tallies only filterVariants reads are taken by the locus's own workgroup at the tail of its chain - or, in a batch
    with loci of the deep class, by k_filter_tallies over the whole chip afterwards (wavefront-items of 4,096-read slices, integer
    adds into the rows): the same bytes either way, on 3,000-read loci (SMC_TALLIES_LATER forces the kernel) and on loci of the
    deep class (where it is the default; SMC_TALLIES_LATER=0 forces the workgroup's own pass)."""
    for name, nl in (("X3", 3000), ("EX", 120)):
        cfg = synth.CONFIGS[name]
        P = synth.params_for(cfg)
        A = synth.generate_alignments(cfg, nl, P)
        out = {}
        for mode in ("0", "1"):
            monkeypatch.setenv("SMC_TALLIES_LATER", mode)
            rb, got = _gpu_rows(engine0, A, P)
            out[mode] = got
        monkeypatch.delenv("SMC_TALLIES_LATER")
        rb, dflt = _gpu_rows(engine0, A, P)
        assert out["0"].tobytes() == out["1"].tobytes() == dflt.tobytes()
        flt = (dflt["cand"]["flt_applied"] != 0).any(axis=1)
        assert int(flt.sum()) >= nl // 8 and int(dflt["ref_tal"][flt][:, 1:9].sum()) > 0


def test_plans_without_the_host_on_loci_of_the_deep_class(engine0):
    """smc_plan_create_dev_spec with loci of the deep class (several workgroups per locus, parts cut on the device by
    k_plan_deep_spec from the record k_plan_classify left, the accumulators and flag scratch sized from the last plan's record): the
    rows of a second and third run of example-depth loci - made without the host - are byte for byte those of the plan made the
    exact way, the device counted no misfit, and a run twice as long fits the scaled sizes."""
    from smcounter_amd import devplanes
    from smcounter_amd.engine import Engine
    eng = Engine(0)
    try:
        cfg = synth.CONFIGS["EX"]
        P = synth.params_for(cfg)
        outs = []
        for n in (60, 60, 120):
            A = synth.generate_alignments(cfg, n, P)
            rb = devplanes.resident_from_alignments(A, eng, P, all_planes=False)
            d_loci = devplanes.DevLoci(eng, rb.loci)
            exact = eng.make_plan_dev(d_loci, rb.n_loci)
            want = exact.run_devbuf([rb.words, rb.planes[4]], P).copy()
            exact.close()
            plan = eng.make_plan_dev(d_loci, rb.n_loci, spec_params=P)
            got = plan.run_devbuf([rb.words, rb.planes[4]], P).copy()
            ok = plan.ok()
            plan.close()
            d_loci.free()
            outs.append((ok, got.tobytes() == want.tobytes(), int(want["cvg"].max())))
        assert outs[0][:2] == (True, True) and outs[1][:2] == (True, True) and outs[2][:2] == (True, True), outs
        assert outs[0][2] > 24576                                   # (loci of the deep class were in it)
        made, exact_n, misfit = eng.spec_counts()
        assert (made, exact_n, misfit) == (2, 1, 0)
    finally:
        eng.close()
