"""The device plane builder (csrc/k_build_planes.inc, through smc_bam_alignments + smc_build_planes) against the host builder
(smc_bam_planes): the same descriptors, barcodes, fragments, reads, sampling marks and allele tables,
on random BAMs (CIGARs with S/M/I/D/N, two chromosomes, odd read names), the variant fixture, and under a barcode cap that
triggers the reference's down-sampling; then the whole command line with planes built either way."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT
from smcounter_amd import abi, bamio, fasta, features
from smcounter_amd.params import VcParams

sys.path.insert(0, os.path.join(ROOT, "tests"))
pytestmark = pytest.mark.gpu


def _same_batch(rb, hb):
    """ResidentBatch (device-built, copied back) vs DeviceBatch (host-built): the same descriptors, the same barcodes with the
    same fragments and the same reads in the same order within a fragment, the same sampling marks and allele tables.  The two
    builders number the barcodes (and a barcode's fragments) of a locus differently - first appearance in the run vs at the
    locus, both within the layout contract - so the comparison is on the order-free fingerprint of planecheck.py."""
    from smcounter_amd import devplanes, planecheck
    db = rb.to_host()
    assert planecheck.differences(db, hb) == []
    # the read words (what the locus kernels load) say what the raw-field planes of the same build say
    words = rb.words.download(np.uint32, rb.n_slots)
    assert np.array_equal(words, devplanes.pack_words_host(db.meta, db.frag, db.loci))
    for l in range(db.n_loci):
        assert int(db.loci["umi_off"][l]) + int(db.loci["n_umi"][l]) + 1 <= len(db.umi_start)


@pytest.mark.parametrize("mt_depth", [1000, 4])          # 4 -> ds = 8 < 25 barcodes: the py2 down-sampling marks
@pytest.mark.parametrize("max_reads", [5000, 2_000_000])
def test_device_built_planes_equal_host_built(engine0, tmp_path, max_reads, mt_depth):
    import test_bamio
    from smcounter_amd import devplanes
    bam, fa_path, loci = test_bamio._random_bam(tmp_path, 23, True)
    fa = fasta.FastaFile(fa_path)
    P = VcParams(mtDepth=mt_depth, rpb=3.0, hpLen=8, minBQ=15, minMQ=20, mismatchThr=8.0)
    host = list(bamio.iter_device_batches_native(bam, fa, loci, P, max_reads=max_reads, nthreads=3))
    dev = list(devplanes.iter_resident_batches(bam, fa, loci, P, engine0, max_reads=max_reads, nthreads=3))
    assert len(host) == len(dev) >= 1
    n_dev = 0
    for (f1, hb), (f2, rb) in zip(host, dev):
        assert f1 == f2
        _same_batch(rb, hb)
        n_dev += rb.n_device_runs
    assert n_dev > 0                                        # (the device path really ran)
    if mt_depth == 4:
        assert any((hb.loci["flags"] & features.LF_SAMPLED).any() for _, hb in host)
        assert any((np.concatenate([hb.umi_start for _, hb in host]) >> 31).any() for _ in (0,))


def test_device_planes_on_the_variant_fixture_and_rows(engine0, tmp_path):
    """bam_fixture (planted variant, indels, soft clips): planes equal, and the rows of the device-built batch equal the
    rows of the host-built one through the same kernels."""
    import bam_fixture
    from smcounter_amd import devplanes, rows, vc
    case = bam_fixture.make_case(str(tmp_path))
    fa = fasta.FastaFile(case["fasta"])
    from smcounter_amd import bedops
    loci = bedops.expand_loci(case["bed"])
    P = VcParams(mtDepth=12, rpb=3.0, hpLen=8)
    host = list(bamio.iter_device_batches_native(case["bam"], fa, loci, P))
    dev = list(devplanes.iter_resident_batches(case["bam"], fa, loci, P, engine0))
    for (_, hb), (_, rb) in zip(host, dev):
        _same_batch(rb, hb)
        assert rb.n_device_runs >= 1
        got = vc.vc_resident(rb, P, fa, engine0)
        want = rows.format_rows(engine0.call_batch_host(hb, P), hb, P, fa)
        assert got == want
    assert any(len(t) > 6 for _, hb in host for t in hb.alleles)      # indel alleles went through the extras list


def test_alignments_flagged_neither_read1_nor_read2(engine0, tmp_path):
    """An alignment with neither READ1 nor READ2: the reference's pairOrder is then whatever the previous pileup read left
    (smCounter.py:359-362).  The device builder takes such a run itself (the walk's exact path looks the previous covering
    alignment up); the batch equals the host builder's.  A pileup that BEGINS with such an alignment - the reference fails
    there - goes back to the host builder, which raises."""
    from smcounter_amd import devplanes
    from smcounter_amd.features import PileupError
    ref = "ACGTTGCAAC" * 30
    fa_path = str(tmp_path / "d.fa")
    open(fa_path, "w").write(">chrD\n" + ref + "\n")

    def make(unflagged):
        recs = [dict(tid=0, pos=10 + (i % 7), qname="r%d:x:BC%d:y" % (i // 2, i % 5), flag=(0x41 if i % 2 == 0 else 0x91) if i not in unflagged else 0x10,
                     mapq=60, cigar=[(0, 40)], seq=ref[10 + (i % 7):50 + (i % 7)], qual=[30] * 40, nm=0) for i in range(40)]
        recs.sort(key=lambda r: r["pos"])
        bam = str(tmp_path / ("d%d.bam" % len(unflagged)))
        bamio.write_bam(bam, [("chrD", len(ref))], recs)
        bamio.write_bai(bam)
        return bam
    fa = fasta.FastaFile(fa_path)
    loci = [("chrD", str(p)) for p in range(15, 45)]
    P = VcParams(mtDepth=100, rpb=2.0)
    bam = make({9, 16, 17, 30})
    host = list(bamio.iter_device_batches_native(bam, fa, loci, P))
    dev = list(devplanes.iter_resident_batches(bam, fa, loci, P, engine0))
    for (_, hb), (_, rb) in zip(host, dev):
        _same_batch(rb, hb)
        assert rb.n_device_runs >= 1 and rb.n_host_runs == 0
    # the first record of the file unflagged: no pairOrder to carry over
    bam = make({0})
    with pytest.raises(PileupError):
        list(devplanes.iter_resident_batches(bam, fa, loci, P, engine0))


def test_a_locus_with_more_than_64_alleles_is_refused(engine0, tmp_path):
    """70 different insertions behind one position: beyond the 64 allele ids a locus's table holds (SMC_MAX_ALLELES).  The device
    builder flags the run (status 8), the host builder it falls back to says which locus - no batch is made of it."""
    from smcounter_amd import devplanes
    rng = np.random.default_rng(64)
    ref = "".join(rng.choice(list("ACGT"), size=300))
    fa_path = str(tmp_path / "a.fa")
    open(fa_path, "w").write(">chrA\n" + ref + "\n")
    recs = []
    seen = set()
    while len(seen) < 70:
        ins = "".join(rng.choice(list("ACGT"), size=int(rng.integers(1, 5))))
        if ins in seen:
            continue
        seen.add(ins)
        i = len(seen)
        recs.append(dict(tid=0, pos=100, qname="r%d:x:BC%02d:y" % (i, i % 9), flag=0x41, mapq=60, cigar=[(0, 30), (1, len(ins)), (0, 20)],
                         seq=ref[100:130] + ins + ref[130:150], qual=[30] * (50 + len(ins)), nm=len(ins)))
    bam = str(tmp_path / "a.bam")
    bamio.write_bam(bam, [("chrA", len(ref))], recs)
    bamio.write_bai(bam)
    fa = fasta.FastaFile(fa_path)
    loci = [("chrA", str(p)) for p in range(120, 140)]
    P = VcParams(mtDepth=100, rpb=2.0)
    with pytest.raises(Exception) as e:
        list(devplanes.iter_resident_batches(bam, fa, loci, P, engine0))
    assert "64" in str(e.value) and "alleles" in str(e.value)


@pytest.mark.parametrize("depth", [9000, 40000])
def test_loci_deeper_than_the_on_chip_sort(engine0, tmp_path, depth):
    """Deep loci (the reference's own example run has 58 k reads per locus; round 2's builder needed a second kernel beyond
    8192): the same batch as the host builder for shallow flanks and a deep core in ONE run, indel alleles, soft clips, pairs,
    and barcodes of very different sizes (a few giant ones: a part of the sorted list then begins inside a barcode)."""
    from smcounter_amd import devplanes
    rng = np.random.default_rng(depth)
    L = 400
    ref = "".join(rng.choice(list("ACGT"), size=L))
    fa_path = str(tmp_path / "deep.fa")
    open(fa_path, "w").write(">chrD\n" + ref + "\n")
    recs = []
    n_bc = max(50, depth // 9)
    for i in range(depth // 2):
        bc = int(rng.integers(0, n_bc)) if i % 3 else int(rng.integers(0, 5))       # a few giant barcodes
        start = 100 + int(rng.integers(0, 6)) if i % 50 else int(rng.integers(20, 200))   # a deep core, shallow flanks
        for mate in (0, 1):
            pos = start + (0 if mate == 0 else int(rng.integers(0, 8)))
            kind = rng.random()
            if kind < 0.01:
                cigar = [(0, 20), (1, 2), (0, 38)]; qlen = 60
            elif kind < 0.02:
                cigar = [(0, 25), (2, 3), (0, 35)]; qlen = 60
            elif kind < 0.05:
                cigar = [(4, 5), (0, 55)]; qlen = 60
            else:
                cigar = [(0, 60)]; qlen = 60
            seq = "".join(rng.choice(list("ACGT"), size=qlen)) if kind < 0.05 else ref[pos:pos + 60]
            if rng.random() < 0.02:
                seq = seq[:30] + "ACGT"[int(rng.integers(0, 4))] + seq[31:]
            recs.append(dict(tid=0, pos=pos, qname="r%d:x:BC%04d:y" % (i, bc), flag=(0x41 if mate == 0 else 0x91),
                             mapq=int(rng.choice([20, 60])), cigar=cigar, seq=seq,
                             qual=rng.choice([12, 25, 30, 37], size=qlen).astype(np.uint8).tolist(), nm=int(rng.integers(0, 3))))
    recs.sort(key=lambda r: r["pos"])
    bam = str(tmp_path / "deep.bam")
    bamio.write_bam(bam, [("chrD", L)], recs)
    bamio.write_bai(bam)
    fa = fasta.FastaFile(fa_path)
    loci = [("chrD", str(p)) for p in range(60, 200)]
    P = VcParams(mtDepth=100000, rpb=2.0, hpLen=8)
    host = list(bamio.iter_device_batches_native(bam, fa, loci, P, max_reads=64_000_000))
    dev = list(devplanes.iter_resident_batches(bam, fa, loci, P, engine0, max_reads=64_000_000))
    assert len(host) == len(dev) == 1
    (_, hb), (_, rb) = host[0], dev[0]
    assert int(hb.loci["n_reads"].max()) > 8192 and int(hb.loci["n_reads"].min()) < 8192
    assert rb.n_device_runs >= 1 and rb.n_host_runs == 0
    _same_batch(rb, hb)


def test_small_and_large_sort_segments_in_one_run(engine0, tmp_path):
    """A run of more than 512 loci whose first 512-locus segment holds a few hundred alignments (sorted by one workgroup,
    k_bp_sort_seg) and whose second has a tile beyond 16,384 (k_bp_hist / k_bp_scan / k_bp_scatter): the multi-launch passes
    have to put their output behind EVERY earlier segment, not only behind the earlier multi-launch ones."""
    from smcounter_amd import devplanes
    rng = np.random.default_rng(512)
    L = 1100
    ref = "".join(rng.choice(list("ACGT"), size=L))
    fa_path = str(tmp_path / "mix.fa")
    open(fa_path, "w").write(">chrS\n" + ref + "\n")
    recs = []
    for i in range(9400):
        deep = i >= 400
        start = 700 + int(rng.integers(0, 6)) if deep else int(rng.integers(20, 1000))
        bc = int(rng.integers(0, 700))
        for mate in (0, 1):
            pos = start + (0 if mate == 0 else int(rng.integers(0, 8)))
            cigar = [(0, 60)] if rng.random() > 0.03 else [(0, 20), (1, 2), (0, 38)]
            seq = ref[pos:pos + 60]
            recs.append(dict(tid=0, pos=pos, qname="r%d:x:BC%04d:y" % (i, bc), flag=(0x41 if mate == 0 else 0x91), mapq=60,
                             cigar=cigar, seq=seq, qual=rng.choice([12, 25, 30, 37], size=60).astype(np.uint8).tolist(), nm=0))
    recs.sort(key=lambda r: r["pos"])
    bam = str(tmp_path / "mix.bam")
    bamio.write_bam(bam, [("chrS", L)], recs)
    bamio.write_bai(bam)
    fa = fasta.FastaFile(fa_path)
    loci = [("chrS", str(p)) for p in range(40, 900)]
    P = VcParams(mtDepth=100000, rpb=2.0, hpLen=8)
    host = list(bamio.iter_device_batches_native(bam, fa, loci, P, max_reads=64_000_000))
    dev = list(devplanes.iter_resident_batches(bam, fa, loci, P, engine0, max_reads=64_000_000))
    assert len(host) == len(dev) == 1
    (_, hb), (_, rb) = host[0], dev[0]
    assert int(hb.loci["n_reads"].max()) > 16384 and int(hb.loci["n_reads"][:512].max()) < 200
    assert rb.n_device_runs >= 1 and rb.n_host_runs == 0
    _same_batch(rb, hb)


def test_a_tile_with_hundreds_of_distinct_indel_alleles(engine0, tmp_path):
    """More distinct extra alleles in one 64-locus tile than k_bp_xfix's report buffer holds (256): the surplus reports take the
    one-by-one path.  Every locus of the tile sees several insertions of different sequences and lengths and a few deletions;
    the allele tables, the numbering by first appearance and the planes must equal the host builder's."""
    from smcounter_amd import devplanes
    rng = np.random.default_rng(11)
    L = 300
    ref = "".join(rng.choice(list("ACGT"), size=L))
    fa_path = str(tmp_path / "x.fa")
    open(fa_path, "w").write(">chrX\n" + ref + "\n")
    recs = []
    i = 0
    for at in range(70, 140):                                   # an insertion / deletion right behind reference position `at`
        for v in range(7):
            start = at - 20 - int(rng.integers(0, 10))
            left = at + 1 - start
            if v < 5:
                ins = "".join(rng.choice(list("ACGT"), size=1 + v))
                cigar = [(0, left), (1, len(ins)), (0, 30)]
                seq = ref[start:start + left] + ins + ref[start + left:start + left + 30]
            else:
                dl = 1 + (v - 5)
                cigar = [(0, left), (2, dl), (0, 30)]
                seq = ref[start:start + left] + ref[start + left + dl:start + left + dl + 30]
            for mate in (0, 1):
                recs.append(dict(tid=0, pos=start, qname="r%d:x:BC%03d:y" % (i, int(rng.integers(0, 40))), flag=(0x41 if mate == 0 else 0x91),
                                 mapq=60, cigar=cigar, seq=seq, qual=[30] * len(seq), nm=1))
            i += 1
    recs.sort(key=lambda r: r["pos"])
    bam = str(tmp_path / "x.bam")
    bamio.write_bam(bam, [("chrX", L)], recs)
    bamio.write_bai(bam)
    fa = fasta.FastaFile(fa_path)
    loci = [("chrX", str(p)) for p in range(65, 65 + 128)]
    P = VcParams(mtDepth=1000, rpb=2.0, hpLen=8)
    host = list(bamio.iter_device_batches_native(bam, fa, loci, P))
    dev = list(devplanes.iter_resident_batches(bam, fa, loci, P, engine0))
    assert len(host) == len(dev) == 1
    (_, hb), (_, rb) = host[0], dev[0]
    assert rb.n_device_runs >= 1 and rb.n_host_runs == 0
    per_tile = [sum(len(t) - 6 for t in hb.alleles[k:k + 64]) for k in range(0, hb.n_loci, 64)]
    assert max(per_tile) > 256, per_tile                        # (the buffer really overflows)
    _same_batch(rb, hb)


def test_targets_out_of_order_and_scattered(engine0, tmp_path):
    """Targets visited out of coordinate order, single scattered positions among them (every stretch of consecutive positions
    is a run of its own; the decoder's cursor has to go back): same batches as the host builder."""
    import test_bamio
    from smcounter_amd import devplanes
    bam, fa_path, loci = test_bamio._random_bam(tmp_path, 31, True)
    fa = fasta.FastaFile(fa_path)
    blocks, cur = [], [loci[0]]
    for a, b in zip(loci, loci[1:]):
        if b[0] == a[0] and int(b[1]) == int(a[1]) + 1 and len(cur) < 17:
            cur.append(b)
        else:
            blocks.append(cur); cur = [b]
    blocks.append(cur)
    rng = np.random.default_rng(5)
    order = rng.permutation(len(blocks))
    shuffled = [x for k in order for x in blocks[int(k)]]
    shuffled = shuffled[::1][:len(shuffled) - 3] + shuffled[-1:] + shuffled[-3:-1]      # a scattered tail
    assert shuffled != loci and len(shuffled) == len(loci)
    P = VcParams(mtDepth=50, rpb=3.0, hpLen=8)
    host = list(bamio.iter_device_batches_native(bam, fa, shuffled, P, max_reads=4000, nthreads=2))
    dev = list(devplanes.iter_resident_batches(bam, fa, shuffled, P, engine0, max_reads=4000, nthreads=2))
    assert len(host) == len(dev) > 1
    for (f1, hb), (f2, rb) in zip(host, dev):
        assert f1 == f2
        _same_batch(rb, hb)


def test_cli_device_and_host_planes_write_the_same_files(tmp_path, monkeypatch):
    import bam_fixture
    from smcounter_amd import cli
    case = bam_fixture.make_case(str(tmp_path))
    outs = []
    for mode in ("device", "host"):
        monkeypatch.setenv("SMC_PLANES", mode)
        prefix = str(tmp_path / mode)
        cli.main(dict(outPrefix=prefix, bamFile=case["bam"], bedTarget=case["bed"], mtDepth=12, rpb=3.0, hpLen=8,
                      refGenome=case["fasta"], threshold=10))
        outs.append(open(prefix + ".smCounter.all.txt").read())
    assert outs[0] == outs[1] and outs[0].count("\n") > 10


def test_cli_sampler_option(tmp_path):
    """`--sampler philox` through the command line: files that differ from the default's (the reference-exact sample) at the loci
    over the UMI cap only, the same for every --batchReads (the sample does not depend on how the file is cut), another for another
    --samplerSeed; the help text says that it is NOT smCounter's sample."""
    import test_bamio
    from smcounter_amd import cli
    bam, fa_path, loci = test_bamio._random_bam(tmp_path, 23, True)
    bed = str(tmp_path / "t.bed")
    with open(bed, "w") as fh:                               # (the fixture's stretches of consecutive positions as BED intervals)
        keep = [(c, int(p)) for c, p in loci if c in ("chrA", "chrB")]
        start = prev = None
        for c, p in keep + [(None, 0)]:
            if start is not None and (c != start[0] or p != prev + 1):
                fh.write("%s\t%d\t%d\n" % (start[0], start[1] - 1, prev))
                start = None
            if c is not None and start is None:
                start = (c, p)
            prev = p

    def run(tag, **kw):
        prefix = str(tmp_path / tag)
        cli.main(dict(outPrefix=prefix, bamFile=bam, bedTarget=bed, mtDepth=4, rpb=3.0, hpLen=8, minBQ=15, minMQ=20, mismatchThr=8.0,
                      refGenome=fa_path, **kw))
        return open(prefix + ".smCounter.all.txt").read().splitlines()

    ref = run("ref")
    ph = run("ph", sampler="philox")
    ph_cut = run("phcut", sampler="philox", batchReads=200)
    ph_seed = run("phseed", sampler="philox", samplerSeed=9)
    assert len(ref) == len(ph) > 20 and ph == ph_cut
    differ = [i for i, (a, b) in enumerate(zip(ref, ph)) if a != b]
    assert differ and len(differ) < len(ref) - 1
    umt = ref[0].split("\t").index("UMT")
    assert all(ref[i].split("\t")[umt] == ph[i].split("\t")[umt] == "8" for i in differ)      # usedMT = ds = 2 x mtDepth at exactly those loci
    assert ph_seed != ph
    help_text = " ".join(cli.build_parser().format_help().split())
    assert "--sampler" in help_text and "NOT the reference's sample" in help_text


@pytest.mark.parametrize("name,n_loci", [("C2", 900), ("C3", 260), ("X2", 130)])
def test_synthetic_alignments_through_the_device_builder_and_through_the_decoder(engine0, tmp_path, name, n_loci):
    """bench.py's `from_alignments` input (synth.generate_alignments: the decoder's output format, made without a BAM) built
    by smc_build_planes, against the SAME alignments written as a BAM and taken through the real decoder and the host
    builder: equivalent planes, equal row strings, and the device-built batch's rows against the CPU restatement run on the
    host-built planes."""
    import oracle_lib
    from smcounter_amd import devplanes, rows, synth, vc
    cfg = synth.CONFIGS[name]
    P = synth.params_for(cfg)
    A = synth.generate_alignments(cfg, n_loci, P, p_ins_aln=0.03, p_del_aln=0.03)
    assert A["n_slots"] > 0 and int(A["loc"]["n"].min()) > 0
    rb = devplanes.resident_from_alignments(A, engine0, P)
    bam, fa_path = str(tmp_path / "s.bam"), str(tmp_path / "s.fa")
    chrom, p0, p1 = synth.alignments_to_bam(A, bam, 0, n_loci, fa_path)
    fa = fasta.FastaFile(fa_path)
    loci = [(chrom, str(p)) for p in range(p0, p1 + 1)]
    host = [b for _, b in bamio.iter_device_batches_native(bam, fa, loci, P, max_reads=1 << 40)]
    assert len(host) == 1
    hb = host[0]
    _same_batch(rb, hb)
    assert any(len(t) > 6 for t in hb.alleles)
    got = vc.vc_resident(rb, P, fa, engine0)
    assert got == rows.format_rows(engine0.call_batch_host(hb, P), hb, P, fa)
    plan = engine0.make_plan(rb.loci)
    dev_rows = plan.run_devbuf(rb.planes, P).copy()                              # raw-field planes: smc_pack_words, then the kernels
    assert plan.run_devbuf([rb.words, rb.planes[4]], P).tobytes() == dev_rows.tobytes()   # the builder's own read words
    plan.close()
    # words only (what the command line builds): the same strings
    rb1 = devplanes.resident_from_alignments(A, engine0, P, all_planes=False)
    assert rb1.planes[0] is None and vc.vc_resident(rb1, P, fa, engine0) == got
    want, fragile, pi_all = oracle_lib.call_batch(hb, abi.c_params(P), abi.ROW_DTYPE, return_fragile=True, return_pi_all=True)
    assert abi.compare_rows(dev_rows, want, 1e-6, 1e-6, fragile, pi_all) == []


def test_plan_made_on_the_device_gives_the_same_rows(engine0, tmp_path):
    """smc_plan_create_dev (descriptors classified and the launch lists written on the device) against smc_plan_create (host):
    identical row bytes - on a run that mixes every workgroup class (shallow flanks, a core beyond 24,576 reads: the deep class)
    and on a synthetic panel shape."""
    import test_bamio
    from smcounter_amd import devplanes, synth
    cases = []
    cfg = synth.CONFIGS["C3"]
    P = synth.params_for(cfg)
    cases.append((devplanes.resident_from_alignments(synth.generate_alignments(cfg, 300, P), engine0, P), P))
    rng = np.random.default_rng(77)
    ref = "".join(rng.choice(list("ACGT"), size=300))
    open(str(tmp_path / "m.fa"), "w").write(">chrM\n" + ref + "\n")
    recs = []
    for i in range(16000):
        start = 100 + int(rng.integers(0, 4)) if i % 9 else int(rng.integers(20, 150))
        for mate in (0, 1):
            pos = start + (0 if mate == 0 else int(rng.integers(0, 6)))
            recs.append(dict(tid=0, pos=pos, qname="r%d:x:BC%04d:y" % (i, int(rng.integers(0, 900))), flag=(0x41 if mate == 0 else 0x91),
                             mapq=60, cigar=[(0, 50)], seq=ref[pos:pos + 50], qual=[30] * 50, nm=0))
    recs.sort(key=lambda r: r["pos"])
    bam = str(tmp_path / "m.bam")
    bamio.write_bam(bam, [("chrM", 300)], recs)
    bamio.write_bai(bam)
    fa = fasta.FastaFile(str(tmp_path / "m.fa"))
    P2 = VcParams(mtDepth=100000, rpb=2.0, hpLen=8)
    loci = [("chrM", str(p)) for p in range(40, 190)]
    for _, rb in devplanes.iter_resident_batches(bam, fa, loci, P2, engine0, max_reads=64_000_000):
        assert int(rb.loci["n_reads"].max()) > 24576 and int(rb.loci["n_reads"].min()) < 4096
        cases.append((rb, P2))
    for rb, prm in cases:
        d_loci = devplanes.DevLoci(engine0, rb.loci)
        p_host, p_dev = engine0.make_plan(rb.loci), engine0.make_plan_dev(d_loci, rb.n_loci)
        r_host = p_host.run_devbuf(rb.planes, prm).copy()
        r_dev = p_dev.run_devbuf(rb.planes, prm).copy()
        p_host.close(); p_dev.close(); d_loci.free()
        assert r_host.tobytes() == r_dev.tobytes()


def test_plans_made_in_a_row_on_two_streams(engine0):
    """The statistics record smc_plan_create_dev sums into is the context's and is zeroed by the previous plan's k_plan_fill, not
    by a memset: six plans of three different batches made back to back on two streams (a plan waits for the event behind the
    previous plan's k_plan_fill), each run twice - the rows of the host-made plans, byte for byte."""
    import torch
    from smcounter_amd import devplanes, synth
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    batches = []
    for name, n in (("C3", 700), ("C2", 2500), ("X6", 150)):
        cfg = synth.CONFIGS[name]
        P = synth.params_for(cfg)
        db = synth.generate_native(cfg, 0, n, P)
        planes = engine0.upload(db)
        ph = engine0.make_plan(db.loci)
        want = ph.download(ph.run(planes, P)).tobytes()
        ph.close()
        batches.append((db, P, planes, devplanes.DevLoci(engine0, db.loci), want))
    torch.cuda.synchronize()
    plans = []
    for k in range(6):
        db, P, planes, d_loci, want = batches[k % 3]
        plans.append((engine0.make_plan_dev(d_loci, len(db.loci), stream=streams[k % 2]), k))
    for plan, k in plans:
        db, P, planes, d_loci, want = batches[k % 3]
        for _ in range(2):
            with torch.cuda.stream(streams[k % 2]):
                got = plan.download(plan.run(planes, P, stream=streams[k % 2]))
            assert got.tobytes() == want
    for plan, _ in plans:
        plan.close()
    for b in batches:
        b[3].free()


def test_long_and_odd_cigars(engine0, tmp_path):
    """k_bp_lin makes the general CIGARs' copies operation by operation, 16 lanes per alignment: alignments with 5 ... 70 operations
    (more than 64: the exact path), a deletion straight after an insertion and the other way round, a reference skip next to an
    indel, an insertion as the first and as the last aligned operation, hard + soft clips at both ends, operations of length 0 -
    the same batch as the host builder, position by position."""
    from smcounter_amd import devplanes
    rng = np.random.default_rng(77)
    L = 3000
    ref = "".join(rng.choice(list("ACGT"), size=L))
    fa_path = str(tmp_path / "odd.fa")
    open(fa_path, "w").write(">chrO\n" + ref + "\n")

    def build(pos, ops):
        """ops: [(op, len)] -> (cigar, seq): bases follow the reference except inserted / clipped ones"""
        seq, p = [], pos
        for op, n in ops:
            if op in (0, 7, 8):
                seq.append(ref[p:p + n]); p += n
            elif op in (1, 4):
                seq.append("".join(rng.choice(list("ACGT"), size=n)))
            elif op in (2, 3):
                p += n
        return ops, "".join(seq)

    shapes = [
        [(5, 3), (4, 4), (0, 30), (1, 2), (2, 3), (0, 20), (4, 3), (5, 2)],          # insertion then deletion
        [(0, 25), (2, 2), (1, 3), (0, 25)],                                           # deletion then insertion
        [(0, 20), (3, 40), (1, 1), (0, 20)],                                          # reference skip, then an insertion
        [(0, 20), (1, 2), (3, 30), (0, 20)],                                          # insertion, then a reference skip
        [(4, 2), (1, 3), (0, 40)],                                                    # an insertion as the first aligned operation
        [(0, 40), (1, 3), (4, 2)],                                                    # ... and as the last
        [(0, 10), (1, 0), (0, 10), (2, 0), (0, 30)],                                  # operations of length 0
        [(0, 30), (2, 1), (0, 1), (2, 1), (0, 30)],                                   # a one-base match between two deletions
    ]
    recs = []
    for i in range(1800):
        pos = int(rng.integers(50, L - 800))
        k = i % (len(shapes) + 3)
        if k < len(shapes):
            ops = shapes[k]
        else:                                                                       # many small indels: 2 m + 1 operations
            m = [2, 15, 35][k - len(shapes)]                                         # 5, 31 and 71 operations
            ops = []
            for j in range(m):
                ops += [(0, int(rng.integers(3, 9))), ((1, int(rng.integers(1, 3))) if j % 2 else (2, int(rng.integers(1, 4))))]
            ops.append((0, 6))
        cigar, seq = build(pos, ops)
        recs.append(dict(tid=0, pos=pos, qname="r%d:n%d:BC%03d:y" % (i, i // 2, i % 61), flag=(0x41 if i % 2 == 0 else 0x91),
                         mapq=60, cigar=cigar, seq=seq, qual=rng.choice([12, 25, 30, 37], size=len(seq)).astype(np.uint8).tolist(),
                         nm=int(rng.integers(0, 3))))
    recs.sort(key=lambda r: r["pos"])
    bam = str(tmp_path / "odd.bam")
    bamio.write_bam(bam, [("chrO", L)], recs)
    bamio.write_bai(bam)
    fa = fasta.FastaFile(fa_path)
    loci = [("chrO", str(p)) for p in range(40, L - 300)]
    P = VcParams(mtDepth=100000, rpb=2.0, hpLen=8)
    host = list(bamio.iter_device_batches_native(bam, fa, loci, P, max_reads=64_000_000))
    dev = list(devplanes.iter_resident_batches(bam, fa, loci, P, engine0, max_reads=64_000_000))
    assert len(host) == len(dev) == 1
    (_, hb), (_, rb) = host[0], dev[0]
    assert rb.n_device_runs >= 1 and rb.n_host_runs == 0
    _same_batch(rb, hb)


def test_plans_made_without_the_host_fit_or_say_so(engine0):
    """smc_plan_create_dev_spec (ABI 8): the launches of a plan sized from the context's LAST plan instead of this batch's own record -
    nothing waits for the device.  The first plan of a context goes the exact way; a batch of the same kind (the same, or twice as
    many loci) fits and gives the exact plan's row bytes; a batch that does not fit (deeper loci: another workgroup class) launches
    nothing, `ok()` says so, the device's counter counts it, and the plan made again goes the exact way and gives the rows."""
    import torch
    from smcounter_amd import devplanes, synth
    from smcounter_amd.engine import Engine
    eng = Engine(0)                                              # (a context of its own: the hints are the context's)
    try:
        def batch(name, lo, n):
            cfg = synth.CONFIGS[name]
            P = synth.params_for(cfg)
            db = synth.generate_native(cfg, lo, lo + n, P)
            raw = eng.upload(db)
            ph = eng.make_plan(db.loci)
            want = ph.download(ph.run(raw, P)).tobytes()
            words = ph.pack_words(raw[0], raw[2], torch.empty_like(raw[0]))      # (such a plan runs read words)
            torch.cuda.synchronize()
            ph.close()
            return db, P, [words, raw[4]], devplanes.DevLoci(eng, db.loci), want

        def run_spec(b):
            db, P, planes, d_loci, want = b
            plan = eng.make_plan_dev(d_loci, len(db.loci), spec_params=P)
            rows = plan.run(planes, P)
            torch.cuda.synchronize()
            ok = plan.ok()
            got = plan.download(rows).tobytes()
            plan.close()
            return ok, got == want

        a1, a2, a3 = batch("C2", 0, 1500), batch("C2", 1500, 1500), batch("C2", 3000, 3000)
        assert run_spec(a1) == (True, True)                      # the context's first: made the exact way
        assert eng.spec_counts() == (0, 1, 0)
        assert run_spec(a2) == (True, True)                      # sized from a1's record
        assert run_spec(a3) == (True, True)                      # ... scaled to twice the loci
        assert run_spec(a1) == (True, True)
        assert eng.spec_counts() == (3, 1, 0)
        deep = batch("C5", 0, 300)                               # 8,000 reads per locus: workgroups of 128 threads - no such launch was sized
        ok, same = run_spec(deep)
        assert ok is False
        assert eng.spec_counts() == (4, 1, 1)
        assert run_spec(deep) == (True, True)                    # made again: the exact way
        assert eng.spec_counts() == (4, 2, 1)
        assert run_spec(deep) == (True, True) and run_spec(a2)[0] is False   # (the sizes follow the data: now C2's loci do not fit C5's launches)
        assert run_spec(a2) == (True, True)                                     # (made again: exact)
        with pytest.raises(Exception, match="raw-field planes"):                # (and such a plan says so when handed raw-field planes)
            db, P, planes, d_loci, want = a1
            raw = eng.upload(db)
            plan = eng.make_plan_dev(d_loci, len(db.loci), spec_params=P)
            try:
                plan.run(raw, P)
            finally:
                plan.close()
        eng.reset_plan_hint()                                                   # (told that another kind of batch follows: no misfit)
        assert run_spec(deep) == (True, True)
        assert eng.spec_counts()[2] == 2
        for b in (a1, a2, a3, deep):
            b[3].free()
    finally:
        eng.close()


def test_the_non_parity_sampler_on_the_device_equals_its_restatement(engine0):
    """smc_philox_marks (the Philox4x32-10 down-sampling SURVEY.md a4 allows as a non-parity mode) against oracle/smc_oracle.c's
    restatement on a batch whose loci are over the barcode cap: the same marks in umi_start, SMC_LF_SAMPLED set, and the rows of the
    marked batch through the locus kernels equal the oracle's - with the barcode's index as its identity, with identities per
    barcode entry, and with a table of identities behind an index array (what the plane builder leaves in u_gid)."""
    import ctypes
    import dataclasses
    import torch
    import oracle_lib
    from smcounter_amd import _lib, devplanes, synth
    cfg = synth.SynthConfig("PH", 96, 70, 12, 20260102)
    P = dataclasses.replace(synth.params_for(cfg), maxMT=30)
    db = synth.generate_native(cfg, 0, 96, P)
    cp = abi.c_params(P)
    pos = np.arange(5000, 5096, dtype=np.int64)
    rng = np.random.default_rng(3)
    n_ent = len(db.umi_start)
    ident_entry = rng.integers(0, 2 ** 63, size=n_ent, dtype=np.uint64)
    table = rng.integers(0, 2 ** 63, size=500, dtype=np.uint64)
    index = rng.integers(0, 500, size=n_ent).astype(np.uint32)
    raw = engine0.upload(db)
    ph = engine0.make_plan(db.loci)
    words = ph.pack_words(raw[0], raw[2], torch.empty_like(raw[0]))
    torch.cuda.synchronize()
    ph.close()
    L = engine0.L
    dev = raw[0].device
    for ident, idx, want_ident in ((None, None, None), (ident_entry, None, ident_entry), (table, index, table[index])):
        d_loci = devplanes.DevLoci(engine0, db.loci)
        us = raw[4].clone()
        d_pos = torch.from_numpy(pos).to(dev)
        st = torch.zeros(4, dtype=torch.int32, device=dev)
        d_id = torch.from_numpy(ident.view(np.int64)).to(dev) if ident is not None else None
        d_ix = torch.from_numpy(idx.view(np.int32)).to(dev) if idx is not None else None
        _lib.check(L.smc_philox_marks(engine0.ctx, ctypes.byref(cp), d_loci.data_ptr(), len(db.loci), d_pos.data_ptr(), words.data_ptr(), 32,
                                      us.data_ptr(), d_id.data_ptr() if d_id is not None else None, d_ix.data_ptr() if d_ix is not None else None,
                                      ctypes.c_uint64(11), st.data_ptr(), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "smc_philox_marks")
        torch.cuda.synchronize()
        got_us = us.cpu().numpy().view(np.uint32)
        got_loci = d_loci.download(features.LOCUS_DTYPE, len(db.loci))
        want_loci, want_us = oracle_lib.philox_marks(db, cp, pos, seed=11, ident=want_ident)
        assert int(st[0]) == 0 and np.array_equal(got_us, want_us) and got_loci.tobytes() == want_loci.tobytes()
        assert (got_loci["flags"] & features.LF_SAMPLED).all() and (got_us >> 31).sum() > 0
        plan = engine0.make_plan(got_loci)
        got = plan.download(plan.run([words, us], P))
        plan.close()
        want = oracle_lib.call_batch(dataclasses.replace(db, loci=want_loci, umi_start=want_us), cp, abi.ROW_DTYPE)
        assert abi.compare_rows(got, want) == []
        assert (got["used_mt"] == P.ds).all() and ((got["status"] & abi.ST_DOWNSAMPLED) != 0).all()
        d_loci.free()


@pytest.mark.parametrize("all_planes", [False, True])
def test_the_non_parity_sampler_does_not_depend_on_how_the_file_is_cut(engine0, tmp_path, all_planes):
    """--sampler philox through the BAM path: a barcode's identity is a hash of its TEXT, so the rows of the loci over the barcode
    cap are the same whether the file is taken as one batch or cut into many runs and batches (where the run-wide barcode ids, the
    barcodes' numbers at a locus and the launches all differ); they differ from the reference-exact sample's rows, carry
    SMC_ST_DOWNSAMPLED and usedMT = ds; loci under the cap are untouched by the choice of sampler."""
    import test_bamio
    from smcounter_amd import devplanes, vc
    bam, fa_path, loci = test_bamio._random_bam(tmp_path, 23, True)
    fa = fasta.FastaFile(fa_path)
    P = VcParams(mtDepth=4, rpb=3.0, hpLen=8, minBQ=15, minMQ=20, mismatchThr=8.0)       # ds = 8 < the file's 25 barcodes

    def rows_of(max_reads, sampler, seed=0):
        out = []
        for _, rb in devplanes.iter_resident_batches(bam, fa, loci, P, engine0, max_reads=max_reads, nthreads=2, all_planes=all_planes,
                                                     sampler=sampler, sampler_seed=seed):
            out.append(vc.vc_resident_rows(rb, P, engine0))
        return np.concatenate(out)

    whole, cut = rows_of(2_000_000, "philox"), rows_of(3000, "philox")
    assert len(whole) == len(cut) == len(loci)
    assert whole.tobytes() == cut.tobytes()
    ref = rows_of(2_000_000, "reference")
    over = (whole["status"] & abi.ST_DOWNSAMPLED) != 0
    assert over.any() and np.array_equal(over, (ref["status"] & abi.ST_DOWNSAMPLED) != 0)
    assert (whole["used_mt"][over] == P.ds).all()
    assert whole[~over].tobytes() == ref[~over].tobytes()
    assert whole[over].tobytes() != ref[over].tobytes()                         # (another subset of the same size)
    assert rows_of(2_000_000, "philox", seed=5)[over].tobytes() != whole[over].tobytes()
