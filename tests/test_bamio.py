"""f2: BGZF/BAM reader and single-position pileup against a naive expansion of the same reads."""
import os
import numpy as np
import pytest

import bam_fixture
from smcounter_amd import bamio, bedops, fasta, features, pileup, rows, abi
from smcounter_amd.params import VcParams

import oracle_lib


@pytest.fixture(autouse=True)
def _experiment_switches(monkeypatch):
    """The environment switches these tests flip (chunk geometries, poisoned scratch, forced code paths) are experiment knobs: the
    libraries read them only under SMC_EXPERIMENTAL."""
    monkeypatch.setenv("SMC_EXPERIMENTAL", "1")


def _naive_columns(recs, pos0):
    """Expand every read base by base (no shortcuts) and report what sits on pos0."""
    out = []
    for r in recs:
        x, y = r["pos"], 0
        ops = r["cigar"]
        hit = None
        for k, (op, l) in enumerate(ops):
            for t in range(l):
                if op == 0:
                    if x == pos0:
                        indel = 0
                        if t == l - 1 and k + 1 < len(ops) and ops[k + 1][0] in (1, 2):
                            indel = ops[k + 1][1] if ops[k + 1][0] == 1 else -ops[k + 1][1]
                        hit = (y, False, indel)
                    x += 1; y += 1
                elif op in (2, 3):
                    if x == pos0:
                        hit = (y, True, 0)
                    x += 1
                elif op in (1, 4):
                    y += 1
        if hit:
            out.append((r["qname"], hit))
    return out


def test_bam_roundtrip_and_pileup(tmp_path):
    case = bam_fixture.make_case(str(tmp_path))
    bam = bamio.BamFile(case["bam"])
    assert bam.refs == [("chrQ", 1000)]
    got = bam.fetch("chrQ", 0, 1000)
    assert len(got) == len(case["records"])
    for a, r in zip(got, case["records"]):
        assert (a.qname, a.pos, a.flag, a.mapq, a.seq, list(a.qual)) == \
            (r["qname"], r["pos"], r["flag"], r["mapq"], r["seq"], r["qual"])
        assert a.cigar == r["cigar"] and a.has_nm == (r["nm"] is not None) and a.nm == (r["nm"] or 0)
    fa = fasta.FastaFile(case["fasta"])
    loci = bedops.expand_loci(case["bed"])
    assert len(loci) == 46
    batches = list(bamio.iter_pileup_batches(bam, fa, loci, max_reads=700))
    assert len(batches) > 1 and batches[0][0] == 0
    pb = pileup.concat([b for _, b in batches])
    assert pb.n_loci == 46
    for l, (chrom, p1) in enumerate(loci):
        exp = _naive_columns(case["records"], int(p1) - 1)
        s = pb.locus_slice(l)
        assert s.stop - s.start == len(exp)
        assert pb.ref[l] == case["ref"][int(p1) - 1]
        for i, (qname, (qpos, is_del, indel)) in zip(range(s.start, s.stop), exp):
            assert bool(pb.is_del[i]) == is_del and int(pb.indel[i]) == indel
            if not is_del:
                assert int(pb.qpos[i]) == qpos
        # ids dense by first appearance
        seen = []
        for u in pb.umi[s]:
            if u not in seen:
                seen.append(int(u))
        assert seen == list(range(len(seen)))
    # allele keys follow the reference's spelling
    keys = {k for t in pb.alleles for k in t[6:]}
    assert any(k.startswith("INS|") for k in keys) and any(k.startswith("DEL|") for k in keys)
    for k in keys:
        if k.startswith("DEL|"):
            _, rd, r = k.split("|")
            assert rd[0] == r and len(rd) == 4


def test_bam_to_rows_finds_the_planted_variant(tmp_path):
    case = bam_fixture.make_case(str(tmp_path), n_umi=16, frags_per_umi=6)
    P = VcParams(mtDepth=16, rpb=3.0, hpLen=8)
    fa = fasta.FastaFile(case["fasta"])
    pb = pileup.concat([b for _, b in bamio.iter_pileup_batches(bamio.BamFile(case["bam"]), fa,
                                                                 bedops.expand_loci(case["bed"]))])
    db = features.extract_features(pb, P)
    R = oracle_lib.call_batch(db, abi.c_params(P), abi.ROW_DTYPE)
    text = rows.format_rows(R, db, P, fa)
    by_pos = {t.split("\t")[1]: t.split("\t") for t in text}
    f = by_pos[str(case["snp_pos"] + 1)]
    assert f[2] == case["ref"][case["snp_pos"]] and f[3] == case["alt"] and f[4] == "SNP"
    assert float(f[10]) > 15.0 and 0.3 < float(f[14]) < 0.7          # PI, VMF ~ half the barcodes
    hp = by_pos["601"]                                                # inside the A x 12 run
    assert hp[2] == "A"


def test_reference_example_bai_parses():
    import os
    p = "/root/reference/example/example.bam.bai"
    if not os.path.exists(p):
        pytest.skip("reference example index not on this box")
    lin = bamio._load_bai(p, 0)
    assert lin and any(len(x) for x in lin)


@pytest.mark.gpu
def test_cli_end_to_end_on_gpu(tmp_path):
    """The command line on a BAM: rows of <prefix>.smCounter.all.txt equal the CPU restatement's rows
    (after FILTER normalisation) and the planted SNV lands in cut.vcf."""
    from smcounter_amd import cli, postfilter
    case = bam_fixture.make_case(str(tmp_path), n_umi=16, frags_per_umi=6)
    prefix = str(tmp_path / "run")
    thr = cli.main(dict(outPrefix=prefix, bamFile=case["bam"], bedTarget=case["bed"], mtDepth=16, rpb=3.0,
                        hpLen=8, refGenome=case["fasta"], threshold=10))
    assert thr == 10
    got = open(prefix + ".smCounter.all.txt").read().split("\n")[1:-1]
    P = VcParams(mtDepth=16, rpb=3.0, hpLen=8)
    fa = fasta.FastaFile(case["fasta"])
    pb = pileup.concat([b for _, b in bamio.iter_pileup_batches(bamio.BamFile(case["bam"]), fa,
                                                                 bedops.expand_loci(case["bed"]))])
    db = features.extract_features(pb, P)
    R, fragile = oracle_lib.call_batch(db, abi.c_params(P), abi.ROW_DTYPE, return_fragile=True)
    want = postfilter.apply_repeat_filters(rows.format_rows(R, db, P, fa), {}, {})
    # loci holding a barcode whose consensus is decided by rounding are not pinned (abi.compare_rows)
    assert (fragile > 0).sum() <= 3
    for l, (g, w) in enumerate(zip(got, want)):
        if not fragile[l]:
            assert g == w, l
    vcf = [l for l in open(prefix + ".smCounter.cut.vcf") if not l.startswith("#")]
    assert any(l.split("\t")[1] == str(case["snp_pos"] + 1) and l.split("\t")[4] == case["alt"] for l in vcf)


@pytest.mark.parametrize("max_reads", [700, 2_000_000])
def test_native_decoder_matches_python_decoder(tmp_path, max_reads):
    """csrc/smc_bam.cpp yields the same chunks, columns and allele tables as the Python decoder."""
    case = bam_fixture.make_case(str(tmp_path))
    fa = fasta.FastaFile(case["fasta"])
    loci = bedops.expand_loci(case["bed"])
    py = list(bamio.iter_pileup_batches(bamio.BamFile(case["bam"]), fa, loci, max_reads=max_reads))
    nat = list(bamio.iter_pileup_batches_native(case["bam"], fa, loci, max_reads=max_reads))
    assert [f for f, _ in py] == [f for f, _ in nat]
    for (_, a), (_, b) in zip(py, nat):
        assert a.chrom == b.chrom and a.ref == b.ref and a.alleles == b.alleles
        assert np.array_equal(a.pos, b.pos) and np.array_equal(a.read_off, b.read_off)
        for k, _ in bamio._COLS:
            x, y = getattr(a, k), getattr(b, k)
            assert x.dtype == y.dtype and np.array_equal(x, y), k


def test_native_decoder_errors(tmp_path):
    p = str(tmp_path / "bad.bam")
    bamio.write_bam(p, [("chrQ", 100)], [dict(tid=0, pos=5, qname="nocolons", flag=0, mapq=60, cigar=[(0, 4)],
                                              seq="ACGT", qual=[30] * 4, nm=0)])
    case = bam_fixture.make_case(str(tmp_path))
    fa = fasta.FastaFile(case["fasta"])
    with pytest.raises(bamio.BamError, match="fewer than 3"):
        list(bamio.iter_pileup_batches_native(p, fa, [("chrQ", "6")]))
    with pytest.raises(bamio.BamError):
        bamio.NativeBam(case["fasta"])
    # unknown chromosome: empty loci, like the Python decoder
    out = list(bamio.iter_pileup_batches_native(case["bam"], fa, [("chrQ", "1000")]))
    assert out[0][1].n_loci == 1


def _random_bam(tmp, seed, with_bai):
    """Two chromosomes, reads with random CIGARs (S/M/I/D/N), sparse loci more than 16 kb apart."""
    rng = np.random.RandomState(seed)
    refs = [("chrA", 90000), ("chrB", 50000)]
    seqs = {n: "".join(rng.choice(list("ACGT"), l)) for n, l in refs + [("chrFastaOnly", 100)]}
    fa_path = str(tmp / "r.fa")
    with open(fa_path, "w") as fh:
        for n in seqs:
            fh.write(">%s\n" % n)
            for i in range(0, len(seqs[n]), 60):
                fh.write(seqs[n][i:i + 60] + "\n")
    centres = [(0, 300), (0, 40000), (0, 88000), (1, 20000), (1, 49000)]
    recs = []
    for tid, c in centres:
        for u in range(25):
            for fr in range(rng.randint(1, 4)):
                for mate in (0, 1):
                    pos = c - rng.randint(20, 120)
                    cig, ln = [], 0
                    if rng.rand() < 0.3:
                        cig.append((4, int(rng.randint(1, 8))))
                    for _ in range(rng.randint(1, 4)):
                        cig.append((0, int(rng.randint(20, 60))))
                        r = rng.rand()
                        if r < 0.2:
                            cig.append((1, int(rng.randint(1, 4))))
                        elif r < 0.4:
                            cig.append((2, int(rng.randint(1, 6))))
                        elif r < 0.45:
                            cig.append((3, int(rng.randint(5, 30))))
                    if cig[-1][0] != 0:
                        cig.append((0, int(rng.randint(5, 20))))
                    qlen = sum(l for op, l in cig if op in (0, 1, 4))
                    recs.append(dict(tid=tid, pos=max(0, pos), qname="rd%d_%d_%d:x:UMI%02d:0" % (c, u, fr, u),
                                     flag=(0x40 if mate == 0 else 0x80) | (0x10 if rng.rand() < 0.5 else 0) | 1,
                                     mapq=int(rng.randint(0, 61)), cigar=cig,
                                     seq="".join(rng.choice(list("ACGTN"), qlen, p=[.24, .24, .24, .24, .04])),
                                     qual=[int(x) for x in rng.randint(2, 42, qlen)],
                                     nm=None if rng.rand() < 0.1 else int(rng.randint(0, 9))))
    recs.sort(key=lambda r: (r["tid"], r["pos"]))
    bam = str(tmp / "r.bam")
    bamio.write_bam(bam, refs, recs, block=4000)
    if with_bai:
        bamio.write_bai(bam)
    loci = []
    for tid, c in centres:
        hi = min(refs[tid][1], c + 40)
        loci += [(refs[tid][0], str(p)) for p in range(max(1, c - 40), hi)]
    loci += [("chrA", "60000"), ("chrFastaOnly", "5")]          # empty locus / unknown chromosome
    return bam, fa_path, loci


@pytest.mark.parametrize("with_bai", [False, True])
def test_native_decoder_random_cigars_multi_chrom(tmp_path, with_bai):
    bam, fa_path, loci = _random_bam(tmp_path, 11, with_bai)
    fa = fasta.FastaFile(fa_path)
    if with_bai:
        assert bamio.BamFile(bam)._lin_index
    py = pileup.concat([b for _, b in bamio.iter_pileup_batches(bamio.BamFile(bam), fa, loci, max_reads=5000)])
    nat = pileup.concat([b for _, b in bamio.iter_pileup_batches_native(bam, fa, loci, max_reads=5000)])
    assert py.n_loci == nat.n_loci == len(loci) and int(py.read_off[-1]) > 10000
    assert py.chrom == nat.chrom and py.ref == nat.ref and py.alleles == nat.alleles
    assert np.array_equal(py.read_off, nat.read_off)
    for k, _ in bamio._COLS:
        assert np.array_equal(getattr(py, k), getattr(nat, k)), k
    assert any(len(t) > 6 for t in nat.alleles)


def _assert_device_batches_equal(a, b):
    assert a.loci.dtype == b.loci.dtype
    for f in a.loci.dtype.names:
        assert np.array_equal(a.loci[f], b.loci[f]), f
    for k in ("meta", "umi", "frag", "dist", "umi_start", "pos"):
        x, y = getattr(a, k), getattr(b, k)
        assert x.dtype == y.dtype and np.array_equal(x, y), k
    assert a.chrom == b.chrom and a.ref == b.ref and a.alleles == b.alleles


@pytest.mark.parametrize("mt_depth", [1000, 4])        # 4 -> ds = 8 < 25 barcodes: the py2 down-sampling path
@pytest.mark.parametrize("nthreads", [1, 3, "4 shards"])   # ("4 shards": barcode / read-name interning sharded by hash over threads)
@pytest.mark.parametrize("max_reads", [5000, 2_000_000])
def test_native_fused_planes_match_extract_features(tmp_path, max_reads, nthreads, mt_depth, monkeypatch):
    if nthreads == "4 shards":
        monkeypatch.setenv("SMC_BAM_SHARDS", "4")
        nthreads = 4
    """smc_bam_planes (decode + features + barcode-major order in one native pass) builds the same
    DeviceBatch, chunk for chunk, as extract_features over the Python decoder's batches."""
    bam, fa_path, loci = _random_bam(tmp_path, 23, True)
    fa = fasta.FastaFile(fa_path)
    P = VcParams(mismatchThr=4.0, mtDepth=mt_depth)
    want = [(f, features.extract_features(pb, P))
            for f, pb in bamio.iter_pileup_batches(bamio.BamFile(bam), fa, loci, max_reads=max_reads)]
    got = list(bamio.iter_device_batches_native(bam, fa, loci, P, max_reads=max_reads, nthreads=nthreads))
    assert [f for f, _ in want] == [f for f, _ in got]
    for (_, a), (_, b) in zip(want, got):
        _assert_device_batches_equal(a, b)
    # a batch whose runs do not fit the shared arrays falls back to per-run arrays + concatenation: same batches
    slack = bamio._ARENA_SLACK
    try:
        bamio._ARENA_SLACK = -max_reads // 2 if max_reads < 10**6 else 0
        again = list(bamio.iter_device_batches_native(bam, fa, loci, P, max_reads=max_reads, nthreads=nthreads))
    finally:
        bamio._ARENA_SLACK = slack
    assert len(again) == len(got)
    for (_, a), (_, b) in zip(got, again):
        _assert_device_batches_equal(a, b)
    flags = np.concatenate([b.meta for _, b in got]) >> 16 & 0xff
    assert len(set((flags >> 3).tolist())) == 4 and (flags & 4).any() and not (flags & 4).all()
    sampled = np.concatenate([b.loci["flags"] for _, b in got]) & features.LF_SAMPLED
    assert sampled.any() == (mt_depth == 4)
    if mt_depth == 4:
        us = np.concatenate([b.umi_start for _, b in got])
        assert (us & features.USTART_DROPPED).any()


def test_native_fused_planes_fixture_and_errors(tmp_path):
    case = bam_fixture.make_case(str(tmp_path))
    fa = fasta.FastaFile(case["fasta"])
    loci = bedops.expand_loci(case["bed"])
    P = VcParams()
    want = features.extract_features(pileup.concat(
        [b for _, b in bamio.iter_pileup_batches(bamio.BamFile(case["bam"]), fa, loci)]), P)
    got = list(bamio.iter_device_batches_native(case["bam"], fa, loci, P))
    assert len(got) == 1
    _assert_device_batches_equal(want, got[0][1])
    # a first read with neither READ1 nor READ2: the reference's pairOrder is undefined there
    p = str(tmp_path / "unpaired.bam")
    bamio.write_bam(p, [("chrQ", 1000)], [dict(tid=0, pos=5, qname="r:x:U:0", flag=0, mapq=60, cigar=[(0, 4)],
                                               seq="ACGT", qual=[30] * 4, nm=0)])
    with pytest.raises(features.PileupError, match="neither read1 nor read2"):
        list(bamio.iter_device_batches_native(p, fa, [("chrQ", "6")], P))
    p = str(tmp_path / "hiq.bam")
    bamio.write_bam(p, [("chrQ", 1000)], [dict(tid=0, pos=5, qname="r:x:U:0", flag=0x40, mapq=60, cigar=[(0, 4)],
                                               seq="ACGT", qual=[200] * 4, nm=0)])
    with pytest.raises(features.PileupError, match="base quality"):
        list(bamio.iter_device_batches_native(p, fa, [("chrQ", "6")], P))


def test_native_decoder_any_locus_order(tmp_path):
    """The decoder resumes a run where the previous one found its first overlapping record (streaming cursor):
    must not matter when the loci come out of order or jump between references."""
    bam, fa_path, loci = _random_bam(tmp_path, 31, True)
    fa = fasta.FastaFile(fa_path)
    rng = np.random.RandomState(3)
    blocks = [loci[i:i + 7] for i in range(0, len(loci), 7)]
    order = rng.permutation(len(blocks))
    shuffled = [l for b in order for l in blocks[b]]
    P = VcParams(mismatchThr=4.0)
    want = pileup.concat([b for _, b in bamio.iter_pileup_batches(bamio.BamFile(bam), fa, shuffled, max_reads=3000)])
    dbw = features.extract_features(want, P)
    got = list(bamio.iter_device_batches_native(bam, fa, shuffled, P, max_reads=3000, nthreads=2))
    assert sum(b.n_loci for _, b in got) == len(shuffled)
    off = 0
    for _, b in got:
        for l in range(b.n_loci):
            o, n = b.read_off(l), int(b.loci["n_reads"][l])
            ow = dbw.read_off(off + l)
            assert n == int(dbw.loci["n_reads"][off + l]) and b.pos[l] == dbw.pos[off + l]
            for k in ("meta", "umi", "frag", "dist"):
                assert np.array_equal(getattr(b, k)[o:o + n], getattr(dbw, k)[ow:ow + n]), k
        off += b.n_loci


@pytest.mark.parametrize("nthreads", [1, 4])
def test_native_decoder_many_blocks_several_index_windows(tmp_path, nthreads):
    """A BAM of a few hundred BGZF blocks over five 16 kb index windows, read in many small runs: the decoder's
    read-ahead bound (linear-index hint) and its streaming cursor never cost a read - same batches as the Python
    decoder, in file order and for targets visited out of order."""
    rng = np.random.RandomState(5)
    L, RL = 80000, 100
    seq = "".join(rng.choice(list("ACGT"), L))
    fa_path = str(tmp_path / "w.fa")
    with open(fa_path, "w") as fh:
        fh.write(">chrW\n")
        for i in range(0, L, 60):
            fh.write(seq[i:i + 60] + "\n")
    recs = []
    for i in range(9000):
        pos = int(rng.randint(100, L - 2 * RL))
        cigar = [(0, RL)] if i % 50 else [(0, 40), (3, 20000 if pos < 30000 else 50), (0, 60)]    # a few long N skips
        qual = rng.randint(20, 41, RL).astype(np.uint8).tolist()
        recs.append(dict(tid=0, pos=pos, qname="m:%d:r%d:NN:%s:x" % (i % 7, i // 2, "ACGT"[i % 4] * 3 + "TG"[i % 2] * 2),
                         flag=(0x41 if i % 2 == 0 else 0x91), mapq=60, cigar=cigar,
                         seq=seq[pos:pos + RL], qual=qual, nm=i % 3))
    recs.sort(key=lambda r: r["pos"])
    bam = str(tmp_path / "w.bam")
    bamio.write_bam(bam, [("chrW", L)], recs, block=6000)            # ~300 BGZF blocks
    bamio.write_bai(bam)
    fa = fasta.FastaFile(fa_path)
    targets = [(150, 260), (16300, 16420), (16500, 16530), (33000, 33100), (49100, 49160), (65500, 65600), (77000, 77050)]
    loci = [("chrW", str(p + 1)) for a, b in targets for p in range(a, b)]
    shuffled = [("chrW", str(p + 1)) for a, b in [targets[i] for i in (3, 0, 6, 1, 5, 2, 4)] for p in range(a, b)]
    P = VcParams(mismatchThr=4.0, mtDepth=1000)
    assert os.path.getsize(bam) > 200 * 1000
    for ll in (loci, shuffled):
        py = pileup.concat([b for _, b in bamio.iter_pileup_batches(bamio.BamFile(bam), fa, ll, max_reads=900)])
        nat = pileup.concat([b for _, b in bamio.iter_pileup_batches_native(bam, fa, ll, max_reads=900)])
        assert py.n_reads > 3000
        assert np.array_equal(py.pos, nat.pos) and np.array_equal(py.read_off, nat.read_off)
        assert py.alleles == nat.alleles
        for k, _ in bamio._COLS:
            assert np.array_equal(getattr(py, k), getattr(nat, k)), k
        # the fused path (threads inflate and parse)
        want = [(f, features.extract_features(pb, P))
                for f, pb in bamio.iter_pileup_batches(bamio.BamFile(bam), fa, ll, max_reads=900)]
        got = list(bamio.iter_device_batches_native(bam, fa, ll, P, max_reads=900, nthreads=nthreads))
        assert [f for f, _ in want] == [f for f, _ in got]
        for (_, a), (_, b) in zip(want, got):
            _assert_device_batches_equal(a, b)
    # the linear index as a volume estimate (what sizes a batch's first run): the whole reference is about the whole file,
    # a part of it a part, an unknown reference unknown
    nb = bamio.NativeBam(bam)
    whole, part = nb.span_bytes("chrW", 0, L), nb.span_bytes("chrW", 20000, 40000)
    assert 0.8 * os.path.getsize(bam) < whole <= os.path.getsize(bam) and 0 < part < 0.6 * whole
    assert nb.span_bytes("chrQ", 0, 100) == -1
    nb.close()


def test_locus_weights_follow_depth_and_balance_the_shards(tmp_path):
    """Two amplicons, the second eight times deeper: the BAI-derived weights put most of the first amplicon's loci
    and few of the second's into rank 0, the read counts of the two halves come out close."""
    from smcounter_amd import dist
    rng = np.random.RandomState(3)
    L, RL = 60000, 100
    seq = "".join(rng.choice(list("ACGT"), L))
    recs = []
    for start, n in ((1000, 300), (40000, 2400)):
        for i in range(n):
            pos = start + int(rng.randint(0, 60))
            recs.append(dict(tid=0, pos=pos, qname="m:1:r%d_%d:NN:%s:x" % (start, i // 2, "ACGT"[i % 4] * 4),
                             flag=(0x41 if i % 2 == 0 else 0x91), mapq=60, cigar=[(0, RL)], seq=seq[pos:pos + RL],
                             qual=[30] * RL, nm=0))
    recs.sort(key=lambda r: r["pos"])
    bam = str(tmp_path / "d.bam")
    bamio.write_bam(bam, [("chrD", L)], recs, block=8000)
    bamio.write_bai(bam)
    loci = [("chrD", str(p + 1)) for a in (1000, 40000) for p in range(a + 20, a + 120)]
    w = bamio.locus_weights(bam, loci)
    assert w[:100].std() == 0 and w[100:].std() == 0 and 4 < w[150] / w[50] < 16
    cuts = dist.shard_by_reads(w, 2)
    assert cuts[0] == 0 and cuts[2] == 200 and 100 < cuts[1] < 160
    depth = np.array([sum(1 for r in recs if r["pos"] <= int(p) - 1 < r["pos"] + RL) for _, p in loci])
    halves = depth[:cuts[1]].sum(), depth[cuts[1]:].sum()
    assert 0.6 < halves[0] / halves[1] < 1.6                       # (by locus count it would be 1 : 8)
    assert (bamio.locus_weights(str(tmp_path / "missing.bam"), loci) == 1).all()


def test_fasta_fetch_is_atomic_across_threads(tmp_path):
    """the decoder's helper thread and the main thread share one FastaFile; fetch() used to be an
    unlocked seek() + read() on one handle and returned the wrong bases when the two interleaved."""
    import threading
    from smcounter_amd import fasta
    rng = np.random.RandomState(11)
    seqs = {"c%d" % k: "".join(rng.choice(list("ACGT"), 20000)) for k in range(3)}
    path = str(tmp_path / "t.fa")
    with open(path, "w") as fh:
        for name, s in seqs.items():
            fh.write(">%s\n" % name)
            for i in range(0, len(s), 60):
                fh.write(s[i:i + 60] + "\n")
    fa = fasta.FastaFile(path)
    bad = []

    def work(seed):
        r = np.random.RandomState(seed)
        for _ in range(20000):
            c = "c%d" % r.randint(3)
            a = int(r.randint(0, 19900))
            b = a + int(r.randint(1, 90))
            if fa.fetch(c, a, b) != seqs[c][a:b]:
                bad.append((c, a, b))
    ts = [threading.Thread(target=work, args=(s,)) for s in (1, 2, 3)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not bad, bad[:3]


def test_deletion_followed_by_insertion_takes_the_indel_branch(tmp_path):
    """samtools' resolve_cigar2 peeks at the next operation from EVERY current operation, so the last column
    of a D block followed by I carries indel > 0, and the reference tests `indel` before `is_del`
    (smCounter.py:371,392,416): the read shows 'INS|..' there, not 'DEL'.  Both decoders."""
    ref = "ACGTTGCAAC" * 10
    fa_path = str(tmp_path / "d.fa")
    with open(fa_path, "w") as fh:
        fh.write(">chrD\n" + ref + "\n")
    seq = "ACGTTTTTGCAAC"                      # 5M 2D 3I 5M: query 5 + 3 + 5
    recs = [dict(tid=0, pos=10, qname="r%d:x:BC%d:y" % (i, i % 2), flag=0x41 if i % 2 == 0 else 0x91, mapq=60,
                 cigar=[(0, 5), (2, 2), (1, 3), (0, 5)], seq=seq, qual=[30 + i] * len(seq), nm=5) for i in range(4)]
    recs += [dict(tid=0, pos=10, qname="q%d:x:BC2:y" % i, flag=0x41, mapq=60, cigar=[(0, 4), (2, 2), (2, 1), (0, 6)],
                  seq="ACGTTGCAAC", qual=[35] * 10, nm=3) for i in range(2)]      # D followed by D: 'DEL|' at its last column
    bam = str(tmp_path / "d.bam")
    bamio.write_bam(bam, [("chrD", len(ref))], recs)
    bamio.write_bai(bam)
    fa = fasta.FastaFile(fa_path)
    loci = [("chrD", str(p)) for p in range(14, 20)]       # 1-based 16, 17 = the two deleted columns of the first shape
    py = pileup.concat([b for _, b in bamio.iter_pileup_batches(bamio.BamFile(bam), fa, loci)])
    nat = pileup.concat([b for _, b in bamio.iter_pileup_batches_native(bam, fa, loci)])
    assert py.alleles == nat.alleles
    for k, _ in bamio._COLS:
        assert np.array_equal(getattr(py, k), getattr(nat, k)), k
    i17 = loci.index(("chrD", "17"))
    s = py.locus_slice(i17)
    first_shape = py.indel[s] == 3
    assert first_shape.sum() == 4 and py.is_del[s][first_shape].all()
    keys = {py.alleles[i17][a] for a in py.allele[s][first_shape]}
    assert keys == {"INS|T|TTTG"}                    # site = seq[qpos] (qpos = first inserted base) + seq[qpos+1 : qpos+1+3], as the reference slices it
    assert (py.bq[s][first_shape] >= 30).all()       # its own quality, not the 'DEL' stand-in
    i16 = loci.index(("chrD", "16"))
    s16 = py.locus_slice(i16)
    assert "DEL" in {py.alleles[i16][a] for a in py.allele[s16][py.indel[s16] == 0]}
    # 4M 2D 1D: the last column of the first D block peeks at the second
    i16b = py.locus_slice(loci.index(("chrD", "16")))
    assert ((py.indel[i16b] == -1) & py.is_del[i16b]).sum() == 2
    P = VcParams(mtDepth=10, rpb=2.0)
    a = pileup.concat([b for _, b in bamio.iter_pileup_batches(bamio.BamFile(bam), fa, loci)])
    db_py = features.extract_features(a, P)
    db_nat = [db for _, db in bamio.iter_device_batches_native(bam, fa, loci, P)]
    assert len(db_nat) == 1
    _assert_device_batches_equal(db_py, db_nat[0])


def test_two_decoders_on_two_threads_share_the_worker_pool(tmp_path):
    """The native decoder's stages run on one process-wide worker pool (csrc/smc_bam.cpp): two handles driven from two
    Python threads (the command line's prefetch thread is one such user) take turns at it and produce the same planes as
    alone, every time."""
    import threading
    (tmp_path / "a").mkdir(); (tmp_path / "b").mkdir()
    cases = [_random_bam(tmp_path / "a", 11, True), _random_bam(tmp_path / "b", 23, True)]
    P = VcParams(mtDepth=100, rpb=3.0)

    def planes(case):
        bam, fa_path, loci = case
        fa = fasta.FastaFile(fa_path)
        return [db.meta.tobytes() + db.frag.tobytes() for _, db in
                bamio.iter_device_batches_native(bam, fa, loci, P, max_reads=3000, nthreads=4)]
    alone = [planes(c) for c in cases]
    got = [[], []]

    def run(k):
        for _ in range(8):
            got[k].append(planes(cases[k]))
    th = [threading.Thread(target=run, args=(k,)) for k in (0, 1)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for k in (0, 1):
        assert len(got[k]) == 8 and all(x == alone[k] for x in got[k])


def test_record_walk_in_pieces_equals_one_thread(tmp_path, monkeypatch):
    """collect_reads finds the record boundaries of a refill in pieces, on threads: every piece but the first guesses a record start
    and the pieces are joined only where one ended exactly at the next one's start.  A run of several MB (many pieces) with read
    names, sequences and qualities of all lengths - and read names made of bytes that look like record headers - gives the same
    alignments whatever the thread count; the pileup of a locus in its middle equals the Python decoder's."""
    rng = np.random.Generator(np.random.PCG64(5))
    L = 30000
    ref = "".join(rng.choice(list("ACGT"), size=L))
    recs = []
    for i in range(24000):
        pos = int(rng.integers(100, L - 400))
        n = int(rng.integers(20, 250))
        # a name whose bytes could be taken for the fixed fields of a record (small little-endian numbers, zeros are not allowed in names)
        junk = "".join(chr(int(c)) for c in rng.integers(1, 8, size=int(rng.integers(0, 40))))
        recs.append(dict(tid=0, pos=pos, qname="r%d%s:n%d:UMI%d:x" % (i, junk, i // 2, i % 97), flag=(0x40 if i % 2 == 0 else 0x80) | 1,
                         mapq=60, cigar=[(0, n)], seq=ref[pos:pos + n], qual=rng.integers(2, 41, size=n).astype(np.uint8).tolist(), nm=0))
    recs.sort(key=lambda r: r["pos"])
    bam = str(tmp_path / "walk.bam")
    bamio.write_bam(bam, [("chrW", L)], recs)
    bamio.write_bai(bam)
    P = VcParams()
    runs = []
    for nt, mode in ((1, "0"), (8, "0"), (8, "1"), (8, "2")):      # 1: every other piece has no guess; 2: every third guesses too far
        monkeypatch.setenv("SMC_BAM_WALK_TEST", mode)
        nb = bamio.NativeBam(bam)
        A = nb.alignments_run("chrW", 500, 29000, 1 << 40, P, nt)
        runs.append({k: np.array(A[k]).copy() if isinstance(A[k], np.ndarray) else A[k] for k in ("aln", "cig", "bq", "loc", "nl", "n_slots", "n_bc", "n_pair", "reads")})
        nb.close()
    a = runs[0]
    for b in runs[1:]:
        assert a["reads"] == b["reads"] and a["nl"] == b["nl"] and a["n_bc"] == b["n_bc"] == 97 and a["n_pair"] == b["n_pair"]
        assert len(a["aln"]) == len(b["aln"]) > 20000
        for k in ("aln", "cig", "loc"):
            assert a[k].tobytes() == b[k].tobytes(), k
        assert np.array_equal(a["bq"][:2 * int(a["aln"]["seq_off"][-1])], b["bq"][:2 * int(b["aln"]["seq_off"][-1])])
    # depth of a locus in the middle against a count over the records
    l = 15000 - 500
    want = sum(1 for r in recs if r["pos"] <= 15000 < r["pos"] + r["cigar"][0][1])
    assert int(a["loc"]["n"][l]) == want
