"""f2: BGZF/BAM reader and single-position pileup against a naive expansion of the same reads."""
import numpy as np
import pytest

import bam_fixture
from smcounter_amd import bamio, bedops, fasta, features, pileup, rows, abi
from smcounter_amd.params import VcParams

import oracle_lib


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
