"""oracle/aln_planes.c - the CPU restatement of the per-pileup-read half of vc() FROM ALIGNMENTS (what bench.py checks the
device plane builder against at full size) - pinned against the host builder smc_bam_planes, which the reference-generated BAM
fixtures pin to the reference's own row strings (tests/test_bam_golden.py):
  * the three BAM fixtures: decoder -> alignments -> aln_planes  ==  decoder -> smc_bam_planes  (up to barcode / fragment
    numbering; both number by first appearance at the locus, so in fact word for word), and its rows through the CPU
    restatement print the reference's strings;
  * synthetic runs (bench.py's from_alignments input) written as BAMs: the same equality."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT
from smcounter_amd import abi, bamio, fasta, planecheck, rows, synth
from smcounter_amd.params import VcParams

sys.path.insert(0, os.path.join(ROOT, "tests"))


def _runs(loci):
    out, cur = [], [loci[0]]
    for a, b in zip(loci, loci[1:]):
        if b[0] == a[0] and int(b[1]) == int(a[1]) + 1:
            cur.append(b)
        else:
            out.append(cur); cur = [b]
    out.append(cur)
    return out


@pytest.mark.parametrize("name", ("bam_cigars", "bam_deep", "bam_unflagged"))
def test_alignments_to_planes_on_the_reference_generated_bams(name, tmp_path):
    import oracle_lib
    import test_bam_golden as G
    bam, fa, loci, P, expected = G.load_case(name, tmp_path)
    nb = bamio.NativeBam(bam)
    text = []
    for run in _runs(loci):
        chrom, lo, hi = run[0][0], int(run[0][1]) - 1, int(run[-1][1])
        A = nb.alignments_run(chrom, lo, hi, 1 << 40, P, 2)
        assert A["nl"] == hi - lo and (A["status"] & ~1) == 0      # (bit 0: an unflagged alignment - restated too)
        refseq = fa.fetch(chrom, lo, hi).upper()
        A.update(start0=lo, refseq=refseq.encode().ljust(hi - lo, b"N"))
        db = oracle_lib.aln_planes(A, P, n_threads=2)
        nl, planes, us, lc, tables = nb.planes_run(chrom, lo, hi, 1 << 40, P, refseq, 2, fa)
        # the host builder's batch of the same run, laid out as a DeviceBatch
        from smcounter_amd.features import DeviceBatch
        hb = DeviceBatch(loci=lc, meta=planes[0], umi=planes[1], frag=planes[2], dist=planes[3], umi_start=us, chrom=[chrom] * nl,
                         pos=np.arange(lo + 1, hi + 1, dtype=np.int64), ref=list(refseq), alleles=tables)
        db.chrom, db.alleles = hb.chrom, hb.alleles                       # (strings: the C side has none)
        db.loci["read_off4"] = hb.loci["read_off4"]                         # (both start their run at slot 0; checked by the planes)
        assert planecheck.differences(db, hb) == []
        got = oracle_lib.call_batch(db, abi.c_params(P), abi.ROW_DTYPE)
        text += rows.format_rows(got, db, P, fa)
    nb.close()
    G.assert_rows(text, expected, "decoder -> alignments -> oracle/aln_planes.c -> smc_oracle.c")


@pytest.mark.parametrize("cfg_name,n_loci", [("C2", 300), ("C3", 140), ("X2", 100)])
def test_alignments_to_planes_on_synthetic_runs(cfg_name, n_loci, tmp_path):
    import oracle_lib
    cfg = synth.CONFIGS[cfg_name]
    P = synth.params_for(cfg)
    A = synth.generate_alignments(cfg, n_loci, P, p_ins_aln=0.03, p_del_aln=0.03)
    db = oracle_lib.aln_planes(A, P, n_threads=3)
    bam, fa_path = str(tmp_path / "s.bam"), str(tmp_path / "s.fa")
    chrom, p0, p1 = synth.alignments_to_bam(A, bam, 0, n_loci, fa_path)
    fa = fasta.FastaFile(fa_path)
    loci = [(chrom, str(p)) for p in range(p0, p1 + 1)]
    host = [b for _, b in bamio.iter_device_batches_native(bam, fa, loci, P, max_reads=8_000_000)]
    assert len(host) == 1
    hb = host[0]
    db.alleles = hb.alleles
    assert planecheck.differences(db, hb) == []
    assert any(len(t) > 6 for t in hb.alleles)
    # a sub-range of the run gives the same loci
    sub = oracle_lib.aln_planes(A, P, 37, 91)
    want = oracle_lib.call_batch(db, abi.c_params(P), abi.ROW_DTYPE)[37:91]
    got = oracle_lib.call_batch(sub, abi.c_params(P), abi.ROW_DTYPE)
    assert got.tobytes() == want.tobytes()


def test_variants_under_synthetic_alignments(tmp_path):
    """smc_synth_alignments with alt_locus_frac / alt_af (round 5): the run does not depend on the thread count; variant sites
    carry the transition in about alt_af of the molecules; the reads of a molecule agree; such loci reach filterVariants in the
    oracle; the decoder's path sees the same batch (the run written as a BAM: its NM makes the same mismatch-ok bits)."""
    import oracle_lib
    cfg = synth.CONFIGS["X3"]
    P = synth.params_for(cfg)
    n = 320
    A = synth.generate_alignments(cfg, n, P, nthreads=5)
    B = synth.generate_alignments(cfg, n, P, nthreads=2)
    assert all(np.array_equal(A[k], B[k]) for k in ("aln", "cig", "bq", "loc"))
    plain = synth.generate_alignments(synth.CONFIGS["C3"], n, synth.params_for(synth.CONFIGS["C3"]), nthreads=3)
    db = oracle_lib.aln_planes(A, P, 0, n, n_threads=4)
    want = oracle_lib.call_batch_mt(db, abi.c_params(P), abi.ROW_DTYPE, 4)
    applied = (want["cand"]["flt_applied"] != 0).any(axis=1)
    assert 0.15 * n < applied.sum() < 0.5 * n                       # alt_locus_frac 0.3
    l = int(np.flatnonzero(applied)[0])
    c = want["cand"][l][0]
    assert 0.02 < c["vmt"] / max(1, want["used_mt"][l]) < 0.3        # alt_af 0.1 of the barcodes
    # the reference allele dominates everywhere, and a run without variants has no filtered locus
    db0 = oracle_lib.aln_planes(plain, synth.params_for(synth.CONFIGS["C3"]), 0, n, n_threads=4)
    w0 = oracle_lib.call_batch_mt(db0, abi.c_params(synth.params_for(synth.CONFIGS["C3"])), abi.ROW_DTYPE, 4)
    assert not (w0["cand"]["flt_applied"] != 0).any()
    # the mismatch count a mapper would report includes the variant bases: fewer alignments pass the 6 / 100 b threshold
    assert ((A["aln"]["oflag"] & 16) != 0).mean() < ((plain["aln"]["oflag"] & 16) != 0).mean() - 0.02


def test_traffic_records_are_tied_to_the_library_they_were_measured_on(tmp_path, monkeypatch):
    """bench_fa.traffic_record: a PMC record of profiles/traffic.json is handed out only when its lib_sha16 is the hash of the
    library that runs now."""
    import json
    import bench_fa
    have = bench_fa.lib_sha16()
    assert len(have) == 16
    t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    assert "lib_sha16" in t["fa:C3:200000"] and "lib_sha16" in t["C3:200000"]
    rec, why = bench_fa.traffic_record("fa:C3:200000")
    assert (rec is not None) == (t["fa:C3:200000"]["lib_sha16"] == have)
    assert (why is None) == (rec is not None)
    rec, why = bench_fa.traffic_record("no:such:key")
    assert rec is None and "no PMC record" in why
