"""f4: the reference's offline helpers (ds.mt.py, ds.reads.withinMT.py, mt_depths_lod.R) re-stated."""
import argparse
import collections

import numpy as np
import pytest

import bam_fixture
from smcounter_amd import bamio
from smcounter_amd.py2compat import Py2Random, py2_dict_order
from smcounter_amd.tools import ds_mt, ds_reads_within_mt, mt_depths_lod


def _qnames(path):
    return [q for tid, q, _ in bamio.iter_raw_records(path)[1] if tid >= 0]


def test_ds_mt_keeps_whole_barcodes(tmp_path):
    case = bam_fixture.make_case(str(tmp_path), n_umi=40, frags_per_umi=4)
    out = str(tmp_path / "ds.bam")
    ns = argparse.Namespace(runPath=None, inBam=case["bam"], outBam=out, pct=0.4, seed=1234567)
    n = ds_mt.main(ns)
    src, got = _qnames(case["bam"]), _qnames(out)
    assert n == len(got) and 0 < len(got) < len(src)
    # the rule, re-derived: py2 key order of first appearance, one draw per barcode, keep r <= pct
    order = list(collections.OrderedDict((ds_mt.barcode_of(q), 1) for q in src))
    rng = Py2Random(1234567)
    kept = {bc for bc in py2_dict_order(order) if rng.random() <= 0.4}
    assert got == [q for q in src if ds_mt.barcode_of(q) in kept]          # order and multiplicity preserved
    assert 0.2 < len(kept) / len(order) < 0.6
    # records are byte-identical copies and the output is a readable BAM with the same header
    a, b = bamio.BamFile(case["bam"]), bamio.BamFile(out)
    assert a.refs == b.refs
    recs_out = b.fetch("chrQ", 0, 1000)
    assert len(recs_out) == len(got) and all(r.seq and r.cigar for r in recs_out)
    ns.pct = 1.0
    assert ds_mt.main(ns) == len(src)


def test_ds_reads_within_mt_hits_the_target(tmp_path):
    case = bam_fixture.make_case(str(tmp_path), n_umi=60, frags_per_umi=6)
    out = str(tmp_path / "ds.bam")
    ns = argparse.Namespace(runPath=None, inBam=case["bam"], outBam=out, rpb=2.5, seed=7)
    ds_reads_within_mt.main(ns)
    src, got = _qnames(case["bam"]), _qnames(out)
    per_bc_src, per_bc = collections.defaultdict(set), collections.defaultdict(set)
    for q in src:
        per_bc_src[ds_mt.barcode_of(q)].add(q)
    for q in got:
        per_bc[ds_mt.barcode_of(q)].add(q)
    assert set(per_bc) == set(per_bc_src)                                   # every barcode keeps its first read
    mean = np.mean([len(v) for v in per_bc.values()])
    assert abs(mean - 2.5) < 0.5 and mean < np.mean([len(v) for v in per_bc_src.values()])
    sel, pk = ds_reads_within_mt.select_reads(src, 2.5, 7)
    assert 0 < pk < 1 and got == [q for q in src if q in sel]


def test_lod_root_and_edge_cases(tmp_path):
    from scipy.stats import binom
    needed = mt_depths_lod.barcodes_needed(1000.0)            # cutoff 26 -> 8 barcodes
    assert needed == 8
    prev = 1.0
    for depth in (20, 100, 1000, 5000):
        lod = mt_depths_lod.find_lod(depth, needed)
        assert 0 < lod < prev                                  # deeper -> lower LOD
        assert abs(binom.cdf(needed - 1, depth, lod) - 0.05) < 0.02
        prev = lod
    assert mt_depths_lod.find_lod(4, needed) == 1.0            # fewer than 5 barcodes
    assert mt_depths_lod.find_lod(float("nan"), needed) == 1.0
    assert mt_depths_lod.find_lod(6, needed) == 1.0            # needed > depth: no sign change -> try-error -> 1
    # Brent restatement on a function with a known root
    r = mt_depths_lod.zeroin(lambda x: x * x - 0.25, 0.0, 1.0, -0.25, 0.75, 1e-10)
    assert abs(r - 0.5) < 1e-9
    fin = tmp_path / "depths.txt"
    fin.write_text("chr1|100|101|1000\nchr1|101|102|3\nchr2|5|6|NA\nchr2|6|7|250\n")
    mt_depths_lod.main(["1000", str(fin), str(tmp_path / "lod.bedgraph")])
    lines = (tmp_path / "lod.bedgraph").read_text().splitlines()
    assert [l.split("\t")[:3] for l in lines] == [["chr1", "100", "101"], ["chr1", "101", "102"], ["chr2", "5", "6"],
                                                   ["chr2", "6", "7"]]
    assert lines[1].endswith("\t1") and lines[2].endswith("\t1") and 0 < float(lines[0].split("\t")[3]) < 0.05
    q = (tmp_path / "lod.bedgraph.quantiles.txt").read_text().splitlines()
    assert [x.split("|")[0] for x in q] == ["1%", "5%", "10%", "50%", "90%", "95%", "99%"]


def test_down_samplers_against_what_the_reference_scripts_wrote():
    """tests/golden/tools_ds.npz (make_tools_golden.py, build container): the read names ds.mt.py and ds.reads.withinMT.py
    themselves wrote - run through lib2to3 with a stub pysam and a py2-ordered dict - for three inputs (5, 40 and 300 barcodes:
    below and beyond the resizes of a py2 dict) and three (seed, parameter) pairs each.  The tools' selection rules
    (tools/ds_mt.select_barcodes, tools/ds_reads_within_mt.select_reads: py2 key order + Py2Random) must pick the same
    records, in the same order."""
    import json
    import os
    from conftest import ROOT
    z = np.load(os.path.join(ROOT, "tests", "golden", "tools_ds.npz"))
    cases = json.loads(bytes(z["meta"]).decode())
    n_runs = 0
    for case in cases:
        q = case["qnames"]
        for run in case["runs"]:
            if run["script"] == "ds.mt.py":
                kept = ds_mt.select_barcodes(q, run["pct"], run["seed"])
                got = [x for x in q if ds_mt.barcode_of(x) in kept]
            else:
                sel, _ = ds_reads_within_mt.select_reads(q, run["rpb"], run["seed"])
                got = [x for x in q if x in sel]
            assert got == run["written"], (run["script"], run["seed"], len(got), len(run["written"]))
            n_runs += 1
    assert n_runs == 18
