"""f1: writers, threshold, repeat post-filter, BED expansion (smCounter.py:674-680, :696-901)."""
import os

import pytest

from conftest import GOLDEN
from smcounter_amd import bedops, postfilter, writers

EX = os.path.join(GOLDEN, "example")


def test_writers_reproduce_reference_cut_files(tmp_path):
    rows = open(os.path.join(EX, "example.smCounter.all.txt")).read().split("\n")[1:-1]
    assert len(rows) == 2000
    thr = writers.pi_threshold(3612, 0)
    assert thr == 58
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        writers.write_outputs("example", rows, thr)
        for name in ("example.smCounter.all.txt", "example.smCounter.cut.txt", "example.smCounter.cut.vcf"):
            assert open(name).read() == open(os.path.join(EX, name)).read(), name
    finally:
        os.chdir(cwd)


def test_threshold_rule():
    assert writers.pi_threshold(3612) == 58 and writers.pi_threshold(100) == 16
    assert writers.pi_threshold(3612, 30) == 30


def test_bed_expansion_and_ops(tmp_path):
    bed = tmp_path / "t.bed"
    bed.write_text("track name=x\nchr1\t10\t13\nchr2\t5\t6\textra\n")
    assert bedops.expand_loci(str(bed)) == [("chr1", "11"), ("chr1", "12"), ("chr1", "13"), ("chr2", "6")]
    iv = [("c", 1, 5, "Simple_repeat"), ("c", 5, 9, "Low_complexity"), ("c", 20, 30, "Satellite"), ("b", 3, 4, "x")]
    m = bedops.merge(bedops.sort_bed(iv), distinct_names=True)
    assert m == [("b", 3, 4, "x"), ("c", 1, 9, "Low_complexity,Simple_repeat"), ("c", 20, 30, "Satellite")]
    assert bedops.intersect([("c", 0, 25, "n")], [("c", 3, 6, ""), ("c", 22, 40, "")]) == \
        [("c", 3, 6, "n"), ("c", 22, 25, "n")]


def test_repeat_post_filter(tmp_path):
    target = tmp_path / "target.bed"
    target.write_text("chr1\t0\t1000\n")
    trf = tmp_path / "trf.bed"
    trf.write_text("chr1\t100\t200\tx\n")
    rmsk = tmp_path / "rm.bed"
    rmsk.write_text("chr1\t150\t160\tSimple_repeat\nchr1\t155\t170\tLow_complexity\nchr1\t500\t510\tL1\n")
    t, r = postfilter.load_repeat_regions(str(target), str(trf), str(rmsk))

    def row(pos, pi, alt="A", flt=";", vmf="0.5"):
        f = [""] * 45
        f[0], f[1], f[3], f[10], f[14], f[44] = "chr1", str(pos), alt, str(pi), vmf, flt
        return "\t".join(f)
    out = postfilter.apply_repeat_filters(
        [row(150, 10.0), row(100, 10.0), row(101, 10.0), row(156, 6.0, flt=";LM;"), row(505, 9.9),
         row(156, 4.99), row(156, 50.0, alt="DEL"), "chr1\t7\tA" + "\t" * 41 + "\tZero_Coverage"], t, r)
    last = [o.split("\t")[-1] for o in out]
    assert last == ["RepT", "PASS", "RepT", "LM;RepT;LowC;RepS", "Other_Repeat", "PASS", "PASS", "Zero_Coverage"]


def test_post_filter_fast_path_equals_the_plain_statement(tmp_path):
    """apply_repeat_filters' shortcut for unchanged rows against the row-by-row statement, on the reference's own
    2000 example rows (FILTER turned back into the raw ';' form) and on malformed / short rows."""
    rows = open(os.path.join(EX, "example.smCounter.all.txt")).read().split("\n")[1:-1]
    raw = []
    for r in rows:
        f = r.split("\t")
        if f[-1] != "Zero_Coverage":
            f[-1] = ";" if f[-1] == "PASS" else ";" + f[-1] + ";"
        raw.append("\t".join(f))
    raw += ["chr1\t5\tA\t;", "x\t;", "chr1\tNaN\tA\tT\tSNP" + "\t1" * 39 + "\t;", "chr1\t9\tA\tT\tSNP\t1\t1\t1\t1\t1\tabc"
            + "\t1" * 33 + "\t;", "chr1\t9\tA\tT\tSNP\t1\t1\t1\t1\t1\t7.5\t1\t1\t1\t" + "\t1" * 29 + "\t;"]
    trf = {"chr1": [(0, 10**9, "RepT;")]}
    rm = {"chr1": [(115250000, 115260000, "LowC;")]}
    for t, r in (({}, {}), (trf, rm)):
        want = []
        for x in raw:
            try:
                want.append(postfilter._apply_one(x, t, r))
            except Exception as e:                   # malformed rows fail the same way on both paths
                want.append(type(e))
        got = []
        for x in raw:
            try:
                got.extend(postfilter.apply_repeat_filters([x], t, r))
            except Exception as e:
                got.append(type(e))
        assert got == want
    assert postfilter.apply_repeat_filters(raw[:2000], {}, {}) == rows
