"""The pure-Python port (oracle/vc_port.py, the CPU baseline of bench.py) against the C restatement on the
golden inputs: same integer fields, PI within 1e-9, same FILTER bits; and its process-pool driver."""
import os

import numpy as np
import pytest

from conftest import golden_files, load_golden
from smcounter_amd import abi, synth

import oracle_lib
import vc_port


def _check(rows_py, R):
    for l, r in enumerate(rows_py):
        if R["status"][l] & abi.ST_UNDERFLOW:
            continue        # (a barcode beyond the double range: the C restatement flags it, the port does not look)
        assert r["status"] == R["status"][l] and r["cvg"] == R["cvg"][l]
        assert r["all_mt"] == R["all_mt"][l] and r["all_frag"] == R["all_frag"][l]
        assert r["dp"] == R["dp"][l].tolist()
        if r["status"] & 0xff:
            continue
        for k in ("used_mt", "used_frag", "mt3", "mt5", "mt7", "mt10", "n_touched", "biallelic"):
            assert r[k] == R[k][l], (l, k)
        assert r["umt"] == R["umt"][l].tolist() and r["vsm"] == R["vsm"][l].tolist()
        assert np.abs(np.array(r["pi"]) - R["pi"][l]).max() <= 1e-9
        for ci, key in ((0, "cand0"), (1, "cand1")):
            c, C = r[key], R["cand"][l][ci]
            if c is None:
                assert C["allele"] == -1
                continue
            if abs(c["pi"] - C["pi"]) > 1e-9:       # order of two PI-tied alleles: not pinned
                continue
            assert c["allele"] == C["allele"], (l, key)
            for k in ("vdp", "vmt", "vsm", "flt_applied", "flt", "vmf_lt_099"):
                assert c[k] == C[k], (l, key, k)
            for a, b in zip(c["p"], (C["p_sb"], C["p_r1"], C["p_r2"], C["p_pr"])):
                assert (np.isnan(a) and np.isnan(b)) or abs(a - b) <= 1e-9


@pytest.mark.parametrize("path", [p for p in golden_files() if "stress4" not in p], ids=os.path.basename)
def test_python_port_matches_c_restatement(path):
    pb, db, P, refp, expected = load_golden(path)
    R = oracle_lib.call_batch(db, abi.c_params(P), abi.ROW_DTYPE)
    n = min(db.n_loci, 60)
    _check(vc_port.call_batch(db, P, n_cpu=1, loci=range(n)), R[:n])


def test_pool_driver_keeps_order():
    cfg = synth.CONFIGS["C2"]
    P = synth.params_for(cfg)
    db = synth.generate_native(cfg, 0, 24, P, nthreads=1)
    R = oracle_lib.call_batch(db, abi.c_params(P), abi.ROW_DTYPE)
    _check(vc_port.call_batch(db, P, n_cpu=2), R)


def test_the_object_adapter_port_gives_the_c_restatements_rows():
    """oracle/vc_port_objects.py - the restatement of smCounter.py:316-479 over pysam-like objects (read names split and joined,
    the tag list walked for NM, the CIGAR walked for indels and clips, string allele keys: the work the integer-plane port skips,
    SURVEY.md 8d) - on a run of synthetic alignments with variants, indels and clips: every integer column, the filter bits of the
    candidate and the prediction indices equal oracle/aln_planes.c + oracle/smc_oracle.c's on the same alignments."""
    import oracle_lib
    import vc_port_objects as vo
    from smcounter_amd import abi, synth
    cfg = synth.SynthConfig("OA", 48, 40, 30, 778, p_overlap=0.5, alt_locus_frac=0.3, alt_af=0.2)
    P = synth.params_for(cfg)
    A = synth.generate_alignments(cfg, 48, P, p_del_aln=0.03, p_ins_aln=0.02, p_clip=0.05, nthreads=2)
    objs = vo.alignment_objects(A, P.mismatchThr)
    db = oracle_lib.aln_planes(A, P, 0, 48, n_threads=2)
    want = oracle_lib.call_batch(db, abi.c_params(P), abi.ROW_DTYPE)
    ref = synth.aln_ref_fetch(int(A["start0"]), int(A["start0"]) + 48 + 64)
    n_flt = n_indel_keys = 0
    for l in range(48):
        col = vo.pileup_objects(A, l, objs)
        row = vo.vc_locus_objects(col, ref[l], (lambda n, l=l: ref[l + 1:l + 1 + n]), P.minBQ, P.minMQ, P.mismatchThr, P.mtDrop,
                                  P.primerDist, P.ds, P.smt)
        w = want[l]
        for k in ("cvg", "all_mt", "all_frag", "used_mt", "used_frag", "mt3", "mt5", "mt7", "mt10"):
            assert row[k] == w[k], (l, k)
        assert row["dp"] == list(w["dp"]) and row["umt"] == list(w["umt"]) and row["vsm"] == list(w["vsm"])
        assert np.allclose(row["pi"], w["pi"], atol=1e-6)
        c0, wc = row["cand0"], w["cand"][0]
        assert (c0["flt_applied"], c0["flt"], c0["vdp"], c0["vmt"]) == (wc["flt_applied"], wc["flt"], wc["vdp"], wc["vmt"])
        n_flt += c0["flt_applied"]
        n_indel_keys += len(row["alleles"]) > 6
    assert n_flt >= 5 and n_indel_keys >= 5
