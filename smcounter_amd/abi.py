"""Python mirror of include/smcounter_hip.h: ctypes structs and numpy record dtypes."""
from __future__ import annotations

import ctypes

import numpy as np

from .params import VcParams

SMC_NT = 12
(T_CNT, T_FWD, T_REV, T_LOWQ, T_R1N, T_R1LE, T_R2N, T_R2BCLE, T_R2PRLE, T_CONCORD, T_DISCORD,
 T_PAD) = range(SMC_NT)

ST_OK, ST_ZERO_COVERAGE, ST_DOWNSAMPLED, ST_BAD_INPUT, ST_UNDERFLOW = 0, 1, 0x100, 0x200, 0x400

F_LM, F_LSM, F_HP, F_LOWC, F_DP, F_SB, F_LOWQ, F_R1CP, F_R2CP, F_PRIMERCP = (
    1, 2, 4, 8, 16, 32, 64, 128, 256, 512)
# FILTER names in the order filterVariants appends them (smCounter.py:187-266)
FILTER_NAMES = ((F_LM, "LM"), (F_LSM, "LSM"), (F_HP, "HP"), (F_LOWC, "LowC"), (F_DP, "DP"),
                (F_SB, "SB"), (F_LOWQ, "LowQ"), (F_R1CP, "R1CP"), (F_R2CP, "R2CP"),
                (F_PRIMERCP, "PrimerCP"))


class SmcParams(ctypes.Structure):
    _fields_ = [("min_bq", ctypes.c_int32), ("min_mq", ctypes.c_int32),
                ("mt_drop", ctypes.c_int32), ("primer_dist", ctypes.c_int32),
                ("ds", ctypes.c_int32), ("reserved", ctypes.c_int32), ("smt", ctypes.c_double),
                ("mismatch_thr", ctypes.c_double)]


class SmcBuildIn(ctypes.Structure):
    """smc_build_in (include/smcounter_hip.h): device pointers to a run's alignment arrays."""
    _fields_ = [("aln", ctypes.c_void_p), ("cig", ctypes.c_void_p), ("bq", ctypes.c_void_p),
                ("loc", ctypes.c_void_p), ("refseq", ctypes.c_void_p), ("start0", ctypes.c_int32), ("n_loci", ctypes.c_int32),
                ("n_bc", ctypes.c_int32), ("n_pair", ctypes.c_int32), ("max_depth", ctypes.c_int32), ("n_aln", ctypes.c_int32),
                ("loc_host", ctypes.c_void_p)]


DEV_ALN_DTYPE = np.dtype([("pos", "<i4"), ("end", "<i4"), ("cig_off", "<u4"), ("seq_off", "<u4"), ("n_cig", "<u2"),
                          ("oflag", "u1"), ("mapq", "u1"), ("left_sp", "<u2"), ("qalen", "<u2"), ("l_seq", "<u2"),
                          ("pad", "<u2"), ("bc_gid", "<u4"), ("pair_gid", "<u4")])
DEV_LOCUS_DTYPE = np.dtype([("w0", "<u4"), ("w1", "<u4"), ("slot_off", "<u4"), ("n", "<u4")])
assert DEV_ALN_DTYPE.itemsize == 36 and DEV_LOCUS_DTYPE.itemsize == 16


def c_params(p: VcParams) -> SmcParams:
    return SmcParams(p.minBQ, p.minMQ, p.mtDrop, p.primerDist, p.ds, 0, p.smt, float(p.mismatchThr))


CAND_DTYPE = np.dtype([
    ("allele", "<i4"), ("flt_applied", "<i4"), ("flt", "<u4"), ("vmf_lt_099", "<i4"),
    ("vdp", "<i4"), ("vmt", "<i4"), ("vsm", "<i4"), ("pad", "<i4"),
    ("tal", "<i4", (SMC_NT,)), ("pi", "<f8"),
    ("p_sb", "<f8"), ("p_r1", "<f8"), ("p_r2", "<f8"), ("p_pr", "<f8")])
assert CAND_DTYPE.itemsize == 120

ROW_DTYPE = np.dtype([
    ("status", "<i4"), ("n_touched", "<i4"),
    ("cvg", "<i4"), ("all_frag", "<i4"), ("all_mt", "<i4"), ("used_frag", "<i4"), ("used_mt", "<i4"),
    ("mt3", "<i4"), ("mt5", "<i4"), ("mt7", "<i4"), ("mt10", "<i4"),
    ("max_allele", "<i4"), ("second_allele", "<i4"), ("biallelic", "<i4"),
    ("dp", "<i4", (4,)), ("umt", "<i4", (4,)), ("vsm", "<i4", (4,)),
    ("pi", "<f8", (4,)), ("touched_mask", "<u8"),
    ("ref_tal", "<i4", (SMC_NT,)), ("cand", CAND_DTYPE, (2,))])
assert ROW_DTYPE.itemsize == 432

# smc_wire_row: the printed part of a row, what the multi-GPU gather moves (include/smcounter_hip.h)
WIRE_CAND_DTYPE = np.dtype([("allele", "<i2"), ("flags", "<u2"), ("vdp", "<i4"), ("vmt", "<i4"), ("vsm", "<i4"), ("pi", "<f8")])
WIRE_DTYPE = np.dtype([("status", "<u4"), ("cvg", "<i4"), ("all_frag", "<i4"), ("all_mt", "<i4"), ("used_frag", "<i4"),
                       ("used_mt", "<i4"), ("mt3", "<i4"), ("mt5", "<i4"), ("mt7", "<i4"), ("mt10", "<i4"),
                       ("dp", "<i4", (4,)), ("umt", "<i4", (4,)), ("vsm", "<i4", (4,)), ("pi", "<f8", (4,)),
                       ("cand", WIRE_CAND_DTYPE, (2,))])
assert WIRE_CAND_DTYPE.itemsize == 24 and WIRE_DTYPE.itemsize == 168
WIRE_BIALLELIC, WIRE_FLT_MASK, WIRE_FLT_APPLIED, WIRE_VMF_LT_099 = 0x10000, 0x3FF, 0x400, 0x800


def pack_wire(rows: np.ndarray) -> np.ndarray:
    """numpy mirror of the device's k_pack_rows (csrc/k_pack_rows.inc): ROW_DTYPE -> WIRE_DTYPE.  The GPU test checks
    the kernel against it byte for byte; the CPU (gloo) tests use it in place of the kernel."""
    w = np.zeros(len(rows), WIRE_DTYPE)
    w["status"] = (rows["status"].astype(np.uint32) & 0xFFFF) | np.where(rows["biallelic"] != 0, WIRE_BIALLELIC, 0).astype(np.uint32)
    for f in ("cvg", "all_frag", "all_mt", "used_frag", "used_mt", "mt3", "mt5", "mt7", "mt10", "dp", "umt", "vsm", "pi"):
        w[f] = rows[f]
    c, wc = rows["cand"], w["cand"]
    wc["allele"] = c["allele"].astype(np.int16)
    wc["flags"] = ((c["flt"] & WIRE_FLT_MASK) | np.where(c["flt_applied"] != 0, WIRE_FLT_APPLIED, 0)
                   | np.where(c["vmf_lt_099"] != 0, WIRE_VMF_LT_099, 0)).astype(np.uint16)
    for f in ("vdp", "vmt", "vsm", "pi"):
        wc[f] = c[f]
    return w


def unpack_wire(wire: np.ndarray) -> np.ndarray:
    """WIRE_DTYPE -> ROW_DTYPE through the library's smc_unpack_rows (host code, no GPU): every printed field restored,
    the rest zero / NaN.  rows.format_rows prints the same strings from the result as from the original rows."""
    from . import _lib
    wire = np.ascontiguousarray(wire)
    assert wire.dtype.itemsize == WIRE_DTYPE.itemsize
    out = np.zeros(len(wire), ROW_DTYPE)
    _lib.check(_lib.load().smc_unpack_rows(wire.ctypes.data, len(wire), out.ctypes.data), "smc_unpack_rows")
    return out


INT_FIELDS = ("status", "n_touched", "cvg", "all_frag", "all_mt", "used_frag", "used_mt", "mt3", "mt5",
              "mt7", "mt10", "max_allele", "second_allele", "biallelic", "dp", "umt", "vsm",
              "touched_mask", "ref_tal")
CAND_INT_FIELDS = ("allele", "flt_applied", "flt", "vmf_lt_099", "vdp", "vmt", "vsm", "tal")
CAND_F64_FIELDS = ("pi", "p_sb", "p_r1", "p_r2", "p_pr")


def _pi_of(row, allele):
    """PI of an allele if the row carries it (A,T,G,C and the candidates), else None."""
    if 0 <= allele < 4:
        return float(row["pi"][allele])
    for c in row["cand"]:
        if int(c["allele"]) == allele:
            return float(c["pi"])
    return None


def near_tie_loci(a: np.ndarray, b: np.ndarray, eps=1e-9, pi_all=None):
    """Loci where the two implementations pick maxBase / secondMaxBase differently although the
    prediction indices of the alleles they disagree on are equal to within floating-point summation
    order (|dPI| <= eps): the reference's own choice there depends on the order its dict iterates
    barcodes (smCounter.py:506, :534), so it is not pinned.  `pi_all` ([n_loci, 64], the CPU restatement's PI of
    every allele key) settles alleles the rows do not carry (a non-candidate indel key)."""
    out = set()
    d = np.nonzero((a["max_allele"] != b["max_allele"]) | (a["second_allele"] != b["second_allele"]))[0]
    for i in d:
        ok = True
        for f in ("max_allele", "second_allele"):
            x, y = int(a[f][i]), int(b[f][i])
            if x == y:
                continue
            px, py = _pi_of(a[i], x), _pi_of(a[i], y)
            if px is None:
                px = _pi_of(b[i], x)
            if py is None:
                py = _pi_of(b[i], y)
            if pi_all is not None:
                if px is None and 0 <= x < 64 and pi_all[i][x] == pi_all[i][x]:
                    px = float(pi_all[i][x])
                if py is None and 0 <= y < 64 and pi_all[i][y] == pi_all[i][y]:
                    py = float(pi_all[i][y])
            if px is None or py is None or abs(px - py) > eps * max(1.0, abs(px)):
                ok = False
        if ok:
            out.add(int(i))
    return out


def compare_rows(a: np.ndarray, b: np.ndarray, pi_tol=1e-6, p_tol=1e-6, fragile=None, pi_all=None):
    """Field-wise comparison of two row arrays: integer fields bit-exact, PI and Fisher p-values
    within tolerance.  Returns a list of human-readable mismatches (empty = equal).

    Two kinds of loci are unpinned by the reference algorithm itself and skipped for the fields
    they affect: `near_tie_loci` (order of two PI-tied alleles) and loci where `fragile[l] > 0`
    (oracle/smc_oracle.c: a barcode whose unique-maximum test hinges on rounding)."""
    bad = []
    assert a.shape == b.shape
    # SMC_ST_UNDERFLOW on either side: a barcode whose calProb products left the normal double range - the reference's own
    # numbers there are denormal rounding in its multiplication order.  Such loci are compared on the fields that do not come
    # out of calProb (the read / fragment / barcode counts) and excused on the rest, like the ties below.
    under = ((a["status"] | b["status"]) & ST_UNDERFLOW) != 0
    if under.any():
        a, b = a.copy(), b.copy()
        for r in (a, b):
            r["status"] &= ~np.int32(ST_UNDERFLOW)
        for f in ("pi", "umt", "vsm", "max_allele", "second_allele", "biallelic", "n_touched", "touched_mask", "cand", "ref_tal"):
            a[f][under] = b[f][under]
    ties = near_tie_loci(a, b, pi_all=pi_all)
    keep = np.ones(len(a), bool)
    keep[list(ties)] = False
    firm = np.ones(len(a), bool) if fragile is None else (np.asarray(fragile) == 0)
    order_dep = ("max_allele", "second_allele", "biallelic")
    mt_dep = ("umt", "vsm", "max_allele", "second_allele", "biallelic")
    fr = np.zeros(len(a), np.int64) if fragile is None else np.asarray(fragile, np.int64)
    # The tallies that only filterVariants reads (SMC_T_FWD .. SMC_T_R2PRLE, words 1-8) are part of a row where a candidate
    # goes through the filters (flt_applied); elsewhere a producer may leave them out (the GPU path does, the CPU restatement
    # fills them always).  alleleCnt is compared everywhere; the pair counts (words 9, 10: concordPairCnt / discordPairCnt)
    # everywhere for the candidates - the DP filter reads the candidate's (:207) - and nowhere for the reference allele,
    # whose pair counts nothing ever reads.
    filt_words = np.zeros(a["ref_tal"].shape[1], bool)
    filt_words[1:9] = True
    ref_dead = np.zeros(a["ref_tal"].shape[1], bool)
    ref_dead[9:11] = True
    applied = (a["cand"]["flt_applied"] != 0) & (b["cand"]["flt_applied"] != 0)            # [n, 2]
    any_applied = applied.any(axis=1)
    for f in INT_FIELDS:
        if f == "ref_tal":
            ne = ((a[f] != b[f]) & ~ref_dead[None, :] & (~filt_words[None, :] | any_applied[:, None])).any(axis=1)
        else:
            ne = (a[f] != b[f]).reshape(len(a), -1).any(axis=1)
        if f in order_dep:
            ne &= keep
        if f in ("umt", "vsm"):
            # a fragile barcode moves at most one count
            d = np.abs(a[f].astype(np.int64) - b[f].astype(np.int64)).sum(axis=1)
            ne &= d > 2 * fr
        elif f in mt_dep:
            ne &= firm
        for i in np.nonzero(ne)[0][:5]:
            bad.append("locus %d: %s %r != %r" % (i, f, a[f][i].tolist(), b[f][i].tolist()))
    ok = (a["status"] & 0xff) == ST_OK
    d = np.abs(a["pi"] - b["pi"])[ok]
    if d.size and d.max() > pi_tol:
        bad.append("pi max-abs-diff %g > %g" % (d.max(), pi_tol))
    keep &= firm
    for f in CAND_INT_FIELDS:
        if f == "tal":
            ne = ((a["cand"][f] != b["cand"][f]) & (~filt_words[None, None, :] | applied[:, :, None])).reshape(len(a), -1).any(axis=1) & keep
        else:
            ne = (a["cand"][f] != b["cand"][f]).reshape(len(a), -1).any(axis=1) & keep
        for i in np.nonzero(ne)[0][:5]:
            bad.append("locus %d: cand.%s %r != %r" % (i, f, a["cand"][f][i].tolist(),
                                                        b["cand"][f][i].tolist()))
    for f in CAND_F64_FIELDS:
        x, y = a["cand"][f][keep], b["cand"][f][keep]
        nan_mismatch = np.isnan(x) != np.isnan(y)
        if nan_mismatch.any():
            bad.append("cand.%s NaN pattern differs at %d loci" % (f, int(nan_mismatch.any(axis=1).sum())))
        tol = pi_tol if f == "pi" else p_tol
        dd = np.abs(np.where(np.isnan(x) | np.isnan(y), 0.0, x - y))
        if dd.size and dd.max() > tol:
            bad.append("cand.%s max-abs-diff %g > %g" % (f, dd.max(), tol))
    return bad


def parity_report(a: np.ndarray, b: np.ndarray, fragile=None, pi_all=None, pi_tol=1e-6, p_tol=1e-6) -> dict:
    """compare_rows plus its denominators: how many loci were compared, how many had candidate / consensus fields
    excused because the reference's own result is order-dependent there (`fragile`: a barcode whose unique-maximum
    test hinges on rounding, smCounter.py:514; `near_tie`: two alleles whose PI differ only by summation order,
    :534), and the largest PI difference seen.  "0 mismatches" means nothing without these."""
    bad = compare_rows(a, b, pi_tol, p_tol, fragile, pi_all)
    ok = ((a["status"] & 0xff) == ST_OK) & ((b["status"] & 0xff) == ST_OK)
    d = np.abs(a["pi"] - b["pi"])[ok]
    # the p-value half of the metric: how many loci reached filterVariants, how many Fisher tests ran there (a p-value that is
    # not NaN in the CPU restatement's row), and the largest difference between the two implementations' p-values
    p_max, n_tests = 0.0, 0
    for f in ("p_sb", "p_r1", "p_r2", "p_pr"):
        x, y = a["cand"][f], b["cand"][f]
        both = ~np.isnan(x) & ~np.isnan(y)
        n_tests += int((~np.isnan(y)).sum())
        if both.any():
            p_max = max(p_max, float(np.abs(x[both] - y[both]).max()))
    return {"loci": int(len(a)), "mismatches": len(bad),
            "fragile_skipped": 0 if fragile is None else int((np.asarray(fragile) > 0).sum()),
            "near_tie_skipped": len(near_tie_loci(a, b, pi_all=pi_all)),
            "pi_max_abs_diff": float(d.max()) if d.size else 0.0,
            "underflow_skipped": int((((a["status"] | b["status"]) & ST_UNDERFLOW) != 0).sum()),
            "loci_filtered": int((b["cand"]["flt_applied"] != 0).any(axis=1).sum()),
            "fisher_tests_run": n_tests, "p_max_abs_diff": p_max, "detail": bad[:3]}
