"""Numeric locus record -> the TAB-joined row vc() returns (reference: smCounter.py:545-600,
convertToVcf :103-117, the FILTER string of filterVariants :184-269).

The device fills one `smc_row` per locus (include/smcounter_hip.h).  What is left for the host is
what needs strings or the reference sequence: allele-key -> (REF, ALT, TYPE), the HP / LowC flags,
the bi-allelic decision (which depends on those flags, :567-573), rounding and printing."""
from __future__ import annotations

from . import abi
from .hpregion import is_hp_or_lowcomp
from .params import VcParams
from .py2compat import py2_round, py2_str

HEADER_ALL = ("CHROM", "POS", "REF", "ALT", "TYPE", "DP", "FR", "MT", "UFR", "UMT", "PI", "VDP", "VAF",
              "VMT", "VMF", "VSM", "DP_A", "DP_T", "DP_G", "DP_C", "AF_A", "AF_T", "AF_G", "AF_C",
              "MT_3RPM", "MT_5RPM", "MT_7RPM", "MT_10RPM", "UMT_A", "UMT_T", "UMT_G", "UMT_C", "UMF_A",
              "UMF_T", "UMF_G", "UMF_C", "VSM_A", "VSM_T", "VSM_G", "VSM_C", "PI_A", "PI_T", "PI_G",
              "PI_C", "FILTER")


class RowError(Exception):
    pass


def convert_to_vcf(orig_ref: str, orig_alt: str):
    vtype, ref, alt = ".", orig_ref, orig_alt
    if len(orig_alt) == 1:
        vtype = "SNP"
    elif orig_alt == "DEL":
        vtype = "SDEL"
    else:
        vals = orig_alt.split("|")
        if vals[0] in ("DEL", "INS"):
            vtype, ref, alt = "INDEL", vals[1], vals[2]
    return ref, alt, vtype


def filter_string(cand, chrom, pos, ref, alt, params: VcParams, refprov) -> str:
    """';' + 'NAME;' per flag, in filterVariants' order.  `cand` is one smc_cand record."""
    if not cand["flt_applied"]:
        return ";"
    bits = int(cand["flt"])
    homop, lowcomp = is_hp_or_lowcomp(chrom, pos, params.hpLen, ref, alt, refprov)
    if homop and cand["vmf_lt_099"]:
        bits |= abi.F_HP
    if lowcomp and cand["vmf_lt_099"]:
        bits |= abi.F_LOWC
    return ";" + "".join(name + ";" for bit, name in abi.FILTER_NAMES if bits & bit)


def _head_and_filter(row, chrom: str, pos: str, orig_ref: str, alleles, params: VcParams, refprov):
    """CHROM..TYPE, FILTER and which candidate the numeric columns are printed from (:541-573)."""
    c0 = row["cand"][0]
    orig_alt = alleles[int(c0["allele"])]
    ref, alt, vtype = convert_to_vcf(orig_ref, orig_alt)
    fltr = filter_string(c0, chrom, pos, ref, alt, params, refprov)
    chosen = 0
    if row["biallelic"]:
        c1 = row["cand"][1]
        ref2, alt2, vtype2 = convert_to_vcf(orig_ref, alleles[int(c1["allele"])])
        fltr2 = filter_string(c1, chrom, pos, ref2, alt2, params, refprov)
        if fltr == ";" and fltr2 == ";":
            alt = alt + "," + alt2
            vtype = vtype.lower() + "," + vtype2.lower()
        elif fltr != ";" and fltr2 == ";":
            alt, fltr, chosen = alt2, fltr2, 1
    return ref, alt, vtype, fltr, chosen


def format_row(row, chrom: str, pos, orig_ref: str, alleles, params: VcParams, refprov) -> str:
    pos = str(int(pos))
    st = int(row["status"])
    if st & abi.ST_BAD_INPUT:
        raise RowError("locus %s:%s: batch violates the layout contract (ids out of range)" % (chrom, pos))
    if (st & 0xff) == abi.ST_ZERO_COVERAGE:
        return "\t".join([chrom, pos, orig_ref] + [""] * 41 + ["Zero_Coverage"])
    ref, alt, vtype, fltr, ci = _head_and_filter(row, chrom, pos, orig_ref, alleles, params, refprov)
    chosen = row["cand"][ci]
    cvg, used = int(row["cvg"]), int(row["used_mt"])
    dp = [int(v) for v in row["dp"]]
    umt = [int(v) for v in row["umt"]]
    vsm = [int(v) for v in row["vsm"]]
    vdp, vmt = int(chosen["vdp"]), int(chosen["vmt"])
    out = [chrom, pos, ref, alt, vtype, cvg, int(row["all_frag"]), int(row["all_mt"]),
           int(row["used_frag"]), used, py2_round(float(chosen["pi"]), 2), vdp,
           py2_round(1.0 * vdp / cvg, 4), vmt, py2_round(1.0 * vmt / used, 4), int(chosen["vsm"])]
    out += dp + [py2_round(1.0 * v / cvg, 4) for v in dp]
    out += [int(row["mt3"]), int(row["mt5"]), int(row["mt7"]), int(row["mt10"])]
    out += umt + [py2_round(1.0 * v / used, 4) for v in umt]
    out += vsm
    out += [py2_round(float(v), 2) for v in row["pi"]]
    out.append(fltr)
    return "\t".join(py2_str(x) for x in out)


def _fmt_float(x: float) -> str:
    # py2 str(float) of an already rounded value ('%.12g' + '.0' for integers), inlined for the batch path
    s = "%.12g" % x
    if "." not in s and "e" not in s and "n" not in s:
        s += ".0"
    return s


_ROWFMT = None


def _rowfmt():
    """libsmc_rowfmt.so (csrc/smc_rowfmt.cpp, include/smcounter_host.h), built on first use."""
    global _ROWFMT
    if _ROWFMT is None:
        import ctypes as C
        from . import build
        L = C.CDLL(build.build_rowfmt())
        L.smc_rowfmt_stride.restype = C.c_int
        L.smc_format_tails.restype = C.c_int64
        L.smc_format_tails.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        L.smc_rowfmt_line_stride.restype = C.c_int
        L.smc_rowfmt_line_stride.argtypes = [C.c_int]
        L.smc_format_lines.restype = C.c_int64
        L.smc_format_lines.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_char_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        _ROWFMT = L
    return _ROWFMT


def format_tails(rows, chosen=None):
    """Columns DP..PI_C of every row, printed by the native formatter (py2 rounding and float printing); '' for the
    rows it leaves to the caller (status != 0, chosen < 0, out-of-range values)."""
    import numpy as np
    n = len(rows)
    if n == 0:
        return []
    L = _rowfmt()
    rows = np.ascontiguousarray(rows)
    assert rows.dtype.itemsize == 432
    buf = np.empty(n * L.smc_rowfmt_stride(), np.uint8)
    ch = None if chosen is None else np.ascontiguousarray(chosen, np.int8)
    nb = L.smc_format_tails(rows.ctypes.data, None if ch is None else ch.ctypes.data, n, buf.ctypes.data)
    return buf[:nb - 1].tobytes().decode("ascii").split("\n")


class RowLines(list):
    """The 45-field strings of a batch, plus what the native printer knows about them: `pred` - per row
    int(float(PI column)), the number the repeat filters and the writers compare (smCounter.py:757, :838), or PRED_NONE
    for a row it did not print - so that the stages behind vc() need not split every string to find it again."""
    pred = None


PRED_NONE = -(1 << 31)
_SNP_LETTER = None


def format_lines(rows, chosen, chrom, pos, ref, alt):
    """Native printer, whole lines (smc_format_lines): rows with alt[i] != 0 come back as the complete string vc() returns for
    a locus no filter applies to, the others as their 39 numeric columns ('' = left to the caller).  -> (lines, pred)"""
    import ctypes as C
    import numpy as np
    n = len(rows)
    L = _rowfmt()
    rows = np.ascontiguousarray(rows)
    assert rows.dtype.itemsize == 432
    names, ids = [], {}
    chrom_id = np.empty(n, np.int32)
    if n and isinstance(chrom, list) and chrom.count(chrom[0]) == n:      # (one chromosome: the usual batch)
        names = [chrom[0]]; chrom_id[:] = 0
    else:
        for i, c in enumerate(chrom):
            k = ids.get(c)
            if k is None:
                k = ids[c] = len(names); names.append(c)
            chrom_id[i] = k
    enc = [c.encode() for c in names]
    off = np.zeros(len(enc) + 1, np.int32)
    off[1:] = np.cumsum([len(e) for e in enc])
    mx = max((len(e) for e in enc), default=0)
    buf = np.empty(n * L.smc_rowfmt_line_stride(mx), np.uint8)
    pred = np.empty(n, np.int32)
    ch = np.ascontiguousarray(chosen, np.int8)
    posa = np.ascontiguousarray(pos, np.int64)
    import os
    nb = L.smc_format_lines(rows.ctypes.data, ch.ctypes.data, n, b"".join(enc), off.ctypes.data, chrom_id.ctypes.data, posa.ctypes.data,
                            ref.ctypes.data, alt.ctypes.data, mx, min(16, len(os.sched_getaffinity(0))), buf.ctypes.data, pred.ctypes.data)
    return buf[:nb - 1].tobytes().decode("ascii").split("\n"), pred


def format_rows(rows, db, params: VcParams, refprov, native: bool = True):
    """`format_row` over a batch.  The numeric columns (39 of the 45) of every callable locus are printed by the
    native formatter in one call (`format_tails`); Python adds CHROM..TYPE and FILTER.  Loci whose candidate goes
    through a filter or is bi-allelic (few) first get their FILTER / bi-allelic decision here (it needs the
    reference sequence), which also selects the candidate the numeric columns are printed from.  Same strings as
    `format_row` either way (tests/test_host_logic.py compares them on the golden loci and on random rows);
    `native=False` keeps everything in Python."""
    import numpy as np
    n = len(rows)
    if n == 0:
        return []
    status = rows["status"]
    cand0 = rows["cand"][:, 0]
    callable_ = ((status & 0xff) == 0) & ((status & abi.ST_BAD_INPUT) == 0)
    simple = callable_ & (rows["biallelic"] == 0) & (cand0["flt_applied"] == 0)
    if not native:
        return _format_rows_py(rows, db, params, refprov, simple.tolist())
    chosen = np.where(callable_, 0, -1).astype(np.int8)
    heads = {}
    chrom, refs, alleles = db.chrom, db.ref, db.alleles
    for l in np.flatnonzero(callable_ & ~simple).tolist():
        ps = str(int(db.pos[l]))
        ref, alt, vtype, fltr, ci = _head_and_filter(rows[l], chrom[l], ps, refs[l], alleles[l], params, refprov)
        chosen[l] = ci
        heads[l] = ("\t".join((chrom[l], ps, ref, alt, vtype)), fltr)
    # A simple locus whose candidate is one of A, T, G, C, N (allele ids 0-4, fixed) on a one-letter reference base is a
    # "SNP" line (convert_to_vcf :103-117 with equal lengths) that needs no string work at all: printed whole.
    c_allele = cand0["allele"]
    ref_b = _ref_letters(refs, n)
    alt_b = np.where(simple & (c_allele >= 0) & (c_allele <= 4) & (ref_b != 0),
                     np.frombuffer(b"ATGCN\0\0\0", np.uint8)[np.clip(c_allele, 0, 5)], 0).astype(np.uint8)
    lines, pred = format_lines(rows, chosen, chrom, db.pos, ref_b, alt_b)
    out = RowLines(lines)
    out.pred = pred
    # what is left: rows the printer declined, simple loci with an indel / 'DEL' candidate, loci with filters
    todo = np.flatnonzero((alt_b == 0) | (pred == PRED_NONE)).tolist()
    if todo:
        c_allele = c_allele.tolist()
        simple = simple.tolist()
        for l in todo:
            t = lines[l]
            if not t:                                   # Zero_Coverage / bad input / out of the native printer's range
                out[l] = format_row(rows[l], chrom[l], db.pos[l], refs[l], alleles[l], params, refprov)
            elif simple[l]:
                ref, alt, vtype = convert_to_vcf(refs[l], alleles[l][c_allele[l]])
                out[l] = "\t".join((chrom[l], str(int(db.pos[l])), ref, alt, vtype, t, ";"))
            else:
                h, fltr = heads[l]
                out[l] = h + "\t" + t + "\t" + fltr
                if fltr != ";":
                    pred[l] = PRED_NONE                 # (the post-filter looks at this row itself)
    return out


def _ref_letters(refs, n):
    """The loci's reference bases as bytes (0 where it is not exactly one ASCII letter)."""
    import numpy as np
    if isinstance(refs, str) and len(refs) == n and refs.isascii():
        return np.frombuffer(refs.encode(), np.uint8).copy()
    s = "".join(refs)
    if len(s) == n and s.isascii() and max(map(len, refs), default=1) == 1:
        return np.frombuffer(s.encode(), np.uint8).copy()
    return np.array([ord(r) if len(r) == 1 and ord(r) < 128 else 0 for r in refs], np.uint8)


def _format_rows_py(rows, db, params: VcParams, refprov, simple):
    """The batch formatter without the native library: column-wise for the simple loci - the structured array is
    unpacked once into Python lists instead of one numpy field access per value."""
    n = len(rows)
    cand0 = rows["cand"][:, 0]
    cols = {k: rows[k].tolist() for k in ("cvg", "all_frag", "all_mt", "used_frag", "used_mt", "mt3", "mt5", "mt7", "mt10",
                                          "dp", "umt", "vsm", "pi")}
    c_allele, c_pi, c_vdp, c_vmt, c_vsm = (cand0[k].tolist() for k in ("allele", "pi", "vdp", "vmt", "vsm"))
    rnd, ff = py2_round, _fmt_float
    out = [None] * n
    # (the loop allocates a few dozen short-lived strings per locus and no cycles: the cyclic collector would only
    # rescan the rows already built, again and again)
    import gc
    gc_was_on = gc.isenabled()
    gc.disable()
    try:
        _format_simple_rows(out, range(n), simple, rows, db, params, refprov, cols, c_allele, c_pi, c_vdp, c_vmt, c_vsm, rnd, ff)
    finally:
        if gc_was_on:
            gc.enable()
    return out


def _format_simple_rows(out, idx, simple, rows, db, params, refprov, cols, c_allele, c_pi, c_vdp, c_vmt, c_vsm, rnd, ff):
    for l in idx:
        if not simple[l]:
            out[l] = format_row(rows[l], db.chrom[l], db.pos[l], db.ref[l], db.alleles[l], params, refprov)
            continue
        ref, alt, vtype = convert_to_vcf(db.ref[l], db.alleles[l][c_allele[l]])
        cvg, used = cols["cvg"][l], cols["used_mt"][l]
        dp, umt, vsm, pi = cols["dp"][l], cols["umt"][l], cols["vsm"][l], cols["pi"][l]
        vdp, vmt = c_vdp[l], c_vmt[l]
        f = [db.chrom[l], str(int(db.pos[l])), ref, alt, vtype, str(cvg), str(cols["all_frag"][l]), str(cols["all_mt"][l]),
             str(cols["used_frag"][l]), str(used), ff(rnd(c_pi[l], 2)), str(vdp), ff(rnd(1.0 * vdp / cvg, 4)), str(vmt),
             ff(rnd(1.0 * vmt / used, 4)), str(c_vsm[l]),
             str(dp[0]), str(dp[1]), str(dp[2]), str(dp[3]),
             ff(rnd(1.0 * dp[0] / cvg, 4)), ff(rnd(1.0 * dp[1] / cvg, 4)), ff(rnd(1.0 * dp[2] / cvg, 4)),
             ff(rnd(1.0 * dp[3] / cvg, 4)),
             str(cols["mt3"][l]), str(cols["mt5"][l]), str(cols["mt7"][l]), str(cols["mt10"][l]),
             str(umt[0]), str(umt[1]), str(umt[2]), str(umt[3]),
             ff(rnd(1.0 * umt[0] / used, 4)), ff(rnd(1.0 * umt[1] / used, 4)), ff(rnd(1.0 * umt[2] / used, 4)),
             ff(rnd(1.0 * umt[3] / used, 4)),
             str(vsm[0]), str(vsm[1]), str(vsm[2]), str(vsm[3]),
             ff(rnd(pi[0], 2)), ff(rnd(pi[1], 2)), ff(rnd(pi[2], 2)), ff(rnd(pi[3], 2)), ";"]
        out[l] = "\t".join(f)


def pi_boundary_loci(rows, eps: float = 1e-8):
    """Loci whose printed text could differ from the reference's although every number agrees to the tolerance: a prediction
    index the row prints (PI_A..PI_C, the candidates' PI: smCounter.py:591-593) within `eps` of a half-way point of
    round(PI, 2) - which is also where int(float(PI)) meets the writers' threshold (:838-847) -, or a candidate's PI within
    `eps` of the filter gate altPI >= 5 (:549).  The device's PI sits within ~ 3e-9 of the reference's arithmetic (other
    summation order, own log / division routines), so a value this close to a boundary can land on its other side.  Nothing
    is changed: the caller reports these loci (the command line logs them, bench.py counts them)."""
    import numpy as np
    ok = (rows["status"] & 0xff) == 0
    vals = np.concatenate([rows["pi"], rows["cand"]["pi"]], axis=1)                 # [n, 6]
    live = np.concatenate([np.ones_like(rows["pi"], bool), rows["cand"]["allele"] >= 0], axis=1) & np.isfinite(vals)
    t = np.abs(vals) * 100.0
    half = np.abs((t - np.floor(t)) - 0.5) < eps * 100.0
    gate = np.abs(rows["cand"]["pi"] - 5.0) < eps
    gate &= rows["cand"]["allele"] >= 0
    return np.flatnonzero(ok & ((half & live).any(axis=1) | gate.any(axis=1)))
