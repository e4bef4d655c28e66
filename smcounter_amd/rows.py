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


def format_row(row, chrom: str, pos, orig_ref: str, alleles, params: VcParams, refprov) -> str:
    pos = str(int(pos))
    st = int(row["status"])
    if st & abi.ST_BAD_INPUT:
        raise RowError("locus %s:%s: batch violates the layout contract (ids out of range)" % (chrom, pos))
    if (st & 0xff) == abi.ST_ZERO_COVERAGE:
        return "\t".join([chrom, pos, orig_ref] + [""] * 41 + ["Zero_Coverage"])
    c0 = row["cand"][0]
    orig_alt = alleles[int(c0["allele"])]
    ref, alt, vtype = convert_to_vcf(orig_ref, orig_alt)
    fltr = filter_string(c0, chrom, pos, ref, alt, params, refprov)
    chosen = c0
    if row["biallelic"]:
        c1 = row["cand"][1]
        ref2, alt2, vtype2 = convert_to_vcf(orig_ref, alleles[int(c1["allele"])])
        fltr2 = filter_string(c1, chrom, pos, ref2, alt2, params, refprov)
        if fltr == ";" and fltr2 == ";":
            alt = alt + "," + alt2
            vtype = vtype.lower() + "," + vtype2.lower()
        elif fltr != ";" and fltr2 == ";":
            alt, fltr, chosen = alt2, fltr2, c1
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


def format_rows(rows, db, params: VcParams, refprov):
    """`format_row` over a batch.  Loci whose candidate goes through no filter and is not bi-allelic (nearly all
    of a panel) take a column-wise path - the structured array is unpacked once into Python lists instead of one
    numpy field access per value; the others go through `format_row`.  Same strings either way
    (tests/test_host_logic.py compares the two on the golden loci)."""
    n = len(rows)
    if n == 0:
        return []
    st = rows["status"].tolist()
    cand0 = rows["cand"][:, 0]
    simple = ((rows["status"] & 0xff) == 0) & ((rows["status"] & abi.ST_BAD_INPUT) == 0) & (rows["biallelic"] == 0) \
        & (cand0["flt_applied"] == 0)
    simple = simple.tolist()
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
