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


def format_rows(rows, db, params: VcParams, refprov):
    return [format_row(rows[l], db.chrom[l], db.pos[l], db.ref[l], db.alleles[l], params, refprov)
            for l in range(len(rows))]
