"""The three output files of smCounter (smCounter.py:787-901): <prefix>.smCounter.all.txt (every
locus), .cut.txt and .cut.vcf (loci whose truncated PI reaches the threshold and whose ALT is not
'DEL')."""
from __future__ import annotations

import math
from typing import List

from .rows import HEADER_ALL

HEADER_VARIANTS = ("CHROM", "POS", "REF", "ALT", "TYPE", "DP", "MT", "UMT", "PI", "THR", "VMT", "VMF",
                   "VSM", "FILTER")
_COL = {name: i for i, name in enumerate(HEADER_ALL)}

_VCF_META = (
    "##fileformat=VCFv4.2",
    "##reference=GRCh37",
    '##INFO=<ID=TYPE,Number=1,Type=String,Description="Variant type: SNP or INDEL">',
    '##INFO=<ID=DP,Number=1,Type=Integer,Description="Total read depth">',
    '##INFO=<ID=MT,Number=1,Type=Integer,Description="Total MT depth">',
    '##INFO=<ID=UMT,Number=1,Type=Integer,Description="Filtered MT depth">',
    '##INFO=<ID=PI,Number=1,Type=Float,Description="Variant prediction index">',
    '##INFO=<ID=THR,Number=1,Type=Integer,Description="Variant prediction index minimum threshold">',
    '##INFO=<ID=VMT,Number=1,Type=Integer,Description="Variant MT depth">',
    '##INFO=<ID=VMF,Number=1,Type=Float,Description="Variant MT fraction">',
    '##INFO=<ID=VSM,Number=1,Type=Integer,Description="Variant strong MT depth">',
    '##FORMAT=<ID=GT,Number=1,Type=String,Description="Genotype">',
    '##FORMAT=<ID=AD,Number=.,Type=Integer,Description="Filtered allelic MT depths for the ref and alt alleles">',
    '##FORMAT=<ID=VF,Number=1,Type=Float,Description="Variant MT fraction, same as VMF">',
    '##FILTER=<ID=RepT,Description="Variant in simple tandem repeat region, as defined by Tandem Repeats Finder">',
    '##FILTER=<ID=RepS,Description="Variant in simple repeat region, as defined by RepeatMasker">',
    '##FILTER=<ID=LowC,Description="Variant in low complexity region, as defined by RepeatMasker">',
    '##FILTER=<ID=SL,Description="Variant in micro-satelite region, as defined by RepeatMasker">',
    '##FILTER=<ID=HP,Description="Inside or flanked by homopolymer region">',
    '##FILTER=<ID=LM,Description="Low coverage (fewer than 5 MTs)">',
    '##FILTER=<ID=LSM,Description="Fewer than 2 strong MTs">',
    '##FILTER=<ID=SB,Description="Strand bias">',
    '##FILTER=<ID=LowQ,Description="Low base quality (mean < 22)">',
    '##FILTER=<ID=MM,Description="Too many genome reference mismatches in reads (default threshold is 6.5 per 100 bases)">',
    '##FILTER=<ID=DP,Description="Too many discordant read pairs">',
    '##FILTER=<ID=R1CP,Description="Variants are clustered at the end of R1 reads">',
    '##FILTER=<ID=R2CP,Description="Variants are clustered at the end of R2 reads">',
    '##FILTER=<ID=PrimerCP,Description="Variants are clustered immediately after the primer, possible enzyme initiation error">',
)


def pi_threshold(mt_depth: int, threshold: int = 0) -> int:
    """Cut-off for about 20 false positives per Mb (smCounter.py:820; mt_depths_lod.R:19)."""
    return int(math.ceil(14.0 + 0.012 * mt_depth)) if threshold == 0 else threshold


def _genotype(chrom: str, alts: List[str], vmf: str) -> str:
    # the reference's "hack attempt to satisfy downstream software" (smCounter.py:867-878)
    if len(alts) == 2:
        return "1/2"
    if len(alts) != 1:
        raise ValueError("cannot derive a genotype for ALT %r" % (alts,))
    if chrom in ("chrY", "chrM"):
        return "1"
    return "1/1" if float(vmf) > 0.95 else "0/1"


def write_outputs(out_prefix: str, rows: List[str], threshold: int, pred=None) -> None:
    """rows: post-filtered 45-column strings in locus order.  `pred` (optional, rows.RowLines.pred): per row int(float(PI)) where
    the native printer knows it - rows below the threshold are then not split again to find that out."""
    cut_rows = rows
    if pred is not None and len(pred) == len(rows):
        import numpy as np
        from .rows import PRED_NONE
        pred = np.asarray(pred)
        cut_rows = [rows[i] for i in np.flatnonzero((pred >= threshold) | (pred == PRED_NONE)).tolist()]
    sample_col = "\t".join(("#CHROM", "POS", "ID", "REF", "ALT", "QUAL", "FILTER", "INFO", "FORMAT", out_prefix))
    with open(out_prefix + ".smCounter.all.txt", "w") as f_all, \
            open(out_prefix + ".smCounter.cut.txt", "w") as f_cut, \
            open(out_prefix + ".smCounter.cut.vcf", "w") as f_vcf:
        f_all.write("\t".join(HEADER_ALL) + "\n")
        f_cut.write("\t".join(HEADER_VARIANTS) + "\n")
        f_vcf.write("\n".join(_VCF_META) + "\n" + sample_col + "\n")
        if len(rows):
            f_all.write(getattr(rows, "text", None) or "\n".join(rows) + "\n")     # (postfilter._PassRows brings its text)
        i_pi, i_alt = _COL["PI"], _COL["ALT"]
        for row in cut_rows:
            f = row.split("\t", i_pi + 1)                    # (most rows stop here: PI below the threshold)
            if not f[i_pi]:
                continue                                     # Zero_Coverage rows
            qual = str(int(float(f[i_pi])))                  # truncated, phred-like
            if int(qual) < threshold or f[i_alt] == "DEL":
                continue
            f = row.split("\t")
            g = {name: f[i] for name, i in _COL.items()}
            thr = str(threshold)
            info = ";".join(k + "=" + v for k, v in (
                ("TYPE", g["TYPE"]), ("DP", g["DP"]), ("MT", g["MT"]), ("UMT", g["UMT"]), ("PI", g["PI"]),
                ("THR", thr), ("VMT", g["VMT"]), ("VMF", g["VMF"]), ("VSM", g["VSM"])))
            alts = g["ALT"].split(",")
            ad = str(int(g["UMT"]) - int(g["VMT"])) + "," + g["VMT"] + (",1" if len(alts) == 2 else "")
            sample = ":".join((_genotype(g["CHROM"], alts, g["VMF"]), ad, g["VMF"]))
            f_vcf.write("\t".join((g["CHROM"], g["POS"], ".", g["REF"], g["ALT"], qual, g["FILTER"], info,
                                   "GT:AD:VF", sample)) + "\n")
            f_cut.write("\t".join((g["CHROM"], g["POS"], g["REF"], g["ALT"], g["TYPE"], g["DP"], g["MT"], g["UMT"],
                                   g["PI"], thr, g["VMT"], g["VMF"], g["VSM"], g["FILTER"])) + "\n")
