"""Homopolymer / low-complexity test around a called variant (reference: isHPorLowComp,
smCounter.py:122-177).  Host side: it needs the reference sequence and runs only for loci whose
alt PI reaches 5 (smCounter.py:549), a small minority."""
from __future__ import annotations


def _top2_window(seq: str, win: int) -> bool:
    # any window of `win` bases whose two most frequent nucleotides make up >= 99 %; the
    # reference's range(totalLen - len2) leaves the last window out (smCounter.py:150,165)
    for i in range(len(seq) - win):
        sub = seq[i:i + win]
        c = sorted((sub.count("A"), sub.count("T"), sub.count("G"), sub.count("C")), reverse=True)
        if 1.0 * (c[0] + c[1]) / win >= 0.99:
            return True
    return False


def is_hp_or_lowcomp(chrom, pos, length, refb, altb, refprov):
    """-> (isHomopolymer, isLowComplexity) for ref/alt strings as convertToVcf returns them."""
    chrom_len = refprov.get_reference_length(chrom)
    pos0 = int(pos) - 1

    def flanks(n):
        left = refprov.fetch(chrom, max(0, pos0 - n), pos0).upper()
        r_ref = refprov.fetch(chrom, pos0 + len(refb), min(pos0 + len(refb) + n, chrom_len)).upper()
        r_alt = refprov.fetch(chrom, pos0 + len(altb), min(pos0 + len(altb) + n, chrom_len)).upper()
        return left + refb + r_ref, left + altb + r_alt

    ref_seq, alt_seq = flanks(length)
    homop = any(nt * length in ref_seq or nt * length in alt_seq for nt in "ATGC")
    ref_lc, alt_lc = flanks(2 * length)
    lowcomp = _top2_window(ref_lc, 2 * length) or _top2_window(alt_lc, 2 * length)
    return homop, lowcomp
