"""Repeat-region flags and FILTER normalisation applied to vc() rows in main()
(smCounter.py:696-785), with the bedtools steps done in-process (bedops.py)."""
from __future__ import annotations

from collections import defaultdict
from typing import Dict, List, Optional

from . import bedops
from .rows import HEADER_ALL

_COL = {name: i for i, name in enumerate(HEADER_ALL)}
_RM_CODES = {"Simple_repeat": "RepS", "Low_complexity": "LowC", "Satellite": "SL"}


def load_repeat_regions(bed_target: str, bed_tandem: Optional[str], bed_repeatmasker: Optional[str]):
    """-> (trfRegions, rmRegions): {chrom: [(start, end, 'Flag;...')]} restricted to the target,
    as main() builds them from bedtools output (smCounter.py:699-734)."""
    target = bedops.sort_bed(bedops.merge(bedops.sort_bed(bedops.read_bed(bed_target))))
    trf: Dict[str, list] = defaultdict(list)
    rm: Dict[str, list] = defaultdict(list)
    if bed_tandem:
        # the reference assumes the TRF track is already merged and sorted (:699)
        for c, s, e, _ in bedops.sort_bed(bedops.intersect(bedops.read_bed(bed_tandem), target)):
            trf[c].append((s, e, "RepT;"))
    if bed_repeatmasker:
        merged = bedops.sort_bed(bedops.merge(bedops.read_bed(bed_repeatmasker), distinct_names=True))
        for c, s, e, codes in bedops.sort_bed(bedops.intersect(merged, target)):
            flags = [_RM_CODES.get(code, "Other_Repeat") for code in codes.split(",")]
            rm[c].append((s, e, ";".join(flags) + ";"))
    return trf, rm


def apply_repeat_filters(rows: List[str], trf, rm) -> List[str]:
    """Append RepT / RepS / LowC / SL / Other_Repeat to the raw FILTER of rows with int(PI) >= 5 and
    ALT != 'DEL', then turn ';' into PASS and strip the semicolons (smCounter.py:751-785).  Rows whose
    POS or VMF is not numeric (the Zero_Coverage rows) pass through untouched, as in the reference."""
    out = []
    for row in rows:
        f = row.split("\t")
        try:
            pos = int(f[_COL["POS"]])
            vmf = float(f[_COL["VMF"]])
        except ValueError:
            out.append(row)
            continue
        try:
            pred = int(float(f[_COL["PI"]]))
        except ValueError:
            pred = 0
        if pred >= 5 and f[_COL["ALT"]] != "DEL":
            chrom = f[_COL["CHROM"]]
            if vmf < 40:          # sic: a fraction compared with 40, always true (:772)
                for lo, hi, flag in trf.get(chrom, ()):
                    if lo < pos <= hi:
                        f[-1] += flag
                        break
            for lo, hi, flag in rm.get(chrom, ()):
                if lo < pos <= hi:
                    f[-1] += flag
                    break
        f[-1] = "PASS" if f[-1] == ";" else f[-1].strip(";")
        out.append("\t".join(f))
    return out
