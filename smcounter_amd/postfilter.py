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


def _apply_one(row: str, trf, rm) -> str:
    f = row.split("\t")
    try:
        pos = int(f[_COL["POS"]])
        vmf = float(f[_COL["VMF"]])
    except ValueError:
        return row
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
    return "\t".join(f)


def apply_repeat_filters(rows: List[str], trf, rm, pred=None) -> List[str]:
    """Append RepT / RepS / LowC / SL / Other_Repeat to the raw FILTER of rows with int(PI) >= 5 and
    ALT != 'DEL', then turn ';' into PASS and strip the semicolons (smCounter.py:751-785).  Rows whose
    POS or VMF is not numeric (the Zero_Coverage rows) pass through untouched, as in the reference.
    (`_apply_one` is the plain statement; the loop below does the same on the unchanged rows - nearly all of a
    panel: PI below 5, FILTER ';' - without splitting all 45 fields.)
    `pred` (optional, rows.RowLines.pred): per row int(float(PI)) as the native printer saw it, for rows it printed with
    numeric POS / VMF and the raw FILTER ';' (rows.PRED_NONE elsewhere): those below 5 - nearly all of a panel - become
    '...PASS' by one replace over the joined text instead of a Python step per row."""
    if pred is not None and len(pred) == len(rows) and rows:
        import numpy as np
        from .rows import PRED_NONE
        pred = np.asarray(pred)
        out = list(rows)
        special = np.flatnonzero((pred >= 5) | (pred == PRED_NONE)).tolist()
        for i in special:
            out[i] = _apply_any(out[i], trf, rm)
        text = "\n".join(out)
        # (a treated row that keeps a bare ';' - POS / VMF not numeric: passed through untouched - or a line break inside a
        # row: the plain way)
        if text.count("\n") == len(out) - 1 and not any(out[i].endswith("\t;") for i in special):
            return _PassRows(out, (text + "\n").replace("\t;\n", "\tPASS\n"))
        return [_apply_any(r, trf, rm) for r in rows]
    out = []
    for row in rows:
        out.append(_apply_any(row, trf, rm))
    return out


class _PassRows(object):
    """The post-filtered rows as a sequence that already holds its own joined text: `rows` are final except that the
    untouched ones (nearly all) still end in the raw ';' - an item is completed to 'PASS' when it is asked for - and `text`
    is all of them, completed, newline-terminated: what the writer of all.txt wants, without a split and a second join."""

    def __init__(self, rows, text):
        self._rows, self.text = rows, text

    def __len__(self):
        return len(self._rows)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[k] for k in range(*i.indices(len(self._rows)))]
        r = self._rows[i]
        return r[:-1] + "PASS" if r.endswith("\t;") else r

    def __iter__(self):
        for r in self._rows:
            yield r[:-1] + "PASS" if r.endswith("\t;") else r

    def __eq__(self, other):
        return list(self) == list(other)

    def __ne__(self, other):
        return not self == other


def _apply_any(row: str, trf, rm) -> str:
    i_pi, i_vmf = _COL["PI"], _COL["VMF"]
    if not row.endswith("\t;"):
        return _apply_one(row, trf, rm)
    head = row.split("\t", i_vmf + 1)
    if len(head) != i_vmf + 2:
        return _apply_one(row, trf, rm)
    try:
        int(head[_COL["POS"]])
        float(head[i_vmf])
        pred = int(float(head[i_pi]))
    except ValueError:
        return _apply_one(row, trf, rm)
    if pred >= 5:
        return _apply_one(row, trf, rm)
    return row[:-1] + "PASS"
