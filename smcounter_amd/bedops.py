"""The handful of BED interval operations smCounter's main() shells out to bedtools for
(smCounter.py:699-710): merge (optionally collapsing column 4 to its distinct values), sort, and
intersect (report the overlapping portion of every A/B pair).  Pure Python on lists of tuples."""
from __future__ import annotations

from typing import Iterable, List, Tuple

Interval = Tuple[str, int, int, str]      # chrom, start, end, name ('' when the BED has 3 columns)


def read_bed(path: str) -> List[Interval]:
    out = []
    with open(path) as fh:
        for line in fh:
            if not line.strip() or line.startswith(("track ", "browser ", "#")):
                continue
            f = line.rstrip("\n").split("\t")
            out.append((f[0], int(f[1]), int(f[2]), f[3] if len(f) > 3 else ""))
    return out


def sort_bed(iv: Iterable[Interval]) -> List[Interval]:
    """`bedtools sort`: by chromosome name, then start."""
    return sorted(iv, key=lambda r: (r[0], r[1]))


def merge(iv: Iterable[Interval], distinct_names: bool = False) -> List[Interval]:
    """`bedtools merge [-c 4 -o distinct]`: fuse overlapping and book-ended intervals of a
    position-sorted input; with distinct_names the merged record's name is the comma-joined set of
    the members' names."""
    out: List[Interval] = []
    cur = None
    names: List[str] = []
    for c, s, e, nm in iv:
        if cur is not None and c == cur[0] and s <= cur[2]:
            cur[2] = max(cur[2], e)
        else:
            if cur is not None:
                out.append((cur[0], cur[1], cur[2], ",".join(sorted(set(names))) if distinct_names else ""))
            cur, names = [c, s, e], []
        names.append(nm)
    if cur is not None:
        out.append((cur[0], cur[1], cur[2], ",".join(sorted(set(names))) if distinct_names else ""))
    return out


def intersect(a: Iterable[Interval], b: Iterable[Interval]) -> List[Interval]:
    """`bedtools intersect -a A -b B`: for every overlapping (A, B) pair, the shared span with A's
    name."""
    by_chrom = {}
    for c, s, e, _ in b:
        by_chrom.setdefault(c, []).append((s, e))
    for v in by_chrom.values():
        v.sort()
    out = []
    for c, s, e, nm in a:
        for bs, be in by_chrom.get(c, ()):
            if bs >= e:
                break
            lo, hi = max(s, bs), min(e, be)
            if lo < hi:
                out.append((c, lo, hi, nm))
    return out


def expand_loci(path: str):
    """BED -> [(chrom, '1-based pos')] in file order, one per base (smCounter.py:674-680)."""
    loci = []
    with open(path) as fh:
        for line in fh:
            if line.startswith("track "):
                continue
            chrom, start, end = line.strip().split("\t")[0:3]
            loci.extend((chrom, str(p + 1)) for p in range(int(start), int(end)))
    return loci
