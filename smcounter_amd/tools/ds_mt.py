"""Down-sample molecular tags: keep every barcode (and all its reads) with probability `pct`.

Mirrors the reference's `ds.mt.py:23-72` under CPython-2.7 semantics: the barcodes are visited in the key
order of a py2 dict filled in order of first appearance (`for bc in bcDict.keys()`, :51) and each draws one
`random.random()` from the generator seeded with `--seed` (:24); a barcode is kept when `r <= pct` (:53).
Reads are copied verbatim (raw BAM records), header included (`template=samfile`, :33).  Only reads placed on
a reference are visited, like `samfile.fetch()` without a region.
"""
from __future__ import annotations

import argparse
import os

from .. import bamio
from ..py2compat import Py2Random, py2_dict_order


def barcode_of(qname: str) -> str:
    return qname.strip().split(":")[-2]                       # ds.mt.py:43-45


def select_barcodes(qnames, pct: float, seed: int):
    """-> set of kept barcodes.  `qnames`: read names in file order."""
    seen, order = set(), []
    for q in qnames:
        bc = barcode_of(q)
        if bc not in seen:
            seen.add(bc)
            order.append(bc)
    rng = Py2Random(int(seed))
    return {bc for bc in py2_dict_order(order) if rng.random() <= pct}


def main(args) -> int:
    if args.runPath:
        os.chdir(args.runPath)
    print("reading in bam file")
    _, recs = bamio.iter_raw_records(args.inBam)
    kept = select_barcodes((q for tid, q, _ in recs if tid >= 0), args.pct, args.seed)
    print("Writing to BAM file")
    header, recs = bamio.iter_raw_records(args.inBam)
    n = [0]

    def chosen():
        for tid, q, raw in recs:
            if tid >= 0 and barcode_of(q) in kept:
                n[0] += 1
                yield raw
    bamio.write_raw(args.outBam, header, chosen())
    return n[0]


def build_parser():
    parser = argparse.ArgumentParser(description="Downsample MTs")
    parser.add_argument("--runPath", default=None, help="path to working directory")
    parser.add_argument("--inBam", default=None, help="Input BAM file")
    parser.add_argument("--outBam", default=None, help="Output BAM file")
    parser.add_argument("--pct", type=float, default=0.5, help="Percent of MTs kept")
    parser.add_argument("--seed", type=int, default=1234567, help="Seed for random number generation")
    return parser


if __name__ == "__main__":
    main(build_parser().parse_args())
