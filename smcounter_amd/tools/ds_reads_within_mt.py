"""Down-sample reads inside molecular tags to a target mean of `rpb` reads per barcode.

Mirrors the reference's `ds.reads.withinMT.py:22-90` under CPython-2.7 semantics: per barcode the distinct
read names in order of first appearance (:37-44); `probKeep = (rpb - 1) * (#barcodes) / (reads of multi-read
barcodes - #multi-read barcodes)` (:58); barcodes visited in py2 dict order (`bcDict.values()`, :62), the
first read name of a barcode always kept, every further one kept when `random.random() <= probKeep`
(:64-73).  All alignments of a kept read name are written (raw records, header copied).
"""
from __future__ import annotations

import argparse
import os

from .. import bamio
from ..py2compat import Py2Random, py2_dict_order
from .ds_mt import barcode_of


def select_reads(qnames, rpb: float, seed: int):
    per_bc, seen, order = {}, set(), []
    for q in qnames:
        bc = barcode_of(q)
        if bc not in per_bc:
            per_bc[bc] = []
            order.append(bc)
        if (bc, q) not in seen:
            seen.add((bc, q))
            per_bc[bc].append(q)
    one = sum(1 for v in per_bc.values() if len(v) == 1)
    multi = sum(1 for v in per_bc.values() if len(v) > 1)
    multi_reads = sum(len(v) for v in per_bc.values() if len(v) > 1)
    prob_keep = 1.0 * (rpb - 1.0) * (one + multi) / (multi_reads - multi)      # :58 (ZeroDivisionError like the reference)
    rng = Py2Random(int(seed))
    selected = set()
    for bc in py2_dict_order(order):
        reads = per_bc[bc]
        selected.add(reads[0])
        for rid in reads[1:]:
            if rng.random() <= prob_keep:
                selected.add(rid)
    return selected, prob_keep


def main(args) -> int:
    if args.runPath:
        os.chdir(args.runPath)
    _, recs = bamio.iter_raw_records(args.inBam)
    selected, _ = select_reads((q for tid, q, _ in recs if tid >= 0), args.rpb, args.seed)
    header, recs = bamio.iter_raw_records(args.inBam)
    n = [0]

    def chosen():
        for tid, q, raw in recs:
            if tid >= 0 and q in selected:
                n[0] += 1
                yield raw
    bamio.write_raw(args.outBam, header, chosen())
    return n[0]


def build_parser():
    parser = argparse.ArgumentParser(description="Downsample MTs")
    parser.add_argument("--runPath", default=None, help="path to working directory")
    parser.add_argument("--inBam", default=None, help="Input BAM file")
    parser.add_argument("--outBam", default=None, help="Output BAM file")
    parser.add_argument("--rpb", type=float, default=1.0, help="target reads per MT")
    parser.add_argument("--seed", type=int, default=1234567, help="Seed for random number generation")
    return parser


if __name__ == "__main__":
    main(build_parser().parse_args())
