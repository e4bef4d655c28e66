"""Theoretical limit of detection per locus from the barcode depth (reference: `mt_depths_lod.R:1-49`).

LOD = the smallest allele fraction p with P(#variant barcodes >= needed) >= 0.95 under Binomial(depth, p),
i.e. the root of `pbinom(needed - 1, depth, p) - 0.05` on [0, 1] (`mt_depths_lod.R:24-37`), where
`needed = ceiling((14.0 + 0.012 * meanMtDepth) / 3.5)` (:4-5, :20-21) and loci with fewer than 5 barcodes (or
no number) get 1.0.  R finds the root with `uniroot` at its default tolerance `.Machine$double.eps^0.25`
(~1.2e-4) and then rounds to 4 decimals, so the last digit depends on the root finder's own iterates:
`zeroin` below restates R's `R_zeroin2` (Brent's method as in R's src/appl/zeroin.c) step for step.
Unpinned: the reference ships no output of this script and R is not available here.
"""
from __future__ import annotations

import math
import sys

import numpy as np

PI_PER_BARCODE = 3.5                       # mt_depths_lod.R:5
EPSILON = 2.220446049250313e-16            # DBL_EPSILON


def zeroin(f, ax: float, bx: float, fa: float, fb: float, tol: float, maxit: int = 1000) -> float:
    """Brent root finder with R's R_zeroin2 control flow (interval end values supplied, like uniroot does)."""
    a, b, c, fc = ax, bx, ax, fa
    if fa == 0.0:
        return a
    if fb == 0.0:
        return b
    for _ in range(maxit + 1):
        prev_step = b - a
        if abs(fc) < abs(fb):               # swap so that b is the best approximation
            a, b, c = b, c, b
            fa, fb, fc = fb, fc, fb
        tol_act = 2 * EPSILON * abs(b) + tol / 2
        new_step = (c - b) / 2
        if abs(new_step) <= tol_act or fb == 0.0:
            return b
        if abs(prev_step) >= tol_act and abs(fa) > abs(fb):
            cb = c - b
            if a == c:                      # linear interpolation
                t1 = fb / fa
                p = cb * t1
                q = 1.0 - t1
            else:                           # inverse quadratic interpolation
                q = fa / fc
                t1 = fb / fc
                t2 = fb / fa
                p = t2 * (cb * q * (q - t1) - (b - a) * (t1 - 1.0))
                q = (q - 1.0) * (t1 - 1.0) * (t2 - 1.0)
            if p > 0:
                q = -q
            else:
                p = -p
            if p < (0.75 * cb * q - abs(tol_act * q) / 2) and p < abs(prev_step * q / 2):
                new_step = p / q
        if abs(new_step) < tol_act:
            new_step = tol_act if new_step > 0 else -tol_act
        a, fa = b, fb
        b += new_step
        fb = f(b)
        if (fb > 0 and fc > 0) or (fb < 0 and fc < 0):
            c, fc = a, fa
    raise RuntimeError("zeroin: no convergence")


def barcodes_needed(mean_mt_depth: float) -> int:
    return int(math.ceil((14.0 + 0.012 * mean_mt_depth) / PI_PER_BARCODE))      # :20-21


def find_lod(barcode_depth, needed: int) -> float:
    """mt_depths_lod.R:26-39."""
    from scipy.stats import binom
    try:
        depth = float(barcode_depth)
    except (TypeError, ValueError):
        return 1.0
    if not (depth == depth) or depth < 5:
        return 1.0
    f = lambda p: float(binom.cdf(needed - 1, depth, p)) - 0.05
    f_lo, f_hi = f(0.0), f(1.0)
    if not (f_lo * f_hi <= 0):              # uniroot: "f() values at end points not of opposite sign" -> try-error
        return 1.0
    try:
        root = zeroin(f, 0.0, 1.0, f_lo, f_hi, EPSILON ** 0.25)
    except RuntimeError:
        return 1.0
    return round(root, 4)


def _fmt(x) -> str:
    return "%.15g" % x                       # R's default number formatting in write.table


def main(argv) -> int:
    mean_depth, file_in, file_out = float(argv[0]), argv[1], argv[2]
    needed = barcodes_needed(mean_depth)
    print('[1] "cutoff.20: %s barcode.needed.20: %d"' % (_fmt(14.0 + 0.012 * mean_depth), needed))
    rows, lods = [], []
    for line in open(file_in):
        if not line.strip():
            continue
        chrom, loc_l, loc_r, mts = line.rstrip("\n").split("|")[:4]
        lod = find_lod(mts if mts not in ("NA", "") else float("nan"), needed)
        rows.append((chrom, loc_l, loc_r, lod))
        lods.append(lod)
    with open(file_out, "w") as fh:          # bedgraph, MTs column dropped (:45-48)
        for chrom, loc_l, loc_r, lod in rows:
            fh.write("\t".join((chrom, loc_l, loc_r, _fmt(lod))) + "\n")
    probs = (0.01, 0.05, 0.10, 0.50, 0.90, 0.95, 0.99)
    q = np.quantile(np.array(lods, float), probs) if lods else [float("nan")] * 7   # R's default type 7
    with open(file_out + ".quantiles.txt", "w") as fh:
        for p, v in zip(probs, q):
            fh.write("%d%%|%s\n" % (round(p * 100), _fmt(v)))
    return 0


if __name__ == "__main__":
    main(sys.argv[1:])
