"""Comparing two batches whose planes follow the layout contract of include/smcounter_hip.h but number the barcodes and the
fragments of a locus differently.

The contract asks for a dense barcode-major layout (reads of a barcode contiguous, fragments of a barcode contiguous, pileup
order kept WITHIN a fragment); which barcode comes first, and which fragment first within its barcode, is the builder's
choice - the host builders (features.extract_features, smc_bam_planes) number by first appearance at the locus, the device
builder (csrc/k_build_planes.inc) by first appearance in the run.  Nothing vc() computes depends on that order
(smCounter.py:462-532 works on dicts keyed by barcode and read id).  `layout_problems` checks that a batch obeys the
contract; `signature` gives per locus an order-free fingerprint - equal fingerprints mean the same set of barcodes, each with
the same set of fragments, each with the same reads (every plane word except the two numberings) in the same order.
"""
from __future__ import annotations

from typing import List

import numpy as np

from .features import USTART_DROPPED

SLOT_MASK = np.uint32(0x07FFFFFF)
_M1, _M2 = np.uint64(0xBF58476D1CE4E5B9), np.uint64(0x94D049BB133111EB)


def _mix(x: np.ndarray) -> np.ndarray:
    x = x.astype(np.uint64, copy=True)
    with np.errstate(over="ignore"):
        x ^= x >> np.uint64(30); x *= _M1
        x ^= x >> np.uint64(27); x *= _M2
        x ^= x >> np.uint64(31)
    return x


def layout_problems(db) -> List[str]:
    """Violations of the layout contract (dense ascending slots, barcode ranges = umi_start, umi plane = barcode index)."""
    bad = []
    for l in range(db.n_loci):
        L = db.loci[l]
        o, n, nu, nf = 4 * int(L["read_off4"]), int(L["n_reads"]), int(L["n_umi"]), int(L["n_frag"])
        us = db.umi_start[int(L["umi_off"]):int(L["umi_off"]) + nu + 1].astype(np.int64) & 0x7FFFFFFF
        if n == 0:
            if nu or nf or int(us[0]) != 0:
                bad.append("locus %d: empty locus with barcodes" % l)
            continue
        slot = (db.frag[o:o + n] & SLOT_MASK).astype(np.int64)
        d = np.diff(slot)
        if slot[0] != 0 or slot[-1] != nf - 1 or ((d != 0) & (d != 1)).any():
            bad.append("locus %d: fragment slots not dense ascending" % l)
        if us[0] != 0 or us[-1] != n or (np.diff(us) <= 0).any():
            bad.append("locus %d: umi_start not ascending over [0, n_reads]" % l)
            continue
        if (d[us[1:-1] - 1] != 1).any():
            bad.append("locus %d: a barcode begins inside a fragment" % l)
        if db.umi is not None:
            want = np.repeat(np.arange(nu), np.diff(us))
            if not np.array_equal(db.umi[o:o + n].astype(np.int64), want):
                bad.append("locus %d: umi plane is not the barcode index" % l)
        pad = ((n + 3) & ~3) - n
        if pad and (db.meta[o + n:o + n + pad].any() or db.frag[o + n:o + n + pad].any()):
            bad.append("locus %d: padding slots not zero" % l)
    return bad


def signature(db, l: int) -> np.ndarray:
    """Sorted fingerprints of the barcodes of locus l (uint64[n_umi]): independent of barcode / fragment numbering."""
    L = db.loci[l]
    o, n, nu = 4 * int(L["read_off4"]), int(L["n_reads"]), int(L["n_umi"])
    if n == 0:
        return np.zeros(0, np.uint64)
    usr = db.umi_start[int(L["umi_off"]):int(L["umi_off"]) + nu + 1]
    us = (usr & np.uint32(0x7FFFFFFF)).astype(np.int64)
    dropped = (usr[:nu] & np.uint32(USTART_DROPPED)) != 0
    meta = db.meta[o:o + n].astype(np.uint64)
    cls = (db.frag[o:o + n] >> np.uint32(27)).astype(np.uint64)
    dist = db.dist[o:o + n].astype(np.uint64) if db.dist is not None else np.zeros(n, np.uint64)
    slot = (db.frag[o:o + n] & SLOT_MASK).astype(np.int64)
    fstart = np.flatnonzero(np.r_[True, np.diff(slot) != 0])                 # first read of every fragment
    k = np.arange(n) - np.repeat(fstart, np.diff(np.r_[fstart, n]))          # position of the read within its fragment
    with np.errstate(over="ignore"):
        hr = _mix(meta | (dist << np.uint64(32))) ^ _mix(cls + np.uint64(977) * (k.astype(np.uint64) + np.uint64(1)))
        hf = _mix(np.add.reduceat(_mix(hr), fstart))                          # fragment: its reads in order
        bfirst = np.searchsorted(fstart, us[:-1])                             # first fragment of every barcode
        hb = np.add.reduceat(hf, bfirst) if len(bfirst) else np.zeros(0, np.uint64)
        hb = _mix(hb + np.uint64(31) * np.diff(us).astype(np.uint64)) ^ (dropped.astype(np.uint64) << np.uint64(63))
    return np.sort(hb)


DESCRIPTOR_FIELDS = ("n_reads", "n_umi", "n_frag", "ref_allele", "n_alleles", "flags", "snp_mask", "read_off4")


def differences(a, b, limit: int = 5) -> List[str]:
    """What distinguishes two batches beyond barcode / fragment numbering (empty list: equivalent)."""
    out = []
    if a.n_loci != b.n_loci:
        return ["%d loci vs %d" % (a.n_loci, b.n_loci)]
    for f in DESCRIPTOR_FIELDS:
        if not np.array_equal(a.loci[f], b.loci[f]):
            out.append("descriptor field %s differs (first at locus %d)" % (f, int(np.flatnonzero(a.loci[f] != b.loci[f])[0])))
    out += layout_problems(a)[:limit] + layout_problems(b)[:limit]
    if out:
        return out[:limit]
    for l in range(a.n_loci):
        if not np.array_equal(signature(a, l), signature(b, l)):
            out.append("locus %d: barcodes / fragments / reads differ" % l)
            if len(out) >= limit:
                break
    if list(a.chrom) != list(b.chrom) or list(a.ref) != list(b.ref) or not np.array_equal(a.pos, b.pos):
        out.append("chrom / pos / ref differ")
    if a.alleles != b.alleles:
        out.append("allele tables differ")
    return out
