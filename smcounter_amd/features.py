"""Per-read feature extraction (reference: smCounter.py:327-366, :371-452) and the HBM layout.

Turns a `PileupBatch` into the structure-of-arrays batch the HIP kernels consume: four 32-bit
planes, 16 bytes per pileup read, every locus's reads contiguous and starting on a 4-read
(16-byte) boundary so a wavefront reads 1 KiB per plane per load:

    meta[i] = allele | bq << 8 | flags << 16 | mq << 24
    umi[i]  = barcode index within the locus
    frag[i] = fragment slot within the locus: fragments of barcode u occupy the contiguous slots
              [sum of fragment counts of barcodes < u, ...), in order of first appearance within
              the barcode (= the index of the readid in allBcDict[BC], smCounter.py:463-464)
    dist[i] = distToBcEnd | distToPrimerEnd << 16        (regular bases only, saturated)

flags: bit0 pairOrder is R2 (smCounter.py:359-362, carried over from the previous read when a
record has neither flag, exactly as the reference's un-reset variable does), bit1 reverse strand
(:365), bit2 `mismatchPer100b <= mismatchThr` (:352-356 and the third term of incCond at
:378/:400/:421/:431), bits3-4 kind: 0 regular base, 1 'DEL' (inside a deletion, :416),
2 insertion start (:371), 3 deletion start (:392).

Plus one 32-byte descriptor per locus (`LOCUS_DTYPE`) with the CSR offset, the counts that size
the on-chip tables (reads, barcodes, fragments) and the allele-table facts the kernels need
(reference allele id, number of alleles, which alleles are single letters).
"""
from __future__ import annotations

import dataclasses
from typing import List, Optional

import numpy as np

from .params import VcParams
from .pileup import (F_READ1, F_READ2, F_REVERSE, MAX_ALLELES, PileupBatch, BASE_ALLELES)

LOCUS_DTYPE = np.dtype([("read_off4", "<u4"), ("umi_off", "<u4"), ("n_reads", "<i4"), ("n_umi", "<i4"),
                        ("n_frag", "<i4"), ("ref_allele", "u1"), ("n_alleles", "u1"),
                        ("flags", "<u2"), ("snp_mask", "<u8")])
assert LOCUS_DTYPE.itemsize == 32

FL_R2, FL_REV, FL_MMOK = 1, 2, 4
LF_SAMPLED = 1                  # smc_locus.flags: host-applied down-sampling (include/smcounter_hip.h)
LF_FP_SHIFT = 1                 # smc_locus.flags bits 1-15: fingerprint of the parameters the planes bake in


def param_fingerprint(params) -> int:
    """smc_param_fingerprint (include/smcounter_hip.h): 15 bits, never 0, of (minBQ, minMQ, mismatchThr, primerDist) -
    the parameters folded into the planes (read class, mismatch flag, in-deletion quality).  smc_plan_run refuses to
    run a batch under any other set."""
    import struct
    m = (1 << 64) - 1
    h = 0x9E3779B97F4A7C15
    w = (params.minBQ & 0xFFFFFFFF, params.minMQ & 0xFFFFFFFF,
         struct.unpack("<Q", struct.pack("<d", float(params.mismatchThr)))[0], params.primerDist & 0xFFFFFFFF)
    for x in w:
        h ^= x
        h = (h * 0xBF58476D1CE4E5B9) & m
        h ^= h >> 29
    fp = (h >> 17) & 0x7FFF
    return fp or 1
USTART_DROPPED = 0x80000000
FRAG_SLOT_MASK = 0x07FFFFFF      # frag plane: bits 0-26 fragment slot, bits 27-31 read class (smcounter_hip.h)
FRAG_CLASS_SHIFT = 27


def read_class(kind, rev, r2, inc, bq_ok, le20, prle):
    """numpy version of smc_read_class (include/smcounter_hip.h): what a read adds to the tallies, as a code."""
    kind = np.asarray(kind)
    rev, r2, inc, bq_ok, le20, prle = (np.asarray(x).astype(np.uint32) for x in (rev, r2, inc, bq_ok, le20, prle))
    sub = np.where(inc == 0, 1 - bq_ok, np.where(r2 == 0, 2 + le20, 4 + le20 + 2 * prle))
    return np.where(kind == KIND_INDEL_GAP, inc, np.where(kind != KIND_BASE, 2 + 2 * rev + inc, 6 + 8 * rev + sub)).astype(np.uint32)
KIND_SHIFT = 3
KIND_BASE, KIND_INDEL_GAP, KIND_INS, KIND_DELSTART = 0, 1, 2, 3
READ_ALIGN = 4
MAX_BQ = 126          # device contract (csrc/device_common.inc: PIDX_UNPAIRED - 1)


class PileupError(Exception):
    """A record the reference itself would fail on (it would raise inside vc())."""


@dataclasses.dataclass
class DeviceBatch:
    """Host-side image of the HBM batch (numpy); `engine` uploads it verbatim."""
    loci: np.ndarray          # LOCUS_DTYPE[n_loci]
    meta: np.ndarray          # uint32[n_slots]
    umi: np.ndarray
    frag: np.ndarray
    dist: np.ndarray
    umi_start: np.ndarray     # uint32[sum(n_umi + 1)]: per locus, first read of each barcode (+ n_reads)
    # host-only context for formatting rows
    chrom: List[str]
    pos: np.ndarray
    ref: List[str]
    alleles: List[List[str]]
    umi_names: Optional[List[List[str]]] = None

    @property
    def n_loci(self) -> int:
        return len(self.loci)

    @property
    def n_slots(self) -> int:
        return len(self.meta)

    @property
    def n_reads(self) -> int:
        return int(self.loci["n_reads"].sum())

    def input_bytes(self) -> int:
        return 16 * self.n_slots + self.loci.nbytes + self.umi_start.nbytes

    def read_off(self, l: int) -> int:
        return 4 * int(self.loci["read_off4"][l])


def ref_allele_id(ref: str, table: List[str]) -> int:
    try:
        return table.index(ref)
    except ValueError:
        return 255          # reference letter never seen as an allele key


def extract_features(pb: PileupBatch, params: VcParams) -> DeviceBatch:
    n_loci = pb.n_loci
    n = pb.n_reads
    lens = np.diff(pb.read_off).astype(np.int64)
    locus_of = np.repeat(np.arange(n_loci, dtype=np.int64), lens)

    # --- pairOrder: R2 wins over R1; neither -> previous read's value (smCounter.py:359-362)
    has1 = (pb.flag & F_READ1) != 0
    has2 = (pb.flag & F_READ2) != 0
    known = has1 | has2
    if n:
        first = pb.read_off[:-1][lens > 0]
        if not known[first].all():
            bad = int(locus_of[first[~known[first]][0]])
            raise PileupError("first pileup read at %s:%d has neither read1 nor read2 set; the "
                              "reference's pairOrder is undefined there (smCounter.py:359-381)"
                              % (pb.chrom[bad], pb.pos[bad]))
        idx = np.where(known, np.arange(n), 0)
        np.maximum.accumulate(idx, out=idx)
        is_r2 = has2[idx]
    else:
        is_r2 = np.zeros(0, bool)
    rev = (pb.flag & F_REVERSE) != 0

    # --- mismatches per 100 bases (smCounter.py:352-356), compared with <= (:378)
    mismatch = np.maximum(0, pb.nm.astype(np.int64) - pb.n_indel.astype(np.int64))
    qlen = pb.qlen.astype(np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        mm100 = np.where(pb.qlen > 0, 100.0 * mismatch / qlen, 0.0)
    mm_ok = mm100 <= params.mismatchThr

    # --- allele kind (smCounter.py:371, :392, :416)
    kind = np.where(pb.indel > 0, KIND_INS,
                    np.where(pb.indel < 0, KIND_DELSTART,
                             np.where(pb.is_del, KIND_INDEL_GAP, KIND_BASE))).astype(np.uint32)

    # --- end distances, regular bases only (smCounter.py:432-452)
    rel = pb.qpos.astype(np.int64) - pb.left_sp.astype(np.int64)
    far = pb.qalen.astype(np.int64) - rel
    d_bc = np.where(is_r2, np.where(rev, rel, far), np.where(rev, far, rel))
    d_pr = np.where(is_r2, np.where(rev, far, rel), 0)
    regular = kind == KIND_BASE
    d_bc = np.where(regular, np.clip(d_bc, 0, 65535), 0).astype(np.uint32)
    d_pr = np.where(regular, np.clip(d_pr, 0, 65535), 0).astype(np.uint32)

    if n and int(pb.bq.max()) > MAX_BQ:
        raise PileupError("base quality %d > %d: not a Phred value a BAM can hold (0..93); the device's "
                          "error-probability table stops at %d" % (int(pb.bq.max()), MAX_BQ, MAX_BQ))
    flags = (is_r2.astype(np.uint32) * FL_R2 | rev.astype(np.uint32) * FL_REV
             | mm_ok.astype(np.uint32) * FL_MMOK | kind << KIND_SHIFT)
    # in-deletion reads get minBQ as their quality (smCounter.py:418) already here
    bq = np.where(kind == KIND_INDEL_GAP, np.uint32(params.minBQ), pb.bq.astype(np.uint32)).astype(np.uint32)
    meta = pb.allele.astype(np.uint32) | bq << 8 | flags << 16 | pb.mq.astype(np.uint32) << 24
    dist = d_bc | d_pr << 16
    # read class (frag plane bits 27-31): the tally predicates evaluated with this run's parameters
    bq_ok = pb.bq.astype(np.int64) >= params.minBQ
    inc = (bq_ok | (kind == KIND_INDEL_GAP)) & (pb.mq.astype(np.int64) >= params.minMQ) & mm_ok      # incCond, :378
    rclass = read_class(kind, rev, is_r2, inc, bq_ok, d_bc <= 20, d_pr <= params.primerDist)

    # --- per-locus counts: barcodes, fragments (allMT / allFrag, smCounter.py:482-483)
    loci = np.zeros(n_loci, LOCUS_DTYPE)
    n_umi = np.zeros(n_loci, np.int64)
    n_frag = np.zeros(n_loci, np.int64)
    slot = np.zeros(n, np.uint32)
    if n:
        np.maximum.at(n_umi, locus_of, pb.umi.astype(np.int64) + 1)
        um = int(n_umi.max()) + 1
        ukey = locus_of * um + pb.umi.astype(np.int64)
        uk, inv = np.unique(ukey, return_inverse=True)           # sorted: by locus, then barcode
        fmax = np.zeros(len(uk), np.int64)
        np.maximum.at(fmax, inv, pb.frag.astype(np.int64) + 1)
        kl = uk // um
        np.add.at(n_frag, kl, fmax)
        excl = np.cumsum(fmax) - fmax                            # exclusive, over all loci
        locus_first = np.zeros(n_loci, np.int64)
        first_key = np.concatenate([[True], kl[1:] != kl[:-1]])
        locus_first[kl[first_key]] = excl[first_key]
        base_local = excl - locus_first[kl]
        slot = (base_local[inv] + pb.frag.astype(np.int64)).astype(np.uint32)

    # --- padded CSR: each locus starts on a READ_ALIGN boundary; within the locus reads are sorted
    # barcode-major (barcode, fragment slot, pileup order): a stable sort, so the order of the reads
    # of one fragment - the only order vc() depends on - is kept
    padded = (lens + READ_ALIGN - 1) // READ_ALIGN * READ_ALIGN
    off = np.zeros(n_loci + 1, np.int64)
    off[1:] = np.cumsum(padded)
    n_slots = int(off[-1])
    if n:
        order = np.lexsort((np.arange(n), slot.astype(np.int64), locus_of))     # last key is primary
        dst = np.empty(n, np.int64)
        dst[order] = off[:-1][locus_of[order]] + (np.arange(n, dtype=np.int64) - pb.read_off[:-1][locus_of[order]])
    else:
        dst = np.zeros(0, np.int64)

    def plane(src):
        out = np.zeros(n_slots, np.uint32)
        out[dst] = src
        return out

    # first read of every barcode, relative to its locus (+ a closing entry per locus)
    umi_off = np.zeros(n_loci + 1, np.int64)
    umi_off[1:] = np.cumsum(n_umi + 1)
    umi_start = np.zeros(int(umi_off[-1]), np.uint32)
    if n:
        cnt = np.zeros(int(umi_off[-1]), np.int64)
        np.add.at(cnt, umi_off[:-1][locus_of] + pb.umi.astype(np.int64) + 1, 1)
        # cumulative within each locus: entry u+1 accumulates reads of barcodes <= u
        csum = np.cumsum(cnt)
        base = np.repeat(csum[umi_off[:-1]], n_umi + 1)
        umi_start[:] = (csum - base).astype(np.uint32)
    loci["read_off4"] = off[:-1] // READ_ALIGN
    loci["umi_off"] = umi_off[:-1]
    loci["n_reads"] = lens
    loci["n_umi"] = n_umi
    loci["n_frag"] = n_frag
    loci["flags"] = param_fingerprint(params) << LF_FP_SHIFT
    for l in range(n_loci):
        tab = pb.alleles[l]
        if len(tab) > MAX_ALLELES:
            raise PileupError("%s:%d has %d distinct alleles; the device path handles at most %d"
                              % (pb.chrom[l], pb.pos[l], len(tab), MAX_ALLELES))
        assert tuple(tab[:6]) == BASE_ALLELES
        loci["ref_allele"][l] = ref_allele_id(pb.ref[l], tab)
        loci["n_alleles"][l] = len(tab)
        mask = 0
        for a, s in enumerate(tab):
            if len(s) == 1:
                mask |= 1 << a
        loci["snp_mask"][l] = mask

    # --- the reference's down-sampling (smCounter.py:485-498), when the barcode texts are known: bcDict's keys
    # are the barcodes with an included read, inserted in order of that read; more keys than ds -> py2
    # random.sample seeded with the position string.  Dropped keys are marked in umi_start (bit 31).
    if pb.umi_names is not None and n and params.ds > 0:      # (ds <= 0: usedMT = 0, Zero_Coverage anyway)
        from .py2compat import py2_downsample_barcodes
        for l in np.nonzero(n_umi > params.ds)[0]:
            s = pb.locus_slice(int(l))
            ui = pb.umi[s][inc[s]]
            _, first = np.unique(ui, return_index=True)
            order = ui[np.sort(first)]                               # barcodes by first included read
            if len(order) <= params.ds:
                continue
            names = pb.umi_names[int(l)]
            kept = set(py2_downsample_barcodes(str(int(pb.pos[l])), [names[int(u)] for u in order], params.ds))
            o = int(umi_off[l])
            for u in order:
                if names[int(u)] not in kept:
                    umi_start[o + int(u)] |= USTART_DROPPED
            loci["flags"][l] |= LF_SAMPLED

    return DeviceBatch(loci=loci, meta=plane(meta), umi=plane(pb.umi), frag=plane(slot | rclass << FRAG_CLASS_SHIFT),
                       dist=plane(dist), umi_start=umi_start, chrom=list(pb.chrom), pos=pb.pos.copy(),
                       ref=list(pb.ref), alleles=[list(t) for t in pb.alleles],
                       umi_names=None if pb.umi_names is None else [list(t) for t in pb.umi_names])
