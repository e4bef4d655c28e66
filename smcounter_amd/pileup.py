"""Primary pileup records: what a single-position pileup yields, before feature extraction.

One `PileupBatch` holds the reads covering each of a list of loci (CSR over loci), with exactly
the per-read facts `vc()` pulls out of pysam at smCounter.py:319-448: the two qname fields that
matter (UMI and fragment), SAM flag bits, MAPQ, NM, the CIGAR summary, lengths, the query
position, the indel marker, and the allele the read shows at the locus.

UMI and fragment identity are carried as dense integers instead of strings:

* `umi[i]`  - index of the read's barcode among the locus's barcodes, in order of first
  appearance in the pileup (the key order of `allBcDict`, smCounter.py:463);
* `frag[i]` - index of the read's `readid` in `allBcDict[BC]`, i.e. order of first appearance
  within its barcode (smCounter.py:463-464).

Alleles are ids into a per-locus string table.  Ids 0-5 are fixed:
A, T, G, C (the reference's `atgc` order, smCounter.py:21), N, and 'DEL' (read is inside a
deletion, smCounter.py:416-417); ids >= 6 are the locus's other keys in order of first
appearance: 'INS|r|ra' / 'DEL|rd|r' strings (smCounter.py:374, :396) or any other base letter.
"""
from __future__ import annotations

import dataclasses
from typing import List, Optional

import numpy as np

BASE_ALLELES = ("A", "T", "G", "C", "N", "DEL")
A_, T_, G_, C_, N_, DEL_ = range(6)
N_FIXED_ALLELES = 6
MAX_ALLELES = 64          # device cap per locus (allele id is also a bit index)

# flag bits of PileupBatch.flag
F_READ1, F_READ2, F_REVERSE, F_HAS_NM = 1, 2, 4, 8


@dataclasses.dataclass
class PileupBatch:
    # per locus
    chrom: List[str]
    pos: np.ndarray            # int64, 1-based
    ref: List[str]             # upper-cased reference base (smCounter.py:312-313)
    alleles: List[List[str]]   # allele string table per locus (ids 0-5 fixed)
    read_off: np.ndarray       # int64[n_loci+1]
    # per read
    umi: np.ndarray            # uint32
    frag: np.ndarray           # uint32
    flag: np.ndarray           # uint8  F_* bits
    mq: np.ndarray             # uint8
    nm: np.ndarray             # uint32 NM tag (0 when absent)
    n_indel: np.ndarray        # uint32 sum of I/D CIGAR lengths (smCounter.py:343-344)
    left_sp: np.ndarray        # uint32 leading soft clip (smCounter.py:345-346)
    qlen: np.ndarray           # uint32 query_length
    qalen: np.ndarray          # uint32 query_alignment_length
    qpos: np.ndarray           # int32  query_position (undefined when is_del)
    indel: np.ndarray          # int32  >0 insertion follows, <0 deletion follows
    is_del: np.ndarray         # bool
    allele: np.ndarray         # uint8  id into alleles[locus]
    bq: np.ndarray             # uint8  base quality at qpos
    # optional, per locus: the barcode strings by dense id (needed only to reproduce the reference's
    # down-sampling, which depends on the py2 hash of the barcode text; smCounter.py:496-498)
    umi_names: Optional[List[List[str]]] = None

    @property
    def n_loci(self) -> int:
        return len(self.chrom)

    @property
    def n_reads(self) -> int:
        return int(self.read_off[-1])

    def locus_slice(self, l: int) -> slice:
        return slice(int(self.read_off[l]), int(self.read_off[l + 1]))

    def select(self, idx) -> "PileupBatch":
        """Sub-batch of the given loci (in the given order)."""
        idx = [int(i) for i in idx]
        lens = [int(self.read_off[i + 1] - self.read_off[i]) for i in idx]
        off = np.zeros(len(idx) + 1, np.int64)
        off[1:] = np.cumsum(lens)
        if idx:
            take = np.concatenate([np.arange(self.read_off[i], self.read_off[i + 1]) for i in idx])
        else:
            take = np.zeros(0, np.int64)
        per_read = {f.name: getattr(self, f.name)[take] for f in dataclasses.fields(self)
                    if f.name not in ("chrom", "pos", "ref", "alleles", "read_off", "umi_names")}
        return PileupBatch(chrom=[self.chrom[i] for i in idx], pos=self.pos[idx].copy(),
                           ref=[self.ref[i] for i in idx],
                           alleles=[list(self.alleles[i]) for i in idx], read_off=off,
                           umi_names=None if self.umi_names is None else [list(self.umi_names[i]) for i in idx],
                           **per_read)


def concat(batches: List[PileupBatch]) -> PileupBatch:
    chrom, ref, alleles = [], [], []
    for b in batches:
        chrom += b.chrom
        ref += b.ref
        alleles += b.alleles
    lens = np.concatenate([np.diff(b.read_off) for b in batches]) if batches else np.zeros(0, np.int64)
    off = np.zeros(len(lens) + 1, np.int64)
    off[1:] = np.cumsum(lens)
    per_read = {}
    for f in dataclasses.fields(PileupBatch):
        if f.name in ("chrom", "pos", "ref", "alleles", "read_off", "umi_names"):
            continue
        per_read[f.name] = np.concatenate([getattr(b, f.name) for b in batches])
    names = None
    if batches and all(b.umi_names is not None for b in batches):
        names = [n for b in batches for n in b.umi_names]
    return PileupBatch(chrom=chrom, pos=np.concatenate([b.pos for b in batches]), ref=ref,
                       alleles=alleles, read_off=off, umi_names=names, **per_read)


def allele_kind(s: str) -> str:
    """Variant type of an allele key, as `convertToVcf` classifies it (smCounter.py:103-117)."""
    if len(s) == 1:
        return "SNP"
    if s == "DEL":
        return "SDEL"
    if s.split("|")[0] in ("DEL", "INS"):
        return "INDEL"
    return "."


_PER_READ = ("umi", "frag", "flag", "mq", "nm", "n_indel", "left_sp", "qlen", "qalen", "qpos", "indel",
             "is_del", "allele", "bq")


def save_npz(path: str, pb: PileupBatch, **extra):
    """Compact on-disk form of a batch (used for the golden fixtures under tests/golden/)."""
    import json
    meta = dict(chrom=pb.chrom, ref=pb.ref, alleles=pb.alleles, extra=extra)
    if pb.umi_names is not None:
        meta["umi_names"] = pb.umi_names
    arrays = {k: getattr(pb, k) for k in _PER_READ}
    np.savez_compressed(path, pos=pb.pos, read_off=pb.read_off,
                        meta=np.frombuffer(json.dumps(meta).encode(), np.uint8), **arrays)


def load_npz(path: str):
    import json
    z = np.load(path)
    meta = json.loads(bytes(z["meta"]).decode())
    pb = PileupBatch(chrom=meta["chrom"], pos=z["pos"], ref=meta["ref"], alleles=meta["alleles"],
                     read_off=z["read_off"], umi_names=meta.get("umi_names"), **{k: z[k] for k in _PER_READ})
    return pb, meta["extra"]
