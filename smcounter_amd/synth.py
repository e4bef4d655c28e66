"""Seeded synthetic pileups of stated depth (BASELINE.json configs C2-C5).

Per locus: `n_umi` barcodes, each with `rpb` reads arranged as fragments of which a fraction
`p_overlap` have both mates on the locus (R1 forward + R2 reverse) and the rest a single R1 or
R2; reads randomly interleaved (coordinate-sort stand-in); barcode / fragment ids relabelled by
first appearance.  True allele per barcode is the reference base (optionally an alt at a small
fraction of loci), per-read sequencing error, a sprinkle of in-deletion reads and one insertion
and one deletion-start allele per locus, base qualities / MAPQ / mismatch counts / read lengths /
soft clips / query positions drawn from fixed small distributions.

The reference genome is the periodic string whose 1-based position p holds "ACGT"[p % 4]
(`CyclicRef`), so deletion alleles and reference bases are consistent without a FASTA file.
"""
from __future__ import annotations

import dataclasses

import numpy as np

from .pileup import (F_HAS_NM, F_READ1, F_READ2, F_REVERSE, PileupBatch, BASE_ALLELES)

_LETTERS = "ACGT"
# allele ids are in atgc order: A=0, T=1, G=2, C=3
_LETTER_TO_ID = {"A": 0, "T": 1, "G": 2, "C": 3}
_REF_ID_BY_PMOD4 = np.array([_LETTER_TO_ID[c] for c in _LETTERS], np.uint8)   # pos % 4 -> id
_ID_TO_LETTER = np.array(list("ATGC"))


class CyclicRef(object):
    """Reference provider: base at 1-based position p is "ACGT"[p % 4]."""

    def __init__(self, length=1 << 31):
        self._len = length

    def get_reference_length(self, chrom):
        return self._len

    def fetch(self, chrom, start, end):
        start = max(0, start)
        end = min(end, self._len)
        if end <= start:
            return ""
        p = np.arange(start + 1, end + 1)
        return "".join(_LETTERS[i] for i in (p % 4))


class StringRef(object):
    """Reference provider over in-memory chromosome strings."""

    def __init__(self, chroms):
        self._c = {k: v.upper() for k, v in chroms.items()}

    def get_reference_length(self, chrom):
        return len(self._c[chrom])

    def fetch(self, chrom, start, end):
        return self._c[chrom][max(0, start):end]


@dataclasses.dataclass(frozen=True)
class SynthConfig:
    name: str
    n_loci: int
    n_umi: int
    rpb: int
    seed: int
    start_pos: int = 1_000_000
    chrom: str = "chrS"
    p_overlap: float = 0.43
    p_err: float = 1e-3
    p_gap: float = 2e-3         # read is inside a deletion ('DEL')
    p_ins: float = 1e-3         # insertion-start allele
    p_delstart: float = 1e-3    # deletion-start allele
    alt_locus_frac: float = 0.0
    alt_af: float = 0.0
    p_n: float = 2e-4           # base called N

    @property
    def depth(self) -> int:
        return self.n_umi * self.rpb


CONFIGS = {
    # BASELINE.json configs[1..4]; seeds 20170410 + config index
    "C2": SynthConfig("C2", 10_000, 30, 10, 20170412),
    "C3": SynthConfig("C3", 200_000, 50, 60, 20170413),
    "C4": SynthConfig("C4", 1_000_000, 50, 60, 20170414),
    "C5": SynthConfig("C5", 100_000, 133, 60, 20170415, alt_locus_frac=0.01, alt_af=0.005),
    # the shape of the reference's own example run (example.run-log: 2000 loci, ~ 58 k reads each, 3612 barcodes
    # x 8.6 reads, README.md / run.example.sh) and a mid-depth panel shape; not BASELINE configs, used by
    # scripts/quick_perf.py to watch the deep-locus launch classes (512 / 1024 threads, global tables)
    "X1": SynthConfig("X1", 2_000, 3600, 9, 20170420),
    "X2": SynthConfig("X2", 20_000, 600, 10, 20170421),
    # shapes either side of the workgroup-size thresholds (on-chip tables per locus: ~17 / 30 / 42 / 51 KB)
    "X6": SynthConfig("X6", 20_000, 100, 60, 20170425),
    "X7": SynthConfig("X7", 15_000, 800, 10, 20170426),
    "X4": SynthConfig("X4", 10_000, 1200, 10, 20170423),
    "X5": SynthConfig("X5", 8_000, 1450, 10, 20170424),
    "X8": SynthConfig("X8", 6_000, 2000, 10, 20170427),       # ~71 KB
    # the depth of the reference's own example run (58 k reads, ~7 k barcodes): beyond the LDS -> global-table path
    "X9": SynthConfig("X9", 1_000, 6500, 9, 20170428),
    "X10": SynthConfig("X10", 300, 20000, 9, 20170429),       # 180 k reads per locus
    # C3's shape with a 10 % variant at 30 % of the loci: a third of the rows goes through k_filter_loci
    "X3": SynthConfig("X3", 40_000, 50, 60, 20170422, alt_locus_frac=0.3, alt_af=0.1),
    # the statistics of the reference's own example run (2000 loci, mean 58 k reads and 4,162 barcodes per
    # locus, 13.9 reads per barcode, 602 of 2000 loci reaching filterVariants): 4162 barcodes x 14 reads, a variant at 30 % of
    # the loci
    "EX": SynthConfig("EX", 2_000, 4162, 14, 20170430, alt_locus_frac=0.3, alt_af=0.1),
}


def params_for(cfg: SynthConfig):
    from .params import VcParams
    return VcParams(minBQ=20, minMQ=30, mtDepth=cfg.n_umi, rpb=float(cfg.rpb), hpLen=8,
                    mismatchThr=6.0, mtDrop=0, maxMT=0, primerDist=2)


def generate(cfg: SynthConfig, lo: int = 0, hi: int = None, return_counts: bool = False):
    """Loci [lo, hi) of the config.  Each locus draws from its own seed sequence child only through
    the chunk RNG below, so a given (cfg, lo, hi) is reproducible; chunks are independent."""
    hi = cfg.n_loci if hi is None else hi
    nl = hi - lo
    U, B = cfg.n_umi, cfg.rpb
    R = U * B
    rng = np.random.Generator(np.random.PCG64([cfg.seed, lo, hi]))
    pos = cfg.start_pos + np.arange(lo, hi, dtype=np.int64)
    ref_id = _REF_ID_BY_PMOD4[pos % 4]                                   # (nl,)

    # ---- fragment structure per barcode, original (l, u, j) order
    f0 = max(1, int(round(B / (1.0 + cfg.p_overlap))))
    k_pairs = np.minimum(rng.binomial(f0, cfg.p_overlap, size=(nl, U)), B // 2)   # (nl,U)
    j = np.arange(B)[None, None, :]
    in_pair = j < 2 * k_pairs[:, :, None]
    frag_orig = np.where(in_pair, j // 2, j - k_pairs[:, :, None])                  # (nl,U,B)
    single_r2 = rng.random((nl, U, B)) < 0.5
    is_r2 = np.where(in_pair, (j % 2) == 1, single_r2)
    flip = rng.random((nl, U, B)) < 0.1
    is_rev = is_r2 ^ flip

    # ---- alleles
    true_id = np.broadcast_to(ref_id[:, None], (nl, U)).copy()
    if cfg.alt_locus_frac > 0:
        alt_locus = rng.random(nl) < cfg.alt_locus_frac
        # fixed transition: A<->G, C<->T  (ids A0 T1 G2 C3)
        trans = np.array([2, 3, 0, 1], np.uint8)
        alt_umi = (rng.random((nl, U)) < cfg.alt_af) & alt_locus[:, None]
        true_id = np.where(alt_umi, trans[true_id], true_id)
    allele = np.broadcast_to(true_id[:, :, None], (nl, U, B)).copy()
    err = rng.random((nl, U, B)) < cfg.p_err
    shift = rng.integers(1, 4, size=(nl, U, B), dtype=np.uint8)
    allele = np.where(err, (allele + shift) % 4, allele).astype(np.uint8)
    carries_alt = allele != ref_id[:, None, None]
    ev = rng.random((nl, U, B))
    is_gap = ev < cfg.p_gap
    is_ins = (ev >= cfg.p_gap) & (ev < cfg.p_gap + cfg.p_ins)
    is_dst = (ev >= cfg.p_gap + cfg.p_ins) & (ev < cfg.p_gap + cfg.p_ins + cfg.p_delstart)
    is_n = (ev >= 1.0 - cfg.p_n)
    allele = np.where(is_n, 4, allele)
    allele = np.where(is_gap, 5, allele)
    # per-locus table: ids 6 / 7 assigned to whichever of INS / DEL-start occurs (first wins id 6)
    allele = np.where(is_ins, 254, allele)     # placeholders, fixed after the shuffle
    allele = np.where(is_dst, 253, allele)

    bq = rng.choice(np.array([12, 25, 30, 37, 40], np.uint8), size=(nl, U, B),
                    p=[.03, .07, .2, .4, .3])
    mq = np.where(rng.random((nl, U, B)) < 0.02, 20, 60).astype(np.uint8)
    mism = rng.choice(np.array([0, 1, 2], np.uint32), size=(nl, U, B), p=[.8, .15, .05])
    mism = mism + (carries_alt & ~is_gap & ~is_ins & ~is_dst & ~is_n)
    mism = np.where(rng.random((nl, U, B)) < 0.01, 8, mism).astype(np.uint32)
    n_indel = (is_gap | is_ins | is_dst).astype(np.uint32)
    nm = mism + n_indel
    qlen = rng.integers(100, 151, size=(nl, U, B), dtype=np.uint32)
    left_sp = np.where(rng.random((nl, U, B)) < 0.05,
                       rng.integers(1, 11, size=(nl, U, B)), 0).astype(np.uint32)
    qalen = qlen - left_sp
    qpos = (left_sp + (rng.random((nl, U, B)) * qalen).astype(np.uint32)).astype(np.int32)
    indel = np.where(is_ins, 1, np.where(is_dst, -1, 0)).astype(np.int32)
    flag = (np.where(is_r2, F_READ2, F_READ1) | np.where(is_rev, F_REVERSE, 0)
            | F_HAS_NM).astype(np.uint8)

    # ---- random interleave within each locus
    order = rng.permuted(np.broadcast_to(np.arange(R, dtype=np.int32), (nl, R)), axis=1)
    rows = np.arange(nl)[:, None]
    newpos = np.empty((nl, R), np.int32)
    newpos[rows, order] = np.arange(R, dtype=np.int32)[None, :]
    newpos3 = newpos.reshape(nl, U, B)

    # barcode ids by first appearance
    first_u = newpos3.min(axis=2)                                      # (nl,U)
    umi_rank = np.empty((nl, U), np.uint32)
    umi_rank[rows, np.argsort(first_u, axis=1)] = np.arange(U, dtype=np.uint32)[None, :]
    # fragment ids within barcode by first appearance
    n_frag_u = B - k_pairs                                             # (nl,U) fragments per barcode
    big = np.int32(R + 1)
    # first position of original fragment f: pairs occupy reads (2f, 2f+1), singles read f + k
    npad = np.concatenate([newpos3, np.full((nl, U, 1), big, np.int32)], axis=2)
    f_idx = np.arange(B)[None, None, :]
    kk = k_pairs[:, :, None]
    pair_first = np.minimum(np.take_along_axis(npad, np.minimum(2 * f_idx, B), axis=2),
                            np.take_along_axis(npad, np.minimum(2 * f_idx + 1, B), axis=2))
    single_first = np.take_along_axis(npad, np.minimum(f_idx + kk, B), axis=2)
    ffirst = np.where(f_idx < kk, pair_first, single_first)           # (nl,U,B), big when f >= nfrag
    frank = np.empty((nl, U, B), np.uint32)
    fo = np.argsort(ffirst, axis=2, kind="stable")
    np.put_along_axis(frank, fo, np.broadcast_to(np.arange(B, dtype=np.uint32), (nl, U, B)), axis=2)
    frag_new = np.take_along_axis(frank, frag_orig, axis=2)
    umi_new = np.broadcast_to(umi_rank[:, :, None], (nl, U, B))

    def shuf(x):
        return np.take_along_axis(np.ascontiguousarray(x).reshape(nl, R), order, axis=1).reshape(-1)

    allele_s = shuf(allele).reshape(nl, R)
    # allele table: INS gets the lower id if it appears first, else DEL-start
    has_ins = (allele_s == 254)
    has_dst = (allele_s == 253)
    first_ins = np.where(has_ins.any(axis=1), has_ins.argmax(axis=1), R + 1)
    first_dst = np.where(has_dst.any(axis=1), has_dst.argmax(axis=1), R + 1)
    ins_id = np.where(first_ins < first_dst, 6, 7)
    dst_id = np.where(first_dst < first_ins, 6, np.where(first_ins <= R, 7, 6))
    allele_s = np.where(has_ins, ins_id[:, None], allele_s)
    allele_s = np.where(has_dst, dst_id[:, None], allele_s).astype(np.uint8)

    ref_letters = _ID_TO_LETTER[ref_id]
    alleles = []
    base = list(BASE_ALLELES)
    pl = pos.tolist()
    fi, fd = first_ins.tolist(), first_dst.tolist()
    for l in range(nl):
        if fi[l] > R and fd[l] > R:
            alleles.append(base)
            continue
        r = ref_letters[l]
        p = pl[l]
        ins_s = "INS|%s|%s%s" % (r, r, _LETTERS[(p + 2) % 4])
        del_s = "DEL|%s%s|%s" % (r, _LETTERS[(p + 1) % 4], r)
        extra = []
        if fi[l] <= R and fd[l] <= R:
            extra = [ins_s, del_s] if fi[l] < fd[l] else [del_s, ins_s]
        elif fi[l] <= R:
            extra = [ins_s]
        else:
            extra = [del_s]
        alleles.append(base + extra)

    read_off = np.arange(nl + 1, dtype=np.int64) * R
    pb = PileupBatch(
        chrom=[cfg.chrom] * nl, pos=pos, ref=ref_letters.tolist(), alleles=alleles,
        read_off=read_off,
        umi=shuf(umi_new).astype(np.uint32), frag=shuf(frag_new).astype(np.uint32),
        flag=shuf(flag), mq=shuf(mq), nm=shuf(nm), n_indel=shuf(n_indel), left_sp=shuf(left_sp),
        qlen=shuf(qlen), qalen=shuf(qalen), qpos=shuf(qpos), indel=shuf(indel),
        is_del=shuf(is_gap), allele=allele_s.reshape(-1), bq=shuf(bq))
    if return_counts:
        return pb, (np.full(nl, U, np.int64), n_frag_u.sum(axis=1).astype(np.int64))
    return pb


# ---------------------------------------------------------------------------------------------
# Stress generator: small loci that walk every branch of vc()/filterVariants (edge cases the
# reference handles: empty and all-filtered loci, N bases, discordant and triple-aligned
# fragments, indel alleles, strand / read-end / primer-end clustering, bi-allelic loci,
# homopolymer and low-complexity context, unflagged reads, missing NM tags).
# ---------------------------------------------------------------------------------------------
STRESS_SCENARIOS = ("plain", "snp", "snp_sb", "snp_endcluster", "snp_primer", "snp_lowq", "biallelic",
                    "het_ins", "het_del", "gap", "discord", "shallow", "empty", "allfail", "multi",
                    "ndominant", "single")


def stress_reference(seed=7):
    """One chromosome with random sequence, homopolymer runs and a dinucleotide repeat."""
    rng = np.random.Generator(np.random.PCG64(seed))
    parts = []
    for k in range(40):
        parts.append("".join(rng.choice(list("ACGT"), size=60)))
        if k % 4 == 1:
            parts.append("A" * 12)
        if k % 4 == 2:
            parts.append("AC" * 14)
        if k % 4 == 3:
            parts.append("GGGGGGGGG")
    return {"chrT": "".join(parts)}


def generate_stress(n_loci: int, seed: int, scenarios=STRESS_SCENARIOS, chroms=None,
                    max_umi: int = 60, deep: bool = False):
    chroms = chroms or stress_reference()
    seq = chroms["chrT"]
    rng = np.random.Generator(np.random.PCG64([seed, n_loci]))
    cols = {k: [] for k in ("umi", "frag", "flag", "mq", "nm", "n_indel", "left_sp", "qlen", "qalen",
                            "qpos", "indel", "is_del", "allele", "bq")}
    chrom, pos, ref, alleles, off = [], [], [], [], [0]
    assert n_loci <= len(seq) - 60
    positions = 30 + rng.permutation(len(seq) - 60)[:n_loci]    # distinct 1-based positions
    for l in range(n_loci):
        sc = scenarios[l % len(scenarios)]
        p = int(positions[l])
        r = seq[p - 1]
        table = list(BASE_ALLELES)

        def aid(s):
            if s not in table:
                table.append(s)
            return table.index(s)

        others = [c for c in "ATGC" if c != r]
        alt = others[int(rng.integers(0, 3))]
        alt2 = [c for c in others if c != alt][int(rng.integers(0, 2))]
        ins_key = "INS|%s|%s%s" % (r, r, "".join(rng.choice(list("ACGT"), size=int(rng.integers(1, 4)))))
        dlen = int(rng.integers(1, 4))
        del_key = "DEL|%s%s|%s" % (r, seq[p:p + dlen], r)
        U = int(rng.integers(1, max_umi + 1))
        if sc == "shallow":
            U = int(rng.integers(1, 5))
        if sc == "empty":
            U = 0
        if sc == "single":
            U = 1
        if deep:
            U = int(rng.integers(300, 900))
        reads = []      # (umi_label, frag_label, dict)
        for u in range(U):
            nfr = int(rng.integers(1, 13)) if sc != "single" else 1
            # true allele of this barcode
            x = rng.random()
            true = r
            if sc in ("snp", "snp_sb", "snp_endcluster", "snp_primer", "snp_lowq") and x < 0.25:
                true = alt
            elif sc == "biallelic":
                true = alt if x < 0.5 else alt2
            elif sc == "het_ins" and x < 0.5:
                true = ins_key
            elif sc == "het_del" and x < 0.5:
                true = del_key
            elif sc == "gap" and x < 0.4:
                true = "DEL"
            elif sc == "ndominant" and x < 0.3:
                true = "N"
            elif sc == "multi":
                true = [r, alt, alt2, ins_key, del_key, "DEL"][int(rng.integers(0, 6))]
            for f in range(nfr):
                y = rng.random()
                nal = 1 if y < 0.45 else (2 if y < 0.93 else 3)
                if sc == "discord":
                    nal = 2
                if sc == "single":
                    nal = 1
                for k in range(nal):
                    a = true
                    e = rng.random()
                    perr = 0.45 if (sc == "discord" and k == 1) else 0.03
                    if e < perr:
                        a = ["A", "T", "G", "C", "N", "DEL", ins_key, del_key][int(rng.integers(0, 8))]
                    is_r2 = (k % 2 == 1) if nal > 1 else bool(rng.random() < 0.5)
                    rev = is_r2 ^ bool(rng.random() < 0.15)
                    if sc == "snp_sb" and a == alt:
                        rev = True
                    if sc == "snp_sb" and a == r:
                        rev = bool(rng.random() < 0.5)
                    qlen = int(rng.integers(60, 151))
                    lsp = int(rng.integers(1, 9)) if rng.random() < 0.1 else 0
                    qalen = qlen - lsp - (int(rng.integers(1, 6)) if rng.random() < 0.1 else 0)
                    qpos = lsp + int(rng.integers(0, qalen))
                    if sc == "snp_endcluster" and a == alt:
                        # put the variant near the barcode end: R1 fwd / R2 rev -> qpos small
                        near = int(rng.integers(0, 15))
                        far_side = (is_r2 and not rev) or ((not is_r2) and rev)
                        qpos = lsp + (qalen - 1 - near if far_side else near)
                    if sc == "snp_primer" and a == alt and is_r2:
                        near = int(rng.integers(0, 3))
                        qpos = lsp + (qalen - near if rev else near)
                        qpos = min(qpos, lsp + qalen - 1) if not rev else qpos
                    bqv = int(rng.choice([2, 10, 15, 19, 20, 21, 25, 30, 33, 37, 40, 41]))
                    if sc == "snp_lowq" and a == alt and rng.random() < 0.6:
                        bqv = int(rng.integers(2, 20))
                    mqv = int(rng.choice([0, 10, 29, 30, 31, 60, 60, 60, 60, 60]))
                    if sc == "allfail":
                        mqv = int(rng.integers(0, 30))
                    kind_ins = a.startswith("INS|")
                    kind_dst = a.startswith("DEL|")
                    is_gap = a == "DEL"
                    n_ind = (len(a.split("|")[2]) - 1) if kind_ins else ((len(a.split("|")[1]) - 1) if kind_dst else (1 if is_gap else 0))
                    mm = int(rng.choice([0, 0, 0, 1, 1, 2, 3, 5, 7, 9]))
                    has_nm = rng.random() > 0.05
                    fl = (F_READ2 if is_r2 else F_READ1)
                    z = rng.random()
                    if z < 0.03 and reads:
                        fl = 0                              # neither flag: pairOrder carried over
                    elif z < 0.05:
                        fl = F_READ1 | F_READ2
                    fl |= (F_REVERSE if rev else 0) | (F_HAS_NM if has_nm else 0)
                    reads.append((u, f, dict(
                        flag=fl, mq=mqv, nm=(mm + n_ind) if has_nm else 0, n_indel=n_ind, left_sp=lsp,
                        qlen=qlen, qalen=qalen, qpos=qpos,
                        indel=n_ind if kind_ins else (-n_ind if kind_dst else 0), is_del=is_gap,
                        allele=aid(a), bq=bqv)))
        perm = rng.permutation(len(reads))
        # keep a flagged read first so the reference's pairOrder is defined (smCounter.py:359-381)
        reads = [reads[i] for i in perm]
        for i, rd in enumerate(reads):
            if rd[2]["flag"] & (F_READ1 | F_READ2):
                reads[0], reads[i] = reads[i], reads[0]
                break
        else:
            if reads:
                reads[0][2]["flag"] |= F_READ1
        umap, fmap = {}, {}
        for (u, f, d) in reads:
            uu = umap.setdefault(u, len(umap))
            fm = fmap.setdefault(uu, {})
            ff = fm.setdefault(f, len(fm))
            cols["umi"].append(uu)
            cols["frag"].append(ff)
            for k, v in d.items():
                cols[k].append(v)
        chrom.append("chrT")
        pos.append(p)
        ref.append(r)
        alleles.append(table)
        off.append(off[-1] + len(reads))
    dt = dict(umi=np.uint32, frag=np.uint32, flag=np.uint8, mq=np.uint8, nm=np.uint32, n_indel=np.uint32,
              left_sp=np.uint32, qlen=np.uint32, qalen=np.uint32, qpos=np.int32, indel=np.int32,
              is_del=bool, allele=np.uint8, bq=np.uint8)
    return PileupBatch(chrom=chrom, pos=np.array(pos, np.int64), ref=ref, alleles=alleles,
                       read_off=np.array(off, np.int64),
                       **{k: np.array(v, dt[k]) for k, v in cols.items()}), chroms


# ---------------------------------------------------------------------------------------------
# Native generator (smcounter_amd/csrc/smc_synth.cpp): same workload definition, written straight
# into the HBM layout by a thread pool; used for the full-size configs.
# ---------------------------------------------------------------------------------------------
def generate_native(cfg: SynthConfig, lo: int = 0, hi: int = None, params=None, nthreads: int = 0):
    """-> smcounter_amd.features.DeviceBatch for loci [lo, hi) of `cfg`."""
    import ctypes
    import os
    from . import build
    from .features import DeviceBatch, LOCUS_DTYPE

    class Cfg(ctypes.Structure):
        _fields_ = [("n_loci_total", ctypes.c_int64), ("n_umi", ctypes.c_int32), ("rpb", ctypes.c_int32),
                    ("seed", ctypes.c_uint64), ("start_pos", ctypes.c_int64),
                    ("p_overlap", ctypes.c_double), ("p_err", ctypes.c_double), ("p_gap", ctypes.c_double),
                    ("p_ins", ctypes.c_double), ("p_delstart", ctypes.c_double), ("p_n", ctypes.c_double),
                    ("alt_locus_frac", ctypes.c_double), ("alt_af", ctypes.c_double),
                    ("mismatch_thr", ctypes.c_double), ("min_bq", ctypes.c_int32), ("min_mq", ctypes.c_int32),
                    ("primer_dist", ctypes.c_int32), ("pad_", ctypes.c_int32)]

    hi = cfg.n_loci if hi is None else hi
    params = params or params_for(cfg)
    lib = ctypes.CDLL(build.build_synth())
    lib.smc_synth_slots.restype = ctypes.c_int64
    c = Cfg(cfg.n_loci, cfg.n_umi, cfg.rpb, cfg.seed, cfg.start_pos, cfg.p_overlap, cfg.p_err, cfg.p_gap,
            cfg.p_ins, cfg.p_delstart, cfg.p_n, cfg.alt_locus_frac, cfg.alt_af, params.mismatchThr,
            params.minBQ, params.minMQ, params.primerDist, 0)
    n = hi - lo
    slots = lib.smc_synth_slots(ctypes.byref(c), ctypes.c_int64(lo), ctypes.c_int64(hi))
    planes = [np.empty(slots, np.uint32) for _ in range(4)]
    umi_start = np.zeros(n * (cfg.n_umi + 1), np.uint32)
    loci = np.zeros(n, LOCUS_DTYPE)
    extra = np.zeros(n, np.uint8)
    nthreads = nthreads or min(32, os.cpu_count() or 1)
    rc = lib.smc_synth_generate(ctypes.byref(c), ctypes.c_int64(lo), ctypes.c_int64(hi),
                                *[p.ctypes.data_as(ctypes.c_void_p) for p in planes],
                                umi_start.ctypes.data_as(ctypes.c_void_p),
                                loci.ctypes.data_as(ctypes.c_void_p), extra.ctypes.data_as(ctypes.c_void_p),
                                ctypes.c_int(nthreads))
    assert rc == 0
    pos = cfg.start_pos + np.arange(lo, hi, dtype=np.int64)
    ref = _ID_TO_LETTER[_REF_ID_BY_PMOD4[pos % 4]].tolist()
    base = list(BASE_ALLELES)
    alleles = [base] * n
    for l in np.nonzero(extra)[0].tolist():
        r, p, e = ref[l], int(pos[l]), int(extra[l])
        ins_s = "INS|%s|%s%s" % (r, r, _LETTERS[(p + 2) % 4])
        del_s = "DEL|%s%s|%s" % (r, _LETTERS[(p + 1) % 4], r)
        if (e & 3) == 3:
            alleles[l] = base + ([ins_s, del_s] if e & 4 else [del_s, ins_s])
        elif e & 1:
            alleles[l] = base + [ins_s]
        else:
            alleles[l] = base + [del_s]
    return DeviceBatch(loci=loci, meta=planes[0], umi=planes[1], frag=planes[2], dist=planes[3],
                       umi_start=umi_start, chrom=[cfg.chrom] * n, pos=pos, ref=ref, alleles=alleles)


# ---------------------------------------------------------------------------------------------------------------
# Synthetic ALIGNMENTS (the input of the device plane builder): csrc/smc_synth.cpp smc_synth_alignments
ALN_CHROM = "chrS"


def aln_ref_fetch(start: int, end: int) -> str:
    """Reference letters of 0-based [start, end): 1-based position p holds "ACGT"[p % 4] (CyclicRef)."""
    p = np.arange(start + 1, end + 1)
    return "".join(_LETTERS[i] for i in (p % 4))


def generate_alignments(cfg: SynthConfig, n_loci: int = None, params=None, nthreads: int = 0, host_array=None,
                        p_ins_aln: float = 0.02, p_del_aln: float = 0.02, p_clip: float = 0.05):
    """A run of `n_loci` consecutive loci at cfg's depth shape (n_umi barcodes x rpb reads per locus; cfg.alt_locus_frac of the
    positions variant sites at which a molecule carries the transition with probability cfg.alt_af) as the decoder would hand
    it to smc_build_planes: dict(aln, cig, bq (+ its views seq, qual), loc, nl, n_slots, n_bc, n_pair, status, reads, start0).
    `host_array(name, dtype, count)` may provide the arrays (page-locked staging)."""
    import ctypes as C
    import os
    from . import build
    from .abi import DEV_ALN_DTYPE, DEV_LOCUS_DTYPE

    class ACfg(C.Structure):
        _fields_ = [("n_loci", C.c_int64), ("start0", C.c_int64), ("n_umi", C.c_int32), ("rpb", C.c_int32), ("seed", C.c_uint64),
                    ("p_overlap", C.c_double), ("p_err", C.c_double), ("p_ins_aln", C.c_double), ("p_del_aln", C.c_double),
                    ("p_clip", C.c_double), ("mismatch_thr", C.c_double), ("alt_locus_frac", C.c_double), ("alt_af", C.c_double)]

    n_loci = cfg.n_loci if n_loci is None else n_loci
    params = params or params_for(cfg)
    lib = C.CDLL(build.build_synth())
    alloc_t = C.CFUNCTYPE(None, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.POINTER(C.c_void_p))
    lib.smc_synth_alignments.restype = C.c_int64
    lib.smc_synth_alignments.argtypes = [C.POINTER(ACfg), alloc_t, C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int32),
                                         C.POINTER(C.c_int32), C.c_int]
    got = {}
    mk = host_array or (lambda name, dtype, count: np.empty(count, dtype))

    def alloc(ctx, n_aln, n_cig, n_seq, nl, out):
        got["aln"] = mk("aln", DEV_ALN_DTYPE, n_aln)
        got["cig"] = mk("cig", np.uint32, max(1, n_cig))
        got["bq"] = mk("bq", np.uint8, 2 * max(1, n_seq))             # (letter, quality) byte pairs: smc_build_in.bq
        got["seq"], got["qual"] = got["bq"][0::2], got["bq"][1::2]      # (views: the letters, the qualities)
        got["loc"] = mk("loc", DEV_LOCUS_DTYPE, nl)
        for k, name in enumerate(("aln", "cig", "bq", "loc")):
            out[k] = got[name].ctypes.data
    start0 = cfg.start_pos - 1                                     # the pileup generator's first locus, 0-based
    c = ACfg(n_loci, start0, cfg.n_umi, cfg.rpb, cfg.seed, cfg.p_overlap, cfg.p_err, p_ins_aln, p_del_aln, p_clip,
             float(params.mismatchThr), cfg.alt_locus_frac, cfg.alt_af)
    n_slots, n_bc, n_pair = C.c_int64(0), C.c_int32(0), C.c_int32(0)
    nthreads = nthreads or min(32, len(os.sched_getaffinity(0)))
    n = lib.smc_synth_alignments(C.byref(c), alloc_t(alloc), None, C.byref(n_slots), C.byref(n_bc), C.byref(n_pair), nthreads)
    if n < 0:
        raise RuntimeError("smc_synth_alignments failed (%d)" % n)
    got.update(nl=n_loci, n_slots=n_slots.value, n_bc=n_bc.value, n_pair=n_pair.value, status=0, reads=n, start0=start0)
    return got


def alignments_to_bam(A: dict, path: str, l0: int, l1: int, fasta_path: str = None):
    """Write the alignments of run `A` that cover loci [l0, l1) as a BAM (+ .bai, and the cyclic reference as a FASTA when asked):
    the same reads through the real decoder - what bench.py checks the device plane builder against.  -> (chrom, first
    1-based position, last)."""
    from . import bamio
    w0, w1 = int(A["loc"]["w0"][l0]), int(A["loc"]["w1"][l1 - 1])
    recs = []
    aln, cig, seq, qual = A["aln"], A["cig"], A["seq"], A["qual"]
    for i in range(w0, w1):
        a = aln[i]
        ops = [(int(w) & 15, int(w) >> 4) for w in cig[int(a["cig_off"]):int(a["cig_off"]) + int(a["n_cig"])]]
        n_ind = sum(l for op, l in ops if op in (1, 2))
        so, ls = int(a["seq_off"]), int(a["l_seq"])
        # mismatches the generator meant (it sets the mismatch-ok bit from them): 0 when ok, else above any threshold
        fl = int(a["oflag"])
        mism = 0 if fl & 16 else 200
        recs.append(dict(tid=0, pos=int(a["pos"]), qname="q%d:x:B%d:y" % (int(a["pair_gid"]), int(a["bc_gid"])),
                         flag=(0x40 if fl & 1 else 0x80) | (0x10 if fl & 4 else 0) | 1, mapq=int(a["mapq"]), cigar=ops,
                         seq=seq[so:so + ls].tobytes().decode(), qual=qual[so:so + ls].tolist(), nm=min(255, mism + n_ind)))
    end = int(A["start0"]) + int(A["nl"]) + 400
    bamio.write_bam(path, [(ALN_CHROM, end)], recs)
    bamio.write_bai(path)
    if fasta_path:
        with open(fasta_path, "w") as fh:
            fh.write(">%s\n" % ALN_CHROM)
            s = aln_ref_fetch(0, end)
            for i in range(0, len(s), 60):
                fh.write(s[i:i + 60] + "\n")
    return ALN_CHROM, int(A["start0"]) + l0 + 1, int(A["start0"]) + l1
