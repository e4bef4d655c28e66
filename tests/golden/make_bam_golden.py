"""Generate the BAM golden fixtures tests/golden/bam_*.npz (BUILD CONTAINER ONLY).

Each fixture holds the BYTES of a small BAM (+ .bai), its FASTA, the target loci, the parameters, and what the reference
itself - /root/reference/smCounter.py through oracle/ref_harness.py - returns for every locus when its pysam calls are
served from that BAM: the stub pysam hands vc() the real records' attributes (query_name, cigar, query_sequence,
query_qualities, flags, NM, ...) and, per pileup read, the (query_position, is_del, indel) of the legacy samtools pileup
engine as smcounter_amd/bamio.py resolves them.  So the reference does its own read-name splitting, CIGAR scan, allele
keys, barcode bookkeeping, down-sampling and filters on these reads; what is pinned is everything downstream of "which
reads cover the locus, at which query position" - the BAM -> planes path (smc_bam_alignments + k_build_planes, or
smc_bam_planes) and the locus kernels.

Cases:  cigars  - two chromosomes, random S/M/I/D/N CIGARs incl. D next to I, hard clips, reads without NM, odd read names;
        deep    - a core of > 8192 reads per locus with shallow flanks, giant and tiny barcodes, indels, soft clips;
        overcap - more barcodes than 2 * mtDepth: the reference's random.sample on barcode texts (py2 emulation);
        deep25k - a core beyond 24,576 reads per locus (the locus kernels' deep class: parts and chunks; the plane builder's
                  multi-launch sort) between shallow flanks, one giant barcode, indels;
        unflagged - alignments flagged neither READ1 nor READ2 among paired ones: the reference's pairOrder is then the value the
                  PREVIOUS pileup read left behind (smCounter.py:359-362).
Usage:  PYTHONHASHSEED=0 python tests/golden/make_bam_golden.py [case ...]
"""
import dataclasses
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import ref_harness  # noqa: E402
import stub_pysam  # noqa: E402
from smcounter_amd import bamio, fasta  # noqa: E402
from smcounter_amd.params import VcParams  # noqa: E402


def reference_rows(bam_path, fa_path, loci, params):
    """The reference's vc_wrapper() per locus, its pysam served from the BAM's records."""
    import builtins
    import contextlib
    import io
    mod = ref_harness._MOD or ref_harness.load_reference()
    ref_harness._MOD = mod
    bam = bamio.BamFile(bam_path)
    fa = fasta.FastaFile(fa_path)
    table, chroms = {}, {}
    for chrom, pos in loci:
        p0 = int(pos) - 1
        reads = []
        for a in bam.fetch(chrom, p0, p0 + 1):
            col = bamio._column(a, p0)
            if col is None:
                continue
            qpos, is_del, indel = col
            cig = [(int(op), int(l)) for op, l in a.cigar]
            reads.append(dict(qname=a.qname, mq=a.mapq, nm=a.nm, has_nm=bool(a.has_nm), cigar=cig, qlen=a.l_seq,
                              qalen=sum(l for op, l in cig if op in (0, 1, 7, 8)), is_read1=bool(a.flag & 0x40),
                              is_read2=bool(a.flag & 0x80), is_reverse=bool(a.flag & 0x10), qpos=int(qpos), seq=a.seq,
                              quals=list(a.qual), indel=int(indel), is_del=bool(is_del)))
        table[(chrom, int(pos))] = reads
        if chrom not in chroms:
            try:
                chroms[chrom] = fa.fetch(chrom, 0, fa.get_reference_length(chrom))
            except Exception:
                chroms[chrom] = ""
    stub_pysam.register_bam("golden.bam", table)
    stub_pysam.register_fasta("golden.fa", chroms)
    out = []
    for chrom, pos in loci:
        ref_harness.CAP.reset()
        with contextlib.redirect_stdout(io.StringIO()):
            row = mod.vc_wrapper("golden.bam", chrom, builtins.str(int(pos)), params.minBQ, params.minMQ, params.mtDepth, params.rpb,
                                 params.hpLen, params.mismatchThr, params.mtDrop, params.maxMT, params.primerDist, "golden.fa")
        assert not row.startswith("Exception thrown!"), row
        out.append(dict(row=row, pi_raw=list(ref_harness.CAP.round2), tie_ambiguous=ref_harness._tie_ambiguous(ref_harness.CAP.final),
                        sampled=bool(ref_harness.CAP.sampled), depth=len(table[(chrom, int(pos))])))
    return out


def emit(name, refs, seqs, recs, loci, params):
    tmp = tempfile.mkdtemp()
    bam, fa_path = os.path.join(tmp, "g.bam"), os.path.join(tmp, "g.fa")
    with open(fa_path, "w") as fh:
        for n in seqs:
            fh.write(">%s\n" % n)
            for i in range(0, len(seqs[n]), 60):
                fh.write(seqs[n][i:i + 60] + "\n")
    recs.sort(key=lambda r: (r["tid"], r["pos"]))
    bamio.write_bam(bam, refs, recs, block=60000)
    bamio.write_bai(bam)
    exp = reference_rows(bam, fa_path, loci, params)
    meta = dict(loci=loci, params=dataclasses.asdict(params), expected=exp)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), bam=np.frombuffer(open(bam, "rb").read(), np.uint8),
                        bai=np.frombuffer(open(bam + ".bai", "rb").read(), np.uint8),
                        fasta=np.frombuffer(open(fa_path, "rb").read(), np.uint8),
                        meta=np.frombuffer(json.dumps(meta).encode(), np.uint8))
    depth = [e["depth"] for e in exp]
    print("%-12s %4d loci, %6d records, depth %d..%d, %d tie-ambiguous, %d down-sampled, %d with an indel allele, %.0f KB" % (
        name, len(loci), len(recs), min(depth), max(depth), sum(e["tie_ambiguous"] for e in exp), sum(e["sampled"] for e in exp),
        sum(("INS" in e["row"] or "DEL" in e["row"].split("\t")[3]) for e in exp), os.path.getsize(os.path.join(HERE, name + ".npz")) / 1e3))


def random_cigar(rng):
    cig = []
    r = rng.rand()
    if r < 0.25:
        cig.append((4, int(rng.randint(1, 8))))
    elif r < 0.30:
        cig.append((5, int(rng.randint(1, 5))))
        if rng.rand() < 0.5:
            cig.append((4, int(rng.randint(1, 4))))
    for _ in range(rng.randint(1, 4)):
        cig.append((0, int(rng.randint(15, 50))))
        r = rng.rand()
        if r < 0.15:
            cig.append((1, int(rng.randint(1, 4))))
        elif r < 0.30:
            cig.append((2, int(rng.randint(1, 6))))
        elif r < 0.36:                                           # a deletion right before an insertion, and the reverse
            cig += [(2, int(rng.randint(1, 4))), (1, int(rng.randint(1, 3)))]
        elif r < 0.42:
            cig += [(1, int(rng.randint(1, 3))), (2, int(rng.randint(1, 4)))]
        elif r < 0.47:
            cig.append((3, int(rng.randint(5, 30))))
    if cig[-1][0] != 0:
        cig.append((0, int(rng.randint(5, 20))))
    if rng.rand() < 0.2:
        cig.append((4, int(rng.randint(1, 6))))
    return cig


def read_from_ref(rng, seq, pos, cig, p_err=0.02):
    """Query bases following the reference through the CIGAR (clips and insertions random), with a few substitutions."""
    out, x = [], pos
    for op, l in cig:
        if op in (0, 7, 8):
            out.append(seq[x:x + l]); x += l
        elif op in (1, 4):
            out.append("".join(rng.choice(list("ACGT"), l)))
        elif op in (2, 3):
            x += l
    s = list("".join(out))
    for i in range(len(s)):
        if rng.rand() < p_err:
            s[i] = "ACGTN"[rng.randint(0, 5)]
    return "".join(s)


def case_cigars():
    rng = np.random.RandomState(20171)
    refs = [("chrA", 60000), ("chrB", 30000)]
    seqs = {n: "".join(rng.choice(list("ACGT"), l)) for n, l in refs}
    # a homopolymer and a low-complexity stretch under two of the target windows (HP / LowC filters)
    seqs["chrA"] = seqs["chrA"][:40010] + "A" * 14 + seqs["chrA"][40024:]
    seqs["chrB"] = seqs["chrB"][:19990] + "AC" * 12 + seqs["chrB"][20014:]
    centres = [(0, 300), (0, 40000), (0, 59900), (1, 20000), (1, 29950)]
    recs = []
    for tid, c in centres:
        name = refs[tid][0]
        for u in range(30):
            umi = "".join(rng.choice(list("ACGT"), 10))
            variant = rng.rand() < 0.3                               # this barcode carries a substitution at the centre
            for fr in range(rng.randint(1, 5)):
                for mate in (0, 1):
                    if mate == 1 and rng.rand() < 0.25:
                        continue
                    cig = random_cigar(rng)
                    pos = max(0, c - int(rng.randint(15, 110)))
                    rl = sum(l for op, l in cig if op in (0, 2, 3, 7, 8))
                    if pos + rl + 5 >= refs[tid][1]:
                        pos = refs[tid][1] - rl - 6
                    s = read_from_ref(rng, seqs[name], pos, cig)
                    qlen = len(s)
                    if variant:
                        # substitute at the query position under c, if the read has one there
                        col = None
                        a = bamio.Alignment(); a.pos, a.cigar, a.l_seq = pos, cig, qlen
                        col = bamio._column(a, c)
                        if col is not None and not col[1]:
                            alt = {"A": "G", "C": "T", "G": "A", "T": "C"}[seqs[name][c]]
                            s = s[:col[0]] + alt + s[col[0] + 1:]
                    weird = rng.rand() < 0.05
                    qn = ("x:y:rd%d_%d_%d:NN:%s:0" if weird else "rd%d_%d_%d:NN:%s:0") % (c, u, fr, umi)
                    recs.append(dict(tid=tid, pos=pos, qname=qn,
                                     flag=(0x40 if mate == 0 else 0x80) | (0x10 if rng.rand() < 0.5 else 0) | 1,
                                     mapq=int(rng.choice([0, 10, 29, 30, 60, 60, 60])), cigar=cig, seq=s,
                                     qual=[int(x) for x in rng.choice([5, 14, 15, 20, 30, 37, 41], qlen)],
                                     nm=None if rng.rand() < 0.1 else int(rng.randint(0, 12))))
    loci = []
    for tid, c in centres:
        loci += [(refs[tid][0], str(p)) for p in range(max(1, c - 35), min(refs[tid][1], c + 35))]
    loci += [("chrA", "50000")]                                       # an empty locus
    emit("bam_cigars", refs, seqs, recs, loci, VcParams(mtDepth=100, rpb=3.0, hpLen=8, minBQ=15, minMQ=20, mismatchThr=8.0, mtDrop=0))


def case_deep():
    rng = np.random.RandomState(20172)
    L = 600
    ref = "".join(rng.choice(list("ACGT"), L))
    recs = []
    depth = 9400
    n_bc = 900
    for i in range(depth // 2 + 400):
        bc = int(rng.randint(0, n_bc)) if i % 3 else int(rng.randint(0, 4))          # four giant barcodes
        deep = i % 40 != 0
        start = 300 + int(rng.randint(0, 5)) if deep else int(rng.randint(180, 420))
        variant = bc % 7 == 0
        for mate in (0, 1):
            pos = start + (0 if mate == 0 else int(rng.randint(0, 6)))
            k = rng.rand()
            if k < 0.015:
                cig = [(0, 22), (1, 2), (0, 34)]
            elif k < 0.03:
                cig = [(0, 25), (2, 3), (0, 33)]
            elif k < 0.06:
                cig = [(4, 4), (0, 54)]
            else:
                cig = [(0, 58)]
            s = read_from_ref(rng, ref, pos, cig, p_err=0.004)
            if variant:
                a = bamio.Alignment(); a.pos, a.cigar, a.l_seq = pos, cig, len(s)
                col = bamio._column(a, 330)
                if col is not None and not col[1]:
                    s = s[:col[0]] + {"A": "G", "C": "T", "G": "A", "T": "C"}[ref[330]] + s[col[0] + 1:]
            recs.append(dict(tid=0, pos=pos, qname="r%d:NN:BC%04d:y" % (i, bc), flag=(0x41 if mate == 0 else 0x91), mapq=int(rng.choice([20, 60, 60])),
                             cigar=cig, seq=s, qual=[int(x) for x in rng.choice([12, 25, 30, 37], len(s))], nm=int(rng.randint(0, 3))))
    loci = [("chrD", str(p)) for p in list(range(296, 304)) + list(range(326, 336)) + list(range(352, 364))]
    emit("bam_deep", [("chrD", L)], {"chrD": ref}, recs, loci, VcParams(mtDepth=3000, rpb=10.0, hpLen=8, mtDrop=1))


def case_overcap():
    rng = np.random.RandomState(20173)
    L = 500
    ref = "".join(rng.choice(list("ACGT"), L))
    recs = []
    for u in range(60):
        umi = "".join(rng.choice(list("ACGT"), 12))
        for fr in range(rng.randint(1, 4)):
            start = 200 + int(rng.randint(0, 30))
            for mate in (0, 1):
                pos = start + (0 if mate == 0 else int(rng.randint(0, 10)))
                cig = [(0, 60)] if rng.rand() < 0.9 else [(4, 3), (0, 30), (1, 1), (0, 26)]
                s = read_from_ref(rng, ref, pos, cig, p_err=0.01)
                recs.append(dict(tid=0, pos=pos, qname="q%d_%d:NN:%s:z" % (u, fr, umi), flag=(0x41 if mate == 0 else 0x91), mapq=60, cigar=cig,
                                 seq=s, qual=[int(x) for x in rng.choice([10, 25, 30, 37], len(s))], nm=int(rng.randint(0, 2))))
    loci = [("chrO", str(p)) for p in range(215, 275)]
    emit("bam_overcap", [("chrO", L)], {"chrO": ref}, recs, loci, VcParams(mtDepth=9, rpb=2.5, hpLen=8))      # ds = 18 < ~60 barcodes


def case_deep25k():
    rng = np.random.RandomState(20174)
    L = 500
    ref = "".join(rng.choice(list("ACGT"), L))
    recs = []
    n_bc = 2600
    for i in range(13400):
        bc = int(rng.randint(0, n_bc)) if i % 5 else 0                                # one giant barcode (a fifth of the reads)
        deep = i % 60 != 0
        start = 250 + int(rng.randint(0, 4)) if deep else int(rng.randint(150, 330))
        variant = bc % 9 == 1
        for mate in (0, 1):
            pos = start + (0 if mate == 0 else int(rng.randint(0, 5)))
            k = rng.rand()
            if k < 0.01:
                cig = [(0, 20), (1, 1), (0, 29)]
            elif k < 0.02:
                cig = [(0, 24), (2, 2), (0, 26)]
            elif k < 0.04:
                cig = [(4, 3), (0, 47)]
            else:
                cig = [(0, 50)]
            s = read_from_ref(rng, ref, pos, cig, p_err=0.003)
            if variant:
                a = bamio.Alignment(); a.pos, a.cigar, a.l_seq = pos, cig, len(s)
                col = bamio._column(a, 270)
                if col is not None and not col[1]:
                    s = s[:col[0]] + {"A": "C", "C": "A", "G": "T", "T": "G"}[ref[270]] + s[col[0] + 1:]
            recs.append(dict(tid=0, pos=pos, qname="r%d:NN:BC%04d:y" % (i, bc), flag=(0x41 if mate == 0 else 0x91), mapq=int(rng.choice([25, 60, 60])),
                             cigar=cig, seq=s, qual=[int(x) for x in rng.choice([12, 25, 30, 37], len(s))], nm=int(rng.randint(0, 2))))
    loci = [("chrE", str(p)) for p in (249, 252, 262, 271, 280, 296, 306)]
    emit("bam_deep25k", [("chrE", L)], {"chrE": ref}, recs, loci, VcParams(mtDepth=3000, rpb=8.0, hpLen=8, mtDrop=1))


def case_unflagged():
    rng = np.random.RandomState(20175)
    L = 400
    ref = "".join(rng.choice(list("ACGT"), L))
    recs = []
    for u in range(40):
        umi = "".join(rng.choice(list("ACGT"), 10))
        for fr in range(rng.randint(1, 5)):
            start = 150 + int(rng.randint(0, 25))
            for mate in (0, 1):
                if mate == 1 and rng.rand() < 0.3:
                    continue
                pos = start + (0 if mate == 0 else int(rng.randint(0, 8)))
                cig = [(0, 60)] if rng.rand() < 0.85 else [(4, 2), (0, 28), (2, 2), (0, 30)]
                s = read_from_ref(rng, ref, pos, cig, p_err=0.01)
                flag = (0x40 if mate == 0 else 0x80) | (0x10 if rng.rand() < 0.5 else 0) | 1
                if rng.rand() < 0.12:
                    flag &= ~0xC0                                             # neither READ1 nor READ2
                recs.append(dict(tid=0, pos=pos, qname="q%d_%d:NN:%s:z" % (u, fr, umi), flag=flag, mapq=60, cigar=cig,
                                 seq=s, qual=[int(x) for x in rng.choice([10, 25, 30, 37], len(s))], nm=int(rng.randint(0, 2))))
    # the first record of the file keeps its flags: a pileup that STARTS with an unflagged read has no pairOrder in the reference
    recs.sort(key=lambda r: r["pos"])
    loci = [("chrU", str(p)) for p in range(160, 225)]
    for _, p in loci:
        p0 = int(p) - 1
        for r in recs:                                                        # (file order = pileup order)
            if r["pos"] <= p0 < r["pos"] + sum(l for op, l in r["cigar"] if op in (0, 2, 3, 7, 8)):
                if not (r["flag"] & 0xC0):
                    r["flag"] |= 0x40
                break
    emit("bam_unflagged", [("chrU", L)], {"chrU": ref}, recs, loci, VcParams(mtDepth=100, rpb=3.0, hpLen=8, primerDist=3))


if __name__ == "__main__":
    todo = sys.argv[1:] or ["cigars", "deep", "overcap", "deep25k", "unflagged"]
    for name in todo:
        globals()["case_" + name]()
