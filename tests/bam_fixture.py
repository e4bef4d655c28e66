"""Builds a small coordinate-sorted BAM + FASTA + BED with reads that exercise the pileup:
matches, soft clips, insertions, deletions, reference skips, mate pairs, several barcodes."""
import os

import numpy as np

from smcounter_amd import bamio

M, I, D, N, S = 0, 1, 2, 3, 4


def make_case(tmp, seed=5, n_umi=12, frags_per_umi=5):
    rng = np.random.Generator(np.random.PCG64(seed))
    ref = "".join(rng.choice(list("ACGT"), size=600)) + "A" * 12 + "".join(rng.choice(list("ACGT"), size=388))
    fa = os.path.join(tmp, "ref.fa")
    with open(fa, "w") as fh:
        fh.write(">chrQ test\n")
        for i in range(0, len(ref), 60):
            fh.write(ref[i:i + 60] + "\n")
    recs = []
    snp_pos = 300                     # 0-based; half the barcodes carry an alt here
    alt = "ACGT"[("ACGT".index(ref[snp_pos]) + 1) % 4]
    for u in range(n_umi):
        carries = u % 2 == 0
        for f in range(frags_per_umi):
            start = int(rng.integers(230, 290))
            for mate in (0, 1):
                if mate == 1 and rng.random() < 0.4:
                    continue
                pos = start if mate == 0 else start + int(rng.integers(5, 30))
                kind = int(rng.integers(0, 6))
                if kind == 0:
                    cigar = [(S, 3), (M, 97)]
                elif kind == 1:
                    cigar = [(M, 40), (I, 2), (M, 58)]
                elif kind == 2:
                    cigar = [(M, 50), (D, 3), (M, 50)]
                elif kind == 3:
                    cigar = [(M, 30), (N, 10), (M, 70)]
                else:
                    cigar = [(M, 100)]
                seq, x = [], pos
                for op, l in cigar:
                    if op == M:
                        seq.append(ref[x:x + l]); x += l
                    elif op in (D, N):
                        x += l
                    elif op == I:
                        seq.append("".join(rng.choice(list("ACGT"), size=l)))
                    elif op == S:
                        seq.append("".join(rng.choice(list("ACGT"), size=l)))
                seq = list("".join(seq))
                # place the alt / a sequencing error at the SNP position if covered by an M block
                x, y = pos, 0
                for op, l in cigar:
                    if op == M:
                        if x <= snp_pos < x + l:
                            q = y + snp_pos - x
                            if carries:
                                seq[q] = alt
                            if rng.random() < 0.03:
                                seq[q] = "ACGT"[int(rng.integers(0, 4))]
                        x += l; y += l
                    elif op in (D, N):
                        x += l
                    else:
                        y += l
                qual = rng.choice([15, 25, 30, 37, 40], size=len(seq)).astype(np.uint8)
                flag = (0x40 if mate == 0 else 0x80 | 0x10) | 0x1
                nm = int(rng.integers(0, 3)) + sum(l for op, l in cigar if op in (I, D))
                recs.append(dict(tid=0, pos=pos, qname="inst:1:r%d_%d:NN:UMI%02d:x" % (u, f, u), flag=flag,
                                 mapq=int(rng.choice([20, 60, 60, 60])), cigar=cigar, seq="".join(seq),
                                 qual=qual.tolist(), nm=nm if rng.random() > 0.1 else None))
    recs.sort(key=lambda r: r["pos"])
    bam = os.path.join(tmp, "case.bam")
    bamio.write_bam(bam, [("chrQ", len(ref))], recs, block=8000)
    bed = os.path.join(tmp, "target.bed")
    with open(bed, "w") as fh:
        fh.write("chrQ\t280\t320\nchrQ\t598\t604\n")
    return dict(ref=ref, fasta=fa, bam=bam, bed=bed, records=recs, snp_pos=snp_pos, alt=alt)
