"""The pure-Python restatement of vc() fed the way the REFERENCE's worker is fed: pysam-like objects, one per pileup read.

TEST / BASELINE INFRASTRUCTURE ONLY (oracle/): bench.py's `cpu_baseline_object_adapter` leg and tests/test_vc_port.py.

oracle/vc_port.py consumes the per-read integer planes the device consumes - it skips exactly the work that is most of the
reference's time: attribute access on alignment objects, the read name split and joined (smCounter.py:319-325), the tag list walked
for NM (:329-334), the CIGAR walked for indels and soft clips (:336-349), the mismatch rate (:352-356), string allele keys and
string-keyed dicts (:371-460).  SURVEY.md 8d asks for the CPU baseline to do that work "through an object adapter"; this module is
the adapter (`pileup_objects`: a locus's pileup from a run of synthetic alignments, as objects with the attributes :319-448 read)
and the restatement of :316-479 over it (`vc_locus_objects`), handing the same intermediate state to vc_port._finish (:482-600).

What the adapter cannot carry over from a BAM it makes up consistently: read names ("inst:lane:<pair id>:NN:<barcode>:x" - the
last-but-one field is the barcode, everything before it the read id, :321-325), and an NM tag such that max(0, NM - nIndel) per 100
bases is on the side of mismatchThr the decoder recorded for the alignment.
"""
from __future__ import annotations

import math
from collections import defaultdict

import numpy as np

import vc_port
from vc_port import T_CNT, T_FWD, T_REV, T_LOWQ, T_R1N, T_R1LE, T_R2N, T_R2BCLE, T_R2PRLE, T_CONC, T_DISC

FIXED = {"A": 0, "T": 1, "G": 2, "C": 3, "N": 4, "DEL": 5}
DA_R1, DA_R2, DA_REV, DA_MMOK = 1, 2, 4, 16         # smc_dev_aln.oflag (include/smcounter_hip.h)


class Aln(object):
    """What smCounter.py:319-448 reads of pysam's AlignedSegment."""
    __slots__ = ("query_name", "mapping_quality", "tags", "cigar", "query_length", "is_read1", "is_read2", "is_reverse",
                 "query_sequence", "query_qualities", "query_alignment_length")


class PileupRead(object):
    """... and of pysam's PileupRead."""
    __slots__ = ("alignment", "indel", "is_del", "query_position")


def _resolve(cigar, pos, p, l_seq):
    """samtools' resolve_cigar2 for reference position p of an alignment that starts at pos: (query_position, is_del, indel)."""
    x, y = pos, 0
    qpos, isdel, indel = -1, False, 0
    for ci, (op, ln) in enumerate(cigar):
        ref_op, gap_op = op in (0, 7, 8), op in (2, 3)
        if ref_op or gap_op:
            d = p - x
            if 0 <= d < ln:
                qpos = y + d if ref_op else y
                isdel = gap_op
                if d == ln - 1 and ci + 1 < len(cigar):
                    nop, nln = cigar[ci + 1]
                    indel = nln if nop == 1 else -nln if nop == 2 else 0
            x += ln
            if ref_op:
                y += ln
            if x > p:
                break
        elif op in (1, 4):
            y += ln
    if isdel and indel != 0 and qpos >= l_seq:
        indel = 0
    return qpos, isdel, indel


def alignment_objects(A, mismatch_thr: float, lo: int = 0, hi: int = None):
    """Alignments [lo, hi) of a run (synth.generate_alignments / the decoder's dict) as objects."""
    aln, cig, seq, qual = A["aln"], A["cig"], A["seq"], A["qual"]
    hi = len(aln) if hi is None else hi
    out = []
    for i in range(lo, hi):
        a = aln[i]
        n_cig, co, so, l_seq = int(a["n_cig"]), int(a["cig_off"]), int(a["seq_off"]), int(a["l_seq"])
        cg = [(int(w) & 15, int(w) >> 4) for w in cig[co:co + n_cig]]
        o = Aln()
        fl = int(a["oflag"])
        o.query_name = "SYN:1:%d:NN:BC%07d:x" % (int(a["pair_gid"]), int(a["bc_gid"]))
        o.mapping_quality = int(a["mapq"])
        n_indel = sum(ln for op, ln in cg if op in (1, 2))
        ok = bool(fl & DA_MMOK)
        nm = n_indel if ok else n_indel + int(math.floor(mismatch_thr * l_seq / 100.0)) + 1
        o.tags = [("MD", "0"), ("NM", nm), ("AS", 0)]
        o.cigar = cg
        o.query_length = l_seq
        o.is_read1, o.is_read2, o.is_reverse = bool(fl & DA_R1), bool(fl & DA_R2), bool(fl & DA_REV)
        o.query_sequence = bytes(seq[so:so + l_seq]).decode()
        o.query_qualities = qual[so:so + l_seq].tolist()
        o.query_alignment_length = int(a["qalen"])
        out.append((int(a["pos"]), int(a["end"]), o))
    return out


def pileup_objects(A, l: int, objs=None, base: int = 0):
    """The pileup column of locus l of the run, file order: PileupRead objects (what `samfile.pileup(...)` yields at :316).  `objs`:
    alignment_objects(A, ...) of alignments [base, ...) made once for several loci."""
    loc = A["loc"][l]
    p = int(A["start0"]) + l
    col = []
    for i in range(int(loc["w0"]), int(loc["w1"])):
        pos, end, o = objs[i - base]
        if not (pos <= p < end):
            continue
        qpos, isdel, indel = _resolve(o.cigar, pos, p, o.query_length)
        if qpos < 0:
            continue
        r = PileupRead()
        r.alignment, r.indel, r.is_del, r.query_position = o, indel, isdel, qpos
        col.append(r)
    return col


def vc_locus_objects(pileup, orig_ref: str, ref_after, min_bq, min_mq, mismatch_thr, mt_drop, primer_dist, ds, smt):
    """smCounter.py:316-479 over a pileup of objects (`ref_after(n)`: the n reference letters behind the locus, for a deletion's
    allele key :392-396), then vc_port._finish.  -> the row dict + `alleles` (the locus's allele keys, ids as in the planes)."""
    allele_cnt, forward_cnt, reverse_cnt, low_q = defaultdict(int), defaultdict(int), defaultdict(int), defaultdict(int)
    concord, discord = defaultdict(int), defaultdict(int)
    r1_bc_end, r2_bc_end, r2_primer_end = defaultdict(list), defaultdict(list), defaultdict(list)
    all_bc, bc_dict = defaultdict(list), defaultdict(dict)
    alleles = ["A", "T", "G", "C", "N", "DEL"]
    ids = dict(FIXED)
    cvg = 0
    pair_order = None
    for pr in pileup:
        aln = pr.alignment
        qname = aln.query_name
        qs = qname.split(":")
        readid = ":".join(qs[:-2])
        BC = qs[-2]
        duplex_tag = qs[-3]                                    # noqa: F841  (:325 - read, never used)
        mq = aln.mapping_quality
        NM = 0
        for tag, value in aln.tags:
            if tag == "NM":
                NM = value
                break
        n_indel, cigar_order, left_sp, right_sp = 0, 1, 0, 0
        for op, value in aln.cigar:
            if op == 1 or op == 2:
                n_indel += value
            if cigar_order == 1 and op == 4:
                left_sp = value
            if cigar_order > 1 and op == 4:
                right_sp += value
            cigar_order += 1
        mismatch = max(0, NM - n_indel)
        read_len = aln.query_length
        mm100 = 100.0 * mismatch / read_len if read_len > 0 else 0.0
        if aln.is_read1:
            pair_order = "R1"
        if aln.is_read2:
            pair_order = "R2"
        strand = "Reverse" if aln.is_reverse else "Forward"
        cvg += 1
        regular = False
        if pr.indel > 0:                                       # :371-389
            site = aln.query_sequence[pr.query_position]
            base = "INS|" + site + "|" + site + aln.query_sequence[pr.query_position + 1:pr.query_position + 1 + pr.indel]
            bq = aln.query_qualities[pr.query_position]
            inc = bq >= min_bq and mq >= min_mq and mm100 <= mismatch_thr
        elif pr.indel < 0:                                     # :392-411
            site = aln.query_sequence[pr.query_position]
            base = "DEL|" + site + ref_after(abs(pr.indel)) + "|" + site
            bq = aln.query_qualities[pr.query_position]
            inc = bq >= min_bq and mq >= min_mq and mm100 <= mismatch_thr
        elif pr.is_del:                                        # :416-421
            base, bq = "DEL", min_bq
            inc = bq >= min_bq and mq >= min_mq and mm100 <= mismatch_thr
        else:                                                  # :424-457
            regular = True
            base = aln.query_sequence[pr.query_position]
            bq = aln.query_qualities[pr.query_position]
            inc = bq >= min_bq and mq >= min_mq and mm100 <= mismatch_thr
            if bq < min_bq:
                low_q[base] += 1
            if inc:
                if pair_order == "R1":
                    d = pr.query_position - left_sp
                    if strand == "Reverse":
                        d = aln.query_alignment_length - d
                    r1_bc_end[base].append(d)
                else:
                    rd = pr.query_position - left_sp
                    far = aln.query_alignment_length - rd
                    r2_bc_end[base].append(rd if strand == "Reverse" else far)
                    r2_primer_end[base].append(far if strand == "Reverse" else rd)
        allele_cnt[base] += 1
        if base != "DEL" or regular:
            if not pr.is_del or pr.indel != 0:
                (reverse_cnt if strand == "Reverse" else forward_cnt)[base] += 1
        if base not in ids:
            ids[base] = len(alleles)
            alleles.append(base)
        if readid not in all_bc[BC]:                           # :463-464 (a list scan, as the reference has it)
            all_bc[BC].append(readid)
        if inc:                                                # :467-479
            prob = pow(10.0, -bq / 10.0)
            one = bc_dict[BC]
            cur = one.get(readid)
            if cur is None:
                one[readid] = [base, prob, pair_order]
            elif base == cur[0] or base in ("N", "*"):
                cur[1] = max(prob, cur[1])
                cur[2] = "Paired"
                if base == cur[0]:
                    concord[base] += 1
            else:
                del one[readid]
                discord[base] += 1
    # ---- the intermediate state in vc_port's terms: allele ids, 11 tallies each (the lists become the two counts the filters read)
    tal = defaultdict(lambda: [0] * 11)
    for b, n in allele_cnt.items():
        t = tal[ids[b]]
        t[T_CNT] = n
        t[T_FWD], t[T_REV], t[T_LOWQ] = forward_cnt[b], reverse_cnt[b], low_q[b]
        t[T_R1N], t[T_R1LE] = len(r1_bc_end[b]), sum(1 for d in r1_bc_end[b] if d <= 20)
        t[T_R2N], t[T_R2BCLE] = len(r2_bc_end[b]), sum(1 for d in r2_bc_end[b] if d <= 20)
        t[T_R2PRLE] = sum(1 for d in r2_primer_end[b] if d <= primer_dist)
        t[T_CONC], t[T_DISC] = concord[b], discord[b]
    bc = {B: {rid: [ids[f[0]], f[1], f[2] == "Paired"] for rid, f in one.items()} for B, one in bc_dict.items()}
    all_frags = {B: set(v) for B, v in all_bc.items()}
    snp_mask = sum(1 << i for i, a in enumerate(alleles) if len(a) == 1)
    row = vc_port._finish(tal, bc, all_frags, cvg, ids.get(orig_ref, 255), snp_mask, mt_drop, ds, smt, None)
    row["alleles"] = alleles
    return row


def _task(args):
    return vc_locus_objects(*args)


# ---- the bench's leg: workers map a run of alignments, make the objects of their own loci (as the reference's worker piles up its
# own region - not what is timed) and time the restatement over them
def share_alignments(A, directory=None):
    import os
    import tempfile
    d = tempfile.mkdtemp(prefix="smc_cpu_obj_", dir=directory or ("/dev/shm" if os.path.isdir("/dev/shm") else None))
    for k in ("aln", "cig", "bq", "loc"):
        np.save(os.path.join(d, k + ".npy"), np.ascontiguousarray(A[k]))
    np.save(os.path.join(d, "meta.npy"), np.array([int(A["start0"]), int(A["nl"])], np.int64))
    return d


def _load_shared(d):
    import os
    A = {k: np.load(os.path.join(d, k + ".npy"), mmap_mode="r") for k in ("aln", "cig", "bq", "loc")}
    A["seq"], A["qual"] = A["bq"][0::2], A["bq"][1::2]
    m = np.load(os.path.join(d, "meta.npy"))
    A["start0"], A["nl"] = int(m[0]), int(m[1])
    return A


def _task_chunk(args):
    """loci [l0, l1) of the shared run: objects and pileups first, then the timed pass.  -> (loci, seconds of the pass, sum of cvg)"""
    import os
    import sys
    import time
    d, l0, l1, prm, ref = args
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from smcounter_amd.params import VcParams
    P = VcParams(**prm)
    A = _load_shared(d)
    lo, hi = int(A["loc"]["w0"][l0]), int(A["loc"]["w1"][l1 - 1])
    objs = alignment_objects(A, P.mismatchThr, lo, hi)
    cols = [pileup_objects(A, l, objs, base=lo) for l in range(l0, l1)]
    t0 = time.perf_counter()
    acc = 0
    for l, col in zip(range(l0, l1), cols):
        row = vc_locus_objects(col, ref[l], (lambda n, l=l: ref[l + 1:l + 1 + n]), P.minBQ, P.minMQ, P.mismatchThr, P.mtDrop, P.primerDist, P.ds, P.smt)
        acc += row["cvg"]
    return l1 - l0, time.perf_counter() - t0, acc


def timed_pass(A, params, ref, n_workers, per_worker, pool):
    """-> (loci, seconds = the slowest worker's pass, workers) over the first n_workers x per_worker loci of the run."""
    prm = dict(minBQ=params.minBQ, minMQ=params.minMQ, mtDepth=params.mtDepth, rpb=params.rpb, hpLen=params.hpLen,
               mismatchThr=params.mismatchThr, mtDrop=params.mtDrop, maxMT=params.maxMT, primerDist=params.primerDist)
    d = share_alignments(A)
    try:
        n = min(int(A["nl"]), n_workers * per_worker)
        tasks = [(d, l, min(n, l + per_worker), prm, ref) for l in range(0, n, per_worker)]
        res = pool.map(_task_chunk, tasks, chunksize=1)
    finally:
        import shutil
        shutil.rmtree(d, ignore_errors=True)
    return sum(r[0] for r in res), max(r[1] for r in res), len(tasks), sum(r[2] for r in res)
