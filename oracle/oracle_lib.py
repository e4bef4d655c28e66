"""ctypes loader for oracle/libsmc_oracle.so (the C restatement).  TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg - never by the product."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-s", "-C", HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(HERE, "libsmc_oracle.so")
        if not os.path.exists(path):
            build()
        _LIB = ctypes.CDLL(path)
        _LIB.smc_oracle_call_batch.restype = ctypes.c_int
        _LIB.smc_oracle_call_batch_ds.restype = ctypes.c_int
        _LIB.smc_oracle_call_batch_full.restype = ctypes.c_int
        _LIB.smc_oracle_fisher.restype = None
    return _LIB


def call_batch(db, cparams, row_dtype, return_fragile=False, return_pi_all=False):
    """db: smcounter_amd.features.DeviceBatch; cparams: ctypes smc_params; -> structured rows
    (and, on request, the per-locus count of barcodes whose consensus hinges on rounding, and the prediction index of
    every allele key [n_loci, 64], NaN where the allele is not a key - abi.compare_rows uses it to recognise order
    flips between PI-tied alleles that the rows themselves do not carry).  The batch's umi_start is passed along:
    it carries the host's down-sampling marks (SMC_LF_SAMPLED loci)."""
    L = lib()
    assert L.smc_oracle_row_size() == row_dtype.itemsize
    rows = np.zeros(db.n_loci, row_dtype)
    loci = np.ascontiguousarray(db.loci)
    fragile = np.zeros(db.n_loci, np.int32)
    pi_all = np.full((db.n_loci, 64), np.nan) if return_pi_all else None
    ptr = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    rc = L.smc_oracle_call_batch_full(ctypes.byref(cparams), ptr(loci), ctypes.c_int64(db.n_loci), ptr(db.meta), ptr(db.umi),
                                      ptr(db.frag), ptr(db.dist), ptr(np.ascontiguousarray(db.umi_start)), ptr(rows),
                                      ptr(fragile), ptr(pi_all) if return_pi_all else None)
    if rc != 0:
        raise RuntimeError("smc_oracle_call_batch_full failed: %d" % rc)
    out = (rows,)
    if return_fragile:
        out += (fragile,)
    if return_pi_all:
        out += (pi_all,)
    return out if len(out) > 1 else rows


def philox(ctr, key):
    """Philox4x32-10 as oracle/smc_oracle.c restates it (known-answer tests)."""
    L = lib()
    c, k, o = np.array(ctr, np.uint32), np.array(key, np.uint32), np.zeros(4, np.uint32)
    L.smc_oracle_philox(c.ctypes.data_as(ctypes.c_void_p), k.ctypes.data_as(ctypes.c_void_p), o.ctypes.data_as(ctypes.c_void_p))
    return [int(x) for x in o]


def philox_marks(db, cparams, pos, seed=0, ident=None):
    """The non-parity down-sampling of loci over the barcode cap (include/smcounter_hip.h: smc_philox_marks) on a DeviceBatch:
    -> (loci with SMC_LF_SAMPLED set where it was applied, umi_start with the dropped keys marked) - copies."""
    L = lib()
    loci = np.ascontiguousarray(db.loci).copy()
    us = np.ascontiguousarray(db.umi_start).copy()
    pos = np.ascontiguousarray(pos, np.int64)
    ptr = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    rc = L.smc_oracle_philox_marks(ctypes.byref(cparams), ptr(loci), ctypes.c_int64(len(loci)), ptr(pos), ptr(np.ascontiguousarray(db.meta)),
                                   ptr(np.ascontiguousarray(db.umi)), ptr(us),
                                   ptr(np.ascontiguousarray(ident, np.uint64)) if ident is not None else None, ctypes.c_uint64(seed))
    if rc != 0:
        raise RuntimeError("smc_oracle_philox_marks failed: %d" % rc)
    return loci, us


def call_batch_mt(db, cparams, row_dtype, n_threads, return_fragile=False, return_pi_all=False):
    """The same C restatement over contiguous locus ranges on n_threads host threads (ctypes releases the
    GIL during the call; the library keeps no state).  Used by bench.py for the all-cores C baseline and by the
    full-size parity checks (optionally with the fragile counts / the PI of every allele key, as call_batch)."""
    import threading
    L = lib()
    rows = np.zeros(db.n_loci, row_dtype)
    loci = np.ascontiguousarray(db.loci)
    fragile = np.zeros(db.n_loci, np.int32)
    pi_all = np.full((db.n_loci, 64), np.nan) if return_pi_all else None
    us = np.ascontiguousarray(db.umi_start)
    n_threads = max(1, min(int(n_threads), max(1, db.n_loci)))
    bounds = [db.n_loci * t // n_threads for t in range(n_threads + 1)]
    errs = []
    vp = ctypes.c_void_p

    def work(lo, hi):
        if hi <= lo:
            return
        rc = L.smc_oracle_call_batch_full(ctypes.byref(cparams), vp(loci.ctypes.data + lo * loci.dtype.itemsize),
                                          ctypes.c_int64(hi - lo), vp(db.meta.ctypes.data), vp(db.umi.ctypes.data),
                                          vp(db.frag.ctypes.data), vp(db.dist.ctypes.data), vp(us.ctypes.data),
                                          vp(rows.ctypes.data + lo * row_dtype.itemsize), vp(fragile.ctypes.data + 4 * lo),
                                          vp(pi_all.ctypes.data + 512 * lo) if return_pi_all else None)
        if rc != 0:
            errs.append(rc)
    th = [threading.Thread(target=work, args=(bounds[t], bounds[t + 1])) for t in range(n_threads)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    if errs:
        raise RuntimeError("smc_oracle_call_batch_full failed: %r" % errs)
    out = (rows,)
    if return_fragile:
        out += (fragile,)
    if return_pi_all:
        out += (pi_all,)
    return out if len(out) > 1 else rows


def fisher(a, b, c, d):
    o, p = ctypes.c_double(), ctypes.c_double()
    lib().smc_oracle_fisher(ctypes.c_int64(a), ctypes.c_int64(b), ctypes.c_int64(c), ctypes.c_int64(d),
                            ctypes.byref(o), ctypes.byref(p))
    return o.value, p.value


_ALN = None


def aln_planes(A, params, l0=0, l1=None, n_threads=1):
    """oracle/aln_planes.c: the raw-field planes of loci [l0, l1) of a run of alignments (the dict synth.generate_alignments /
    the decoder make: aln, cig, seq, qual, loc, start0, nl), built the reference's way on the host cores -> a DeviceBatch
    (its string context - chromosome, reference letters, allele texts - left for the caller to fill where it prints)."""
    global _ALN
    import threading
    from smcounter_amd import features, synth
    if _ALN is None:
        path = os.path.join(HERE, "libaln_planes.so")
        if not os.path.exists(path):
            build()
        _ALN = ctypes.CDLL(path)
        _ALN.smc_aln_planes.restype = ctypes.c_int
    l1 = int(A["nl"]) if l1 is None else int(l1)
    l0 = int(l0)
    loc = np.ascontiguousarray(A["loc"])
    s0 = int(loc["slot_off"][l0])
    s1 = int(loc["slot_off"][l1 - 1]) + ((int(loc["n"][l1 - 1]) + 3) // 4 * 4)
    n_slots, nl = s1 - s0, l1 - l0
    planes = [np.zeros(n_slots, np.uint32) for _ in range(4)]
    umi_start = np.zeros(n_slots + nl + 1, np.uint32)
    loci = np.zeros(int(A["nl"]), features.LOCUS_DTYPE)
    lo = int(A["start0"])
    refseq = np.frombuffer(A["refseq"] if "refseq" in A else synth.aln_ref_fetch(lo, lo + int(A["nl"])).encode(), np.uint8)
    fp = features.param_fingerprint(params)
    vp = ctypes.c_void_p
    # slot_base / umi_base such that locus l0 starts at slot 0 / entry 0 (modulo 2^32, as the C side computes them)
    slot_base = (-s0) & 0xFFFFFFFF
    umi_base = (-(s0 + l0)) & 0xFFFFFFFF
    errs = []
    n_threads = max(1, min(int(n_threads), nl))
    bounds = [l0 + nl * t // n_threads for t in range(n_threads + 1)]
    arrs = [np.ascontiguousarray(A[k]) for k in ("aln", "cig", "seq", "qual")]

    def work(a, b):
        if b <= a:
            return
        rc = _ALN.smc_aln_planes(vp(arrs[0].ctypes.data), vp(arrs[1].ctypes.data), vp(arrs[2].ctypes.data), vp(arrs[3].ctypes.data),
                                 vp(loc.ctypes.data), vp(refseq.ctypes.data), ctypes.c_int32(lo), ctypes.c_int32(a), ctypes.c_int32(b),
                                 ctypes.c_int32(params.minBQ), ctypes.c_int32(params.minMQ), ctypes.c_int32(params.primerDist),
                                 ctypes.c_uint32(fp), ctypes.c_uint32(slot_base), ctypes.c_uint32(umi_base),
                                 vp(planes[0].ctypes.data), vp(planes[1].ctypes.data), vp(planes[2].ctypes.data), vp(planes[3].ctypes.data),
                                 vp(umi_start.ctypes.data), vp(loci.ctypes.data))
        if rc != 0:
            errs.append(rc)
    th = [threading.Thread(target=work, args=(bounds[t], bounds[t + 1])) for t in range(n_threads)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    if errs:
        raise RuntimeError("smc_aln_planes failed: %r" % errs)
    pos = np.arange(lo + l0 + 1, lo + l1 + 1, dtype=np.int64)
    return features.DeviceBatch(loci=loci[l0:l1].copy(), meta=planes[0], umi=planes[1], frag=planes[2], dist=planes[3],
                                umi_start=umi_start, chrom=[synth.ALN_CHROM] * nl, pos=pos,
                                ref=[chr(c) for c in refseq[l0:l1]], alleles=[[] for _ in range(nl)])
