"""Pure-Python restatement of smCounter's vc() worker, driven the way main() drives it.

TEST / BASELINE INFRASTRUCTURE ONLY (oracle/): used by tests/test_vc_port.py and by bench.py's CPU leg
as the stand-in for "smCounter.py's own multiprocessing CPU path" - the reference's Python cannot travel
to the GPU box, so its algorithm is restated here in plain dict/loop Python over the same per-read
integer arrays the device consumes, and run like main() runs vc(): `multiprocessing.Pool(nCPU)`, one
`apply_async` task per locus, results collected in order (smCounter.py:683-685).

Fidelity: tests/test_vc_port.py checks it against the golden vectors produced by the reference itself.
Follows smCounter.py:316-600 (scan, UMI / fragment bookkeeping, calProb :26-98, PI / consensus, ranking,
filterVariants :182-269 with scipy.stats.fisher_exact) and returns the same smc_row fields as a dict.
"""
from __future__ import annotations

import math
import multiprocessing
from collections import defaultdict

import numpy as np

PCR_NO_ERROR = 1.0 - 3e-5
N_ID, GAP_ID = 4, 5
T_CNT, T_FWD, T_REV, T_LOWQ, T_R1N, T_R1LE, T_R2N, T_R2BCLE, T_R2PRLE, T_CONC, T_DISC = range(11)
R8 = {0: 0, 1: 5, 2: 6, 3: 2, 4: 7, 5: 4}
R32 = {0: 0, 1: 21, 2: 6, 3: 2, 4: 15, 5: 20}


def cal_prob(frags, mt_drop):
    """frags: list of [allele, prob, paired].  -> {allele: posterior}   (smCounter.py:26-98)"""
    if len(frags) <= mt_drop:
        return {0: 0.0, 1: 0.0, 2: 0.0, 3: 0.0}
    exist = sorted({f[0] for f in frags})
    uniq = list(exist)
    for b in (0, 1, 2, 3):
        if len(uniq) >= 4:
            break
        if b not in uniq:
            uniq.append(b)
    uniq.sort()
    prod = {b: 1.0 for b in uniq}
    cnt = defaultdict(int)
    right = 1.0
    for base, prob, paired in frags:
        p = prob if paired else 0.1
        prod[base] *= 1.0 - p
        cnt[base] += 1
        for c in uniq:
            if c != base:
                prod[c] *= p
        right *= 1.0 - p
    pcr = {c: 10.0 ** (-6.0 * ((cnt[c] + 0.5) / (len(frags) + 0.5 * len(uniq)))) for c in uniq}
    tmp, total = {}, 0.0
    for key in uniq:
        if key in exist:
            tmp[key] = PCR_NO_ERROR * prod[key] + right * min(pcr[c] for c in uniq if c != key)
        else:
            t = right
            for c in exist:
                if c != key:
                    t *= pcr[c]
            tmp[key] = t
        total += tmp[key]
    return {key: (0.0 if total <= 0 else tmp[key] / total) for key in uniq}


def _fisher(table):
    import scipy.stats
    r = scipy.stats.fisher_exact(table)
    return float(r[0]), float(r[1])


def _filters(alt, ref, snp_mask, used, tal, mtcnt, strong, cvg):
    """filterVariants minus HP / LowC (smCounter.py:182-269) -> (bits, vmf_lt_099, p-values)."""
    f = 0
    ta, tr = tal[alt], tal[ref]
    is_snp = (snp_mask >> alt) & 1
    if used < 5:
        f |= 1
    if strong[alt] < 2:
        f |= 2
    vmf = 1.0 * mtcnt[alt] / used < 0.99
    af = 100.0 * ta[T_CNT] / cvg
    pairs = ta[T_DISC] + ta[T_CONC]
    p_sb = p_r1 = p_r2 = p_pr = float("nan")
    if pairs >= 1000 and 1.0 * ta[T_DISC] / pairs >= 0.5:
        f |= 16
    elif af <= 60.0:
        o, p_sb = _fisher([[tr[T_REV], tr[T_FWD]], [ta[T_REV], ta[T_FWD]]])
        if p_sb < 0.00001 and (o >= 50 or o <= 1.0 / 50):
            f |= 32
    bq_alt = 1.0 * ta[T_LOWQ] / ta[T_CNT] if (is_snp and ta[T_LOWQ] > 0) else 0.0
    if bq_alt > 0.4:
        f |= 64
    if is_snp:
        o, p_r1 = _fisher([[tr[T_R1LE], tr[T_R1N] - tr[T_R1LE]], [ta[T_R1LE], ta[T_R1N] - ta[T_R1LE]]])
        if p_r1 < 0.001 and o < 0.05 and af <= 60.0:
            f |= 128
        o, p_r2 = _fisher([[tr[T_R2BCLE], tr[T_R2N] - tr[T_R2BCLE]], [ta[T_R2BCLE], ta[T_R2N] - ta[T_R2BCLE]]])
        if p_r2 < 0.001 and o < 0.05 and af <= 60.0:
            f |= 256
        le, gt = ta[T_R2PRLE], ta[T_R2N] - ta[T_R2PRLE]
        o, p_pr = _fisher([[tr[T_R2PRLE], tr[T_R2N] - tr[T_R2PRLE]], [le, gt]])
        if le + gt > 0 and (1.0 * le / (le + gt) >= 0.98 or (p_pr < 0.001 and o < 1.0 / 20)):
            f |= 512
    return f, int(vmf), (p_sb, p_r1, p_r2, p_pr)


def vc_locus(meta, umi, frag, dist, ref, n_alleles, snp_mask, min_bq, min_mq, mt_drop, primer_dist, ds, smt, dropped=None):
    """One locus.  Arrays are the device planes restricted to the locus (see smcounter_amd/features.py)."""
    tal = defaultdict(lambda: [0] * 11)
    bc = {}                      # barcode -> {slot: [allele, prob, paired]}  (bcDict, insertion-ordered)
    all_frags = defaultdict(set)
    cvg = 0
    for m, u, s, d in zip(meta.tolist(), umi.tolist(), frag.tolist(), dist.tolist()):
        a, bq, fl, mq = m & 0xff, (m >> 8) & 0xff, (m >> 16) & 0xff, m >> 24
        kind, r2, rev = (fl >> 3) & 3, fl & 1, fl & 2
        dbc, dpr = d & 0xffff, d >> 16
        cvg += 1
        t = tal[a]
        if kind == 0 and bq < min_bq:
            t[T_LOWQ] += 1
        if kind == 1:
            bq = min_bq
        inc = bq >= min_bq and mq >= min_mq and bool(fl & 4)
        t[T_CNT] += 1
        if kind != 1:
            t[T_REV if rev else T_FWD] += 1
        if kind == 0 and inc:
            if not r2:
                t[T_R1N] += 1
                t[T_R1LE] += dbc <= 20
            else:
                t[T_R2N] += 1
                t[T_R2BCLE] += dbc <= 20
                t[T_R2PRLE] += dpr <= primer_dist
        all_frags[u].add(s)
        if inc:
            one = bc.setdefault(u, {})
            prob = pow(10.0, -bq / 10.0)
            cur = one.get(s)
            if cur is None:
                one[s] = [a, prob, False]
            elif a == cur[0] or a == N_ID:
                cur[1] = max(prob, cur[1])
                cur[2] = True
                if a == cur[0]:
                    t[T_CONC] += 1
            else:
                del one[s]
                t[T_DISC] += 1
    return _finish(tal, bc, all_frags, cvg, ref, snp_mask, mt_drop, ds, smt, dropped)


def _finish(tal, bc, all_frags, cvg, ref, snp_mask, mt_drop, ds, smt, dropped=None):
    """smCounter.py:482-600 on what the scan of a locus left: tallies per allele id, bcDict (barcode -> {fragment: [allele id, error
    probability, paired]}), allBcDict (barcode -> fragments), coverage.  (Shared by vc_locus - integer planes in - and
    vc_port_objects.vc_locus_objects - pysam-like objects in.)"""
    row = dict(status=0, cvg=cvg, all_mt=len(all_frags), all_frag=sum(len(v) for v in all_frags.values()),
               dp=[tal[k][T_CNT] for k in range(4)])
    used = min(ds, len(bc))
    row["used_mt"] = used
    if used == 0:
        row["status"] = 1
        return row
    keys = list(bc.keys())
    if len(bc) > ds:
        row["status"] |= 0x100
        if dropped is not None:
            # the reference's random.sample (smCounter.py:496-498), run by the host (features.py / py2compat.py)
            keys = [u for u in keys if u not in dropped]
            assert len(keys) == used
        else:
            keys = sorted(keys)[:ds]        # documented non-parity stand-in
    fin = defaultdict(float)
    mtcnt, strong = defaultdict(int), defaultdict(int)
    mt = [0, 0, 0, 0]
    used_frag = 0
    for u in keys:
        frs = list(bc[u].values())
        used_frag += len(frs)
        post = cal_prob(frs, mt_drop)
        pred = {}
        for k, p in post.items():
            x = 1.0 - p
            pred[k] = -math.log10(x) if x > 0.0 else 16.0
            fin[k] += pred[k]
        mx = max(pred.values())
        top = [k for k, v in pred.items() if v == mx]
        if len(top) == 1:
            mtcnt[top[0]] += 1
            if pred[top[0]] > smt:
                strong[top[0]] += 1
        elif len(frs) == 1:
            mtcnt[frs[0][0]] += 1
        for i, thr in enumerate((3, 5, 7, 10)):
            mt[i] += len(frs) >= thr
    nkeys = len(fin)
    rank = (lambda a: (R8 if nkeys <= 5 else R32)[a] if a < 6 else 64 + a)
    order = sorted(fin.keys(), key=lambda a: (-fin[a], rank(a)))
    best, second = order[0], order[1]
    row.update(used_frag=used_frag, mt3=mt[0], mt5=mt[1], mt7=mt[2], mt10=mt[3], n_touched=nkeys,
               max_allele=best, second_allele=second, umt=[mtcnt[k] for k in range(4)],
               vsm=[strong[k] for k in range(4)], pi=[fin[k] for k in range(4)])

    def cand(a, run_filter):
        c = dict(allele=a, pi=fin[a], vdp=tal[a][T_CNT], vmt=mtcnt[a], vsm=strong[a], flt_applied=0, flt=0,
                 vmf_lt_099=0, p=(float("nan"),) * 4)
        is_filterable = ((snp_mask >> a) & 1) or a != GAP_ID
        if run_filter and fin[a] >= 5 and is_filterable:
            c["flt_applied"] = 1
            c["flt"], c["vmf_lt_099"], c["p"] = _filters(a, ref, snp_mask, used, tal, mtcnt, strong, cvg)
        return c
    alt = second if best == ref else best
    row["cand0"] = cand(alt, True)
    row["biallelic"] = int(best != ref and second != ref and 1.0 * mtcnt[best] / used >= 0.45
                           and 1.0 * mtcnt[second] / used >= 0.45)
    row["cand1"] = cand(second, True) if row["biallelic"] else None
    return row


def _task(args):
    return vc_locus(*args)


def _task_own_input(args):
    """One locus whose input the WORKER makes itself - as the reference's worker opens the BAM and piles up its own locus
    (smCounter.py:275, :316): the parent sends a config name and a locus index, not the reads."""
    cfg_name, l, prm = args
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    from smcounter_amd import synth
    from smcounter_amd.params import VcParams
    P = VcParams(**prm)
    db = synth.generate_native(synth.CONFIGS[cfg_name], l, l + 1, P, nthreads=1)
    L = db.loci[0]
    n = int(L["n_reads"])
    return vc_locus(db.meta[:n], db.umi[:n], db.frag[:n] & 0x07FFFFFF, db.dist[:n], int(L["ref_allele"]), int(L["n_alleles"]),
                    int(L["snp_mask"]), P.minBQ, P.minMQ, P.mtDrop, P.primerDist, P.ds, P.smt, None)


def _task_own_input_chunk(args):
    cfg_name, l0, l1, prm = args
    return [_task_own_input((cfg_name, l, prm)) for l in range(l0, l1)]


def call_config_chunked(cfg_name, params, lo, hi, pool, chunk):
    """The same with `chunk` consecutive loci per task: what the cores do when the parent's per-task round trip (which the
    reference's 0.02 - 2 s per locus never notices) is out of the way."""
    prm = dict(minBQ=params.minBQ, minMQ=params.minMQ, mtDepth=params.mtDepth, rpb=params.rpb, hpLen=params.hpLen,
               mismatchThr=params.mismatchThr, mtDrop=params.mtDrop, maxMT=params.maxMT, primerDist=params.primerDist)
    results = [pool.apply_async(_task_own_input_chunk, ((cfg_name, l, min(hi, l + chunk), prm),)) for l in range(lo, hi, chunk)]
    return [r for part in results for r in part.get()]


def call_config(cfg_name, params, loci, pool):
    """The port over loci of a synthetic config, one task per locus, every worker generating its own locus."""
    prm = dict(minBQ=params.minBQ, minMQ=params.minMQ, mtDepth=params.mtDepth, rpb=params.rpb, hpLen=params.hpLen,
               mismatchThr=params.mismatchThr, mtDrop=params.mtDrop, maxMT=params.maxMT, primerDist=params.primerDist)
    results = [pool.apply_async(_task_own_input, ((cfg_name, int(l), prm),)) for l in loci]
    return [r.get() for r in results]


_SHARED = {}


def share_batch(db, n_loci, directory=None):
    """The first `n_loci` loci of a batch written where every worker can map them (made BEFORE a timed pass: the workers then
    read their locus's reads from it as the reference's worker reads its BAM region - the making of the input is not what a CPU
    baseline should time).  -> the directory."""
    import os
    import tempfile
    d = tempfile.mkdtemp(prefix="smc_cpu_leg_", dir=directory or ("/dev/shm" if os.path.isdir("/dev/shm") else None))
    end = 4 * int(db.loci["read_off4"][n_loci - 1]) + int(db.loci["n_reads"][n_loci - 1])
    for name in ("meta", "umi", "frag", "dist"):
        np.save(os.path.join(d, name + ".npy"), getattr(db, name)[:end])
    np.save(os.path.join(d, "loci.npy"), db.loci[:n_loci])
    return d


def unshare_batch(d):
    import shutil
    shutil.rmtree(d, ignore_errors=True)


def _task_shared(args):
    """One locus whose reads the worker takes from the shared batch (mapped once per worker)."""
    d, l, prm = args
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    from smcounter_amd.params import VcParams
    B = _SHARED.get(d)
    if B is None:
        B = _SHARED[d] = {k: np.load(os.path.join(d, k + ".npy"), mmap_mode="r") for k in ("meta", "umi", "frag", "dist", "loci")}
    P = VcParams(**prm)
    L = B["loci"][l]
    o, n = 4 * int(L["read_off4"]), int(L["n_reads"])
    return vc_locus(np.asarray(B["meta"][o:o + n]), np.asarray(B["umi"][o:o + n]), np.asarray(B["frag"][o:o + n]) & 0x07FFFFFF,
                    np.asarray(B["dist"][o:o + n]), int(L["ref_allele"]), int(L["n_alleles"]), int(L["snp_mask"]), P.minBQ, P.minMQ,
                    P.mtDrop, P.primerDist, P.ds, P.smt, None)


def call_shared(d, params, loci, pool):
    """The port over loci of a shared batch, one task per locus (smCounter.py:683-685): the parent sends a directory name and a
    locus index."""
    prm = dict(minBQ=params.minBQ, minMQ=params.minMQ, mtDepth=params.mtDepth, rpb=params.rpb, hpLen=params.hpLen,
               mismatchThr=params.mismatchThr, mtDrop=params.mtDrop, maxMT=params.maxMT, primerDist=params.primerDist)
    results = [pool.apply_async(_task_shared, ((d, int(l), prm),)) for l in loci]
    return [r.get() for r in results]


def _noop(x):
    return x


def make_pool(n_cpu):
    """A started, warmed worker pool ("spawn": fresh interpreters, never a fork of a GPU process)."""
    pool = multiprocessing.get_context("spawn").Pool(processes=n_cpu)
    pool.map(_noop, range(4 * n_cpu))
    return pool


def call_batch(db, params, n_cpu=1, loci=None, pool=None):
    """Run the port over loci of a DeviceBatch with a process pool, one task per locus (the reference's
    dispatch, smCounter.py:683-685).  -> list of row dicts in locus order."""
    idx = range(db.n_loci) if loci is None else loci
    tasks = []
    for l in idx:
        L = db.loci[l]
        o, n = 4 * int(L["read_off4"]), int(L["n_reads"])
        dropped = None
        if int(L["flags"]) & 1:
            us = db.umi_start[int(L["umi_off"]):int(L["umi_off"]) + int(L["n_umi"])]
            dropped = set(np.nonzero(us >> 31)[0].tolist())
        tasks.append((db.meta[o:o + n], db.umi[o:o + n], db.frag[o:o + n] & 0x07FFFFFF, db.dist[o:o + n], int(L["ref_allele"]),
                      int(L["n_alleles"]), int(L["snp_mask"]), params.minBQ, params.minMQ, params.mtDrop,
                      params.primerDist, params.ds, params.smt, dropped))
    if n_cpu <= 1 and pool is None:
        return [vc_locus(*t) for t in tasks]
    own = pool is None
    if own:
        pool = make_pool(n_cpu)
    try:
        results = [pool.apply_async(_task, (t,)) for t in tasks]
        return [r.get() for r in results]
    finally:
        if own:
            pool.close()
            pool.join()
