"""Run the ACTUAL reference implementation of vc() on in-memory pileups (build container only).

TEST INFRASTRUCTURE - not part of the product path, never imported by smcounter_amd.

The reference (`/root/reference/smCounter.py`) is Python 2.7 and imports pysam; neither exists
here.  This harness
  1. translates the reference source IN MEMORY with lib2to3 (mechanical py2->py3 fixers only),
  2. injects `oracle/stub_pysam.py` as `pysam`,
  3. executes the translated source as a module, with three names pre-bound in its namespace to
     emulate CPython 2.7 where it leaks into results:
        round  -> half-away-from-zero (smcounter_amd.py2compat.py2_round),
        str    -> 12-significant-digit float printing,
        sorted -> stable sort preceded by py2 dict key order (ties in PI, smCounter.py:534),
  4. wraps scipy.stats.fisher_exact and round to capture the Fisher p-values and the unrounded
     prediction indices of every call.
Nothing from the reference is copied into the repo: this file reads it by path at run time and
only the generated input/output vectors (tests/golden/) are committed.
"""
from __future__ import annotations

import builtins
import io
import os
import sys
import types
import contextlib

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

import stub_pysam  # noqa: E402
from smcounter_amd import py2compat  # noqa: E402
from smcounter_amd.pileup import F_HAS_NM, F_READ1, F_READ2, F_REVERSE  # noqa: E402

REFERENCE = os.environ.get("SMC_REFERENCE", "/root/reference/smCounter.py")


class _Capture(object):
    def __init__(self):
        self.reset()

    def reset(self):
        self.round2 = []     # arguments of round(x, 2) in call order -> PI_A, PI_T, PI_G, PI_C, PI_alt
        self.fisher = []     # (table, oddsratio, pvalue)
        self.calprob = []    # per-UMI posterior dicts, in bcKeys order
        self.final = []      # finalDict.items() as handed to sorted() (smCounter.py:534)
        self.sampled = False # random.sample ran (smCounter.py:497-498)


CAP = _Capture()


def _py2_sorted(iterable, key=None, reverse=False):
    items = list(iterable)
    CAP.final = [(k, float(v)) for k, v in items] if items and isinstance(items[0], tuple) else []
    if items and all(isinstance(it, tuple) and len(it) == 2 and isinstance(it[0], str)
                     for it in items):
        order = py2compat.py2_dict_order([k for k, _ in items])
        rank = {k: i for i, k in enumerate(order)}
        items.sort(key=lambda kv: rank[kv[0]])
    return builtins.sorted(items, key=key, reverse=reverse)


def _py2_round(x, n=0):
    if n == 2:
        CAP.round2.append(float(x))
    return py2compat.py2_round(x, n)


def _py2_str(x=""):
    if isinstance(x, float):
        return py2compat.py2_str_float(x)
    return builtins.str(x)


def load_reference():
    from lib2to3 import refactor
    src = open(REFERENCE).read()
    tool = refactor.RefactoringTool(refactor.get_fixers_from_package("lib2to3.fixes"))
    with contextlib.redirect_stderr(io.StringIO()):
        py3 = builtins.str(tool.refactor_string(src + "\n", "smCounter.py"))
    sys.modules["pysam"] = stub_pysam
    mod = types.ModuleType("smCounter_reference")
    mod.__dict__["round"] = _py2_round
    mod.__dict__["str"] = _py2_str
    mod.__dict__["sorted"] = _py2_sorted
    exec(compile(py3, REFERENCE, "exec"), mod.__dict__)

    import scipy.stats as _ss

    def fisher(table, *a, **k):
        r = _ss.fisher_exact(table, *a, **k)
        CAP.fisher.append(([[int(v) for v in row] for row in table], float(r[0]), float(r[1])))
        return r

    shim = types.SimpleNamespace(stats=types.SimpleNamespace(fisher_exact=fisher))
    mod.__dict__["scipy"] = shim
    real_cal = mod.calProb

    def cal(oneBC, mtDrop):
        out = real_cal(oneBC, mtDrop)
        CAP.calprob.append(dict(out))
        return out

    mod.__dict__["calProb"] = cal

    # random.seed(pos) / random.sample(bcDict.keys(), ds) (smCounter.py:496-498) as CPython 2.7 computes them:
    # the py3 dict hands the keys over in insertion order; py2 would iterate them in hash-slot order and draw
    # with int(random() * n) from a generator seeded with the 64-bit string hash (py2compat.Py2Random)
    class _Py2RandomModule(object):
        def __init__(self):
            self._r = None

        def seed(self, a):
            self._r = py2compat.Py2Random(a)

        def sample(self, population, k):
            CAP.sampled = True
            return self._r.sample(py2compat.py2_dict_order(list(population)), k)

        def random(self):
            return self._r.random()

    mod.__dict__["random"] = _Py2RandomModule()
    return mod


def locus_to_stub_reads(pb, l):
    """PileupBatch locus -> the dicts stub_pysam serves as pileup reads."""
    s = pb.locus_slice(l)
    tab = pb.alleles[l]
    out = []
    for i in range(s.start, s.stop):
        a = tab[int(pb.allele[i])]
        indel = int(pb.indel[i])
        site, ins = a, ""
        if a.startswith("INS|"):
            _, r, ra = a.split("|")
            site, ins = r, ra[1:]
            assert indel == len(ins)
        elif a.startswith("DEL|"):
            _, rd, r = a.split("|")
            site = r
            assert indel == -(len(rd) - 1)
        qlen, lsp, nind = int(pb.qlen[i]), int(pb.left_sp[i]), int(pb.n_indel[i])
        cigar = []
        if lsp:
            cigar.append((4, lsp))
        body = int(pb.qalen[i])
        cigar.append((0, max(1, body // 2)))
        if nind:
            cigar.append((1 if indel > 0 else 2, nind))
        cigar.append((0, max(1, body - body // 2)))
        fl = int(pb.flag[i])
        out.append(dict(
            qname="f%d_%d:NN:%s:x" % (int(pb.umi[i]), int(pb.frag[i]),
                                      pb.umi_names[l][int(pb.umi[i])] if pb.umi_names is not None else "U%d" % int(pb.umi[i])),
            mq=int(pb.mq[i]), nm=int(pb.nm[i]), has_nm=bool(fl & F_HAS_NM), cigar=cigar,
            qlen=qlen, qalen=int(pb.qalen[i]), is_read1=bool(fl & F_READ1),
            is_read2=bool(fl & F_READ2), is_reverse=bool(fl & F_REVERSE),
            qpos=None if pb.is_del[i] else int(pb.qpos[i]), site=site, ins=ins,
            bq=int(pb.bq[i]), indel=indel, is_del=bool(pb.is_del[i])))
    return out


def _tie_ambiguous(final_items):
    """True when the allele picked at smCounter.py:535-542 hinges on an exact PI tie that involves a
    key other than A/T/G/C: its order in a py2 dict depends on hash collisions and insertion
    history that cannot be reproduced, so ALT-dependent columns are not pinned."""
    if len(final_items) < 2:
        return False
    srt = builtins.sorted(final_items, key=lambda kv: -kv[1])
    groups = []
    for idx in (0, 1):
        v = srt[idx][1]
        groups.append([k for k, x in final_items if x == v])
    for g in groups:
        if len(g) > 1 and any(k not in ("A", "T", "G", "C") for k in g):
            return True
    return False


_MOD = None


def run_reference(pb, params, chroms, use_wrapper=True):
    """Call the reference's vc_wrapper() per locus.  `chroms`: {name: sequence} fake FASTA.

    Returns a list of dicts: row (the TAB-joined string vc() returns), pi_raw (unrounded
    PI_A, PI_T, PI_G, PI_C, PI_alt; empty for zero-coverage rows), fisher (captured calls).
    """
    global _MOD
    if _MOD is None:
        _MOD = load_reference()
    mod = _MOD
    bam, fa = "synthetic.bam", "synthetic.fa"
    loci = {}
    for l in range(pb.n_loci):
        loci[(pb.chrom[l], int(pb.pos[l]))] = locus_to_stub_reads(pb, l)
    stub_pysam.register_bam(bam, loci)
    stub_pysam.register_fasta(fa, chroms)
    res = []
    fn = mod.vc_wrapper if use_wrapper else mod.vc
    for l in range(pb.n_loci):
        CAP.reset()
        with contextlib.redirect_stdout(io.StringIO()):
            row = fn(bam, pb.chrom[l], builtins.str(int(pb.pos[l])), params.minBQ, params.minMQ,
                     params.mtDepth, params.rpb, params.hpLen, params.mismatchThr, params.mtDrop,
                     params.maxMT, params.primerDist, fa)
        res.append(dict(row=row, pi_raw=list(CAP.round2), fisher=list(CAP.fisher),
                        n_umi_used=len(CAP.calprob), tie_ambiguous=_tie_ambiguous(CAP.final),
                        sampled=bool(CAP.sampled)))
    return res


if __name__ == "__main__":
    from smcounter_amd import synth
    import time
    cfg = synth.CONFIGS["C3"]
    pb = synth.generate(cfg, 0, 4)
    ref = {cfg.chrom: synth.CyclicRef().fetch(cfg.chrom, 0, cfg.start_pos + 100)}
    t = time.time()
    out = run_reference(pb, synth.params_for(cfg), ref)
    print("%.2f s" % (time.time() - t))
    for o in out:
        print(o["row"])
        print(o["pi_raw"], len(o["fisher"]))
