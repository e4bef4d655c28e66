"""Generate tests/golden/tools_ds.npz (BUILD CONTAINER ONLY): what the reference's offline down-samplers themselves write.

/root/reference/ds.mt.py and ds.reads.withinMT.py are translated in memory by lib2to3 (mechanical py2 -> py3 fixers) and run
with a stub `pysam` (AlignmentFile read + write: only `fetch()`, `query_name`, `write()`, `close()` are touched, ds.mt.py:31-66,
ds.reads.withinMT.py:30-84) and with `defaultdict` bound to a dict that iterates its keys in CPython-2.7 order (the scripts walk
`bcDict.keys()` / `.values()` while drawing random numbers: ds.mt.py:51, ds.reads.withinMT.py:62).  `random` is the interpreter's
own module: for an integer seed CPython 2 and 3 initialise the Mersenne Twister the same way and `random()` is the same function -
so this also checks py2compat.Py2Random against the real generator.
The fixture holds the input read names and, per (script, seed, parameter), the read names written, in order.
Usage: python tests/golden/make_tools_golden.py
"""
import collections
import contextlib
import io
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from smcounter_amd import py2compat  # noqa: E402

REF_DIR = os.environ.get("SMC_REFERENCE_DIR", "/root/reference")


class _Py2OrderDefaultDict(collections.defaultdict):
    """defaultdict whose keys() / values() / items() come in CPython-2.7 hash-slot order of the insertion history."""

    def _order(self):
        return py2compat.py2_dict_order(list(collections.defaultdict.keys(self)))

    def keys(self):
        return self._order()

    def values(self):
        return [self[k] for k in self._order()]

    def items(self):
        return [(k, self[k]) for k in self._order()]

    def __iter__(self):
        return iter(self._order())


class _Read(object):
    def __init__(self, qname):
        self.query_name = qname


class _StubPysam(object):
    def __init__(self, qnames):
        self.qnames, self.written = qnames, []
        outer = self

        class AlignmentFile(object):
            def __init__(self, path, mode="rb", template=None, text=None):
                self.mode = mode

            def fetch(self):
                return [_Read(q) for q in outer.qnames]

            def write(self, read):
                outer.written.append(read.query_name)

            def close(self):
                pass
        self.AlignmentFile = AlignmentFile


def run_script(fname, qnames, **args):
    from lib2to3 import refactor
    src = open(os.path.join(REF_DIR, fname)).read()
    tool = refactor.RefactoringTool(refactor.get_fixers_from_package("lib2to3.fixes"))
    with contextlib.redirect_stderr(io.StringIO()):
        py3 = str(tool.refactor_string(src + "\n", fname))
    stub = _StubPysam(qnames)
    sys.modules["pysam"] = stub
    mod = types.ModuleType("ref_" + fname.replace(".", "_"))
    exec(compile(py3, fname, "exec"), mod.__dict__)
    mod.__dict__["defaultdict"] = _Py2OrderDefaultDict
    ns = types.SimpleNamespace(runPath=".", inBam="in.bam", outBam="out.bam", **args)
    with contextlib.redirect_stdout(io.StringIO()):
        mod.main(ns)
    return list(stub.written)


def make_qnames(rng, n_bc, max_frag):
    out = []
    for b in range(n_bc):
        bc = "".join(rng.choice(list("ACGT"), 12))
        for f in range(int(rng.randint(1, max_frag + 1))):
            name = "M03:%d:000-X:1:%d:%d:%s:%d" % (rng.randint(1, 99), 1100 + f, rng.randint(1000, 29999), bc, rng.randint(0, 9))
            out.append(name)
            if rng.rand() < 0.7:
                out.append(name)                    # the mate: same read name
    perm = rng.permutation(len(out))                # "coordinate order": barcodes interleaved
    return [out[i] for i in perm]


if __name__ == "__main__":
    rng = np.random.RandomState(20176)
    cases = []
    for n_bc, max_frag in ((5, 3), (40, 6), (300, 9)):          # 5: below the first py2 dict resize; 300: several resizes
        q = make_qnames(rng, n_bc, max_frag)
        runs = []
        for seed, pct in ((1234567, 0.5), (7, 0.2), (99, 0.9)):
            runs.append(dict(script="ds.mt.py", seed=seed, pct=pct, written=run_script("ds.mt.py", q, pct=pct, seed=seed)))
        for seed, rpb in ((1234567, 1.5), (7, 2.5), (99, 1.0)):
            runs.append(dict(script="ds.reads.withinMT.py", seed=seed, rpb=rpb, written=run_script("ds.reads.withinMT.py", q, rpb=rpb, seed=seed)))
        cases.append(dict(qnames=q, runs=runs))
        print("%d barcodes, %d records: " % (n_bc, len(q)) + ", ".join("%s %d" % (r["script"][:8], len(r["written"])) for r in runs))
    np.savez_compressed(os.path.join(HERE, "tools_ds.npz"), meta=np.frombuffer(json.dumps(cases).encode(), np.uint8))
