"""Where the command line's decode stage spends its time (dev tool): the native call, the Python around it, the
iterator's own bookkeeping.  usage: decode_breakdown.py [n_loci] [depth] [reads_per_umi]  (fixture of e2e_perf.py)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = open(os.path.join(ROOT, "scripts", "e2e_perf.py")).read()
g = {"__file__": os.path.join(ROOT, "scripts", "e2e_perf.py"), "__name__": "fixture"}
exec(compile(src[:src.index("from smcounter_amd.engine import Engine")], "e2e_head", "exec"), g)
bamio, P, ref_f, loci, bam = g["bamio"], g["P"], g["ref_f"], g["loci"], g["bam"]
T = {"native": 0.0, "planes_run": 0.0, "tables": 0.0}
orig_run, orig_tables = bamio.NativeBam.planes_run, bamio.NativeBam._tables


def timed_run(self, *a, **k):
    fn = self._lib.smc_bam_planes

    class Proxy(object):
        def __getattr__(s, name):
            f = getattr(self.__dict__["_real_lib"], name)
            if name != "smc_bam_planes":
                return f

            def call(*aa):
                t = time.perf_counter(); r = f(*aa); T["native"] += time.perf_counter() - t
                return r
            return call
    self.__dict__.setdefault("_real_lib", self._lib)
    self._lib = Proxy()
    t = time.perf_counter()
    try:
        return orig_run(self, *a, **k)
    finally:
        T["planes_run"] += time.perf_counter() - t
        self._lib = self.__dict__["_real_lib"]


def timed_tables(self, *a, **k):
    t = time.perf_counter()
    try:
        return orig_tables(self, *a, **k)
    finally:
        T["tables"] += time.perf_counter() - t


bamio.NativeBam.planes_run, bamio.NativeBam._tables = timed_run, timed_tables
for rep in range(3):
    for k in T:
        T[k] = 0.0
    t = time.perf_counter(); n = nb = 0
    for first, db in bamio.iter_device_batches_native(bam, ref_f, loci, P, max_reads=8_000_000):
        n += db.n_loci; nb += 1
    tot = time.perf_counter() - t
    print("pass %d: %d loci in %d batches, %.3f s: native call %.3f | rest of planes_run %.3f (allele tables %.3f) | iterator %.3f"
          % (rep, n, nb, tot, T["native"], T["planes_run"] - T["native"], T["tables"], tot - T["planes_run"]))
