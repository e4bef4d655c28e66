"""cProfile of the in-process command-line path with device-built planes (dev tool, GPU box): where the Python around the
native decoder and the kernels spends its time.  usage: e2e_profile.py [n_loci] [depth] [reads_per_umi]"""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = open(os.path.join(ROOT, "scripts", "e2e_perf.py")).read()
g = {"__file__": os.path.join(ROOT, "scripts", "e2e_perf.py"), "__name__": "fixture"}
exec(compile(src[:src.index("from smcounter_amd.engine import Engine")], "e2e_head", "exec"), g)
bam, P, ref_f, loci, tmp = g["bam"], g["P"], g["ref_f"], g["loci"], g["tmp"]
from smcounter_amd import cli, devplanes, vc, postfilter, writers
from smcounter_amd.engine import Engine
import gc
gc.disable()          # as cli.main does for the duration of a run
eng = Engine(0)


def one_pass():
    out = cli._Rows()
    for first, rb in devplanes.iter_resident_batches(bam, ref_f, loci, P, eng):
        out.add(vc.vc_resident(rb, P, ref_f, eng))
    out.done()
    final = postfilter.apply_repeat_filters(out, {}, {}, pred=out.pred)
    writers.write_outputs(os.path.join(tmp, "o2"), final, writers.pi_threshold(P.mtDepth, 0), pred=out.pred)
    return len(out)


for rep in range(3):
    t = time.perf_counter(); n = one_pass(); dt = time.perf_counter() - t
    print("pass %d: %d loci, %.1f ms -> %.0f loci/s" % (rep, n, 1e3 * dt, n / dt))
pr = cProfile.Profile()
pr.enable(); one_pass(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
