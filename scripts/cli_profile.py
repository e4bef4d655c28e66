"""cProfile of cli.main in process on the e2e fixture (dev tool, GPU box).  usage: cli_profile.py [n_loci] [depth] [rpu]"""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = open(os.path.join(ROOT, "scripts", "e2e_perf.py")).read()
g = {"__file__": os.path.join(ROOT, "scripts", "e2e_perf.py"), "__name__": "fixture"}
exec(compile(src[:src.index("from smcounter_amd.engine import Engine")], "e2e_head", "exec"), g)
bam, bed, fa, P, tmp = g["bam"], g["bed"], g["fa"], g["P"], g["tmp"]
from smcounter_amd import cli
args = dict(outPrefix=os.path.join(tmp, "p"), bamFile=bam, bedTarget=bed, mtDepth=P.mtDepth, rpb=P.rpb, refGenome=fa,
            logFile=os.path.join(tmp, "p.log"))
import io, contextlib
for rep in range(3):
    t = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        cli.main(dict(args))
    print("cli.main pass %d: %.1f ms" % (rep, 1e3 * (time.perf_counter() - t)), flush=True)
pr = cProfile.Profile()
pr.enable()
with contextlib.redirect_stdout(io.StringIO()):
    cli.main(dict(args))
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(30)
