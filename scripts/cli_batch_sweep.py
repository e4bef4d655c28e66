"""cli.main in process on the e2e fixture over --batchReads (dev tool, GPU box): smaller batches let the prefetch thread decode batch
i + 1 while batch i is called and printed.  usage: cli_batch_sweep.py [n_loci] [depth] [rpu]"""
import os, sys, time, io, contextlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = open(os.path.join(ROOT, "scripts", "e2e_perf.py")).read()
g = {"__file__": os.path.join(ROOT, "scripts", "e2e_perf.py"), "__name__": "fixture"}
exec(compile(src[:src.index("from smcounter_amd.engine import Engine")], "e2e_head", "exec"), g)
bam, bed, fa, P, tmp = g["bam"], g["bed"], g["fa"], g["P"], g["tmp"]
from smcounter_amd import cli
n_loci = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
for br in (4_000_000, 1_000_000, 500_000, 250_000):
    args = dict(outPrefix=os.path.join(tmp, "p"), bamFile=bam, bedTarget=bed, mtDepth=P.mtDepth, rpb=P.rpb, refGenome=fa,
                logFile=os.path.join(tmp, "p.log"), batchReads=br)
    ts = []
    for rep in range(5):
        t = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            cli.main(dict(args))
        ts.append(time.perf_counter() - t)
    print("batchReads %8d (device batches of %d reads): cli.main %s ms -> best %.0f loci/s" % (
        br, 8 * br, " ".join("%.1f" % (1e3 * x) for x in ts), n_loci / min(ts)), flush=True)
