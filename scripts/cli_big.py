"""The command line as a child process on a larger synthetic BAM (several device batches), with and without the prefetch
thread (dev tool, GPU box).  usage: cli_big.py [n_loci] [depth] [reads_per_umi]"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = open(os.path.join(ROOT, "scripts", "e2e_perf.py")).read()
g = {"__file__": os.path.join(ROOT, "scripts", "e2e_perf.py"), "__name__": "fixture"}
exec(compile(src[:src.index("from smcounter_amd.engine import Engine")], "e2e_head", "exec"), g)
bam, bed, fa, P, tmp, loci = g["bam"], g["bed"], g["fa"], g["P"], g["tmp"], g["loci"]
outs = {}
for rep in range(4):
    for mode in ("prefetch", "no prefetch"):
        env = dict(os.environ)
        if mode == "no prefetch":
            env["SMC_NO_PREFETCH"] = "1"
        t = time.time()
        subprocess.check_call([sys.executable, "-m", "smcounter_amd.cli", "--outPrefix", os.path.join(tmp, mode.replace(" ", "_")),
                               "--bamFile", bam, "--bedTarget", bed, "--mtDepth", str(P.mtDepth), "--rpb", str(P.rpb), "--refGenome", fa,
                               "--logFile", os.path.join(tmp, "cli.log")], cwd=ROOT, env=env)
        dt = time.time() - t
        outs[mode] = open(os.path.join(tmp, mode.replace(" ", "_") + ".smCounter.all.txt")).read()
        print("%s, run %d: %.2f s wall for %d loci -> %.0f loci/s" % (mode, rep, dt, len(loci), len(loci) / dt), flush=True)
print("same all.txt:", outs["prefetch"] == outs["no prefetch"])
