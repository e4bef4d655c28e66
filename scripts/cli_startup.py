"""Where a command-line run of the small fixture spends its wall time outside main(): interpreter + imports before, teardown
after (dev tool, GPU box).  usage: cli_startup.py [n_loci] [depth] [rpu]"""
import datetime, os, re, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = open(os.path.join(ROOT, "scripts", "e2e_perf.py")).read()
g = {"__file__": os.path.join(ROOT, "scripts", "e2e_perf.py"), "__name__": "fixture"}
exec(compile(src[:src.index("from smcounter_amd.engine import Engine")], "e2e_head", "exec"), g)
bam, bed, fa, P, tmp, loci = g["bam"], g["bed"], g["fa"], g["P"], g["tmp"], g["loci"]
for rep in range(3):
    log = os.path.join(tmp, "s%d.log" % rep)
    t0 = datetime.datetime.now(); t = time.time()
    subprocess.check_call([sys.executable, "-m", "smcounter_amd.cli", "--outPrefix", os.path.join(tmp, "s"), "--bamFile", bam,
                           "--bedTarget", bed, "--mtDepth", str(P.mtDepth), "--rpb", str(P.rpb), "--refGenome", fa, "--logFile", log], cwd=ROOT)
    wall = time.time() - t
    import glob
    txt = open(sorted(glob.glob(log + ".run-log*"))[-1]).read()
    st = datetime.datetime.strptime(re.search(r"smCounter started at (.*)", txt).group(1).strip(), "%Y-%m-%d %H:%M:%S.%f")
    en = datetime.datetime.strptime(re.search(r"smCounter completed running at (.*)", txt).group(1).strip(), "%Y-%m-%d %H:%M:%S.%f")
    print("run %d: wall %.3f s = before main %.3f + main %.3f + after main %.3f" % (
        rep, wall, (st - t0).total_seconds(), (en - st).total_seconds(), wall - (en - t0).total_seconds()), flush=True)
t = time.time(); subprocess.check_call([sys.executable, "-c", "pass"]); print("python -c pass: %.3f s" % (time.time() - t))
t = time.time(); subprocess.check_call([sys.executable, "-c", "import numpy"]); print("import numpy: %.3f s" % (time.time() - t))
t = time.time(); subprocess.check_call([sys.executable, "-c", "import smcounter_amd.cli"], cwd=ROOT); print("import smcounter_amd.cli: %.3f s" % (time.time() - t))
if os.environ.get("SMC_CLI_STARTUP_PROFILE"):
    out = subprocess.run([sys.executable, "-m", "cProfile", "-s", "cumulative", "-m", "smcounter_amd.cli", "--outPrefix", os.path.join(tmp, "s"),
                          "--bamFile", bam, "--bedTarget", bed, "--mtDepth", str(P.mtDepth), "--rpb", str(P.rpb), "--refGenome", fa],
                         cwd=ROOT, capture_output=True, text=True).stdout
    i = out.index("Ordered by")
    print("\n".join(l[:150] for l in out[i:].splitlines()[:60]))
