"""Host half of the device plane builder alone (dev tool, no GPU): smc_bam_alignments over the e2e fixture, run by run, with
the decoder's own stage times (SMC_BAM_TIMING).  usage: decode_alignments_perf.py [n_loci] [depth] [reads_per_umi] [nthreads]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("SMC_BAM_TIMING", "1")
src = open(os.path.join(ROOT, "scripts", "e2e_perf.py")).read()
g = {"__file__": os.path.join(ROOT, "scripts", "e2e_perf.py"), "__name__": "fixture"}
argv = sys.argv
sys.argv = argv[:4]
exec(compile(src[:src.index("from smcounter_amd.engine import Engine")], "e2e_head", "exec"), g)
bamio, P, loci, bam = g["bamio"], g["P"], g["loci"], g["bam"]
nthreads = int(argv[4]) if len(argv) > 4 else len(os.sched_getaffinity(0))
B = bamio.NativeBam(bam)
lo, hi = int(loci[0][1]) - 1, int(loci[-1][1])
for rep in range(4):
    t = time.perf_counter()
    i, tot = lo, 0
    while i < hi:
        j = min(hi, i + (8192 if i == lo else 16384))
        A = B.alignments_run(loci[0][0], i, j, 64_000_000, P, nthreads)
        tot += A["reads"]; i += A["nl"]
    print("pass %d: %d pileup reads, %d threads: %.1f ms" % (rep, tot, nthreads, 1e3 * (time.perf_counter() - t)), flush=True)
