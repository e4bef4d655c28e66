"""The command line under torch.distributed.run with two ranks sharing the box's one GPU (SMC_SHARE_GPU=1: gloo between them)
against a single-process run: the same three files (dev tool, GPU box; a functional check of the multi-rank path).
usage: cli_two_ranks.py [n_loci] [depth] [reads_per_umi]"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = open(os.path.join(ROOT, "scripts", "e2e_perf.py")).read()
g = {"__file__": os.path.join(ROOT, "scripts", "e2e_perf.py"), "__name__": "fixture"}
exec(compile(src[:src.index("from smcounter_amd.engine import Engine")], "e2e_head", "exec"), g)
bam, bed, fa, P, tmp, loci = g["bam"], g["bed"], g["fa"], g["P"], g["tmp"], g["loci"]
common = ["--bamFile", bam, "--bedTarget", bed, "--mtDepth", str(P.mtDepth), "--rpb", str(P.rpb), "--refGenome", fa]
t = time.time()
subprocess.check_call([sys.executable, "-m", "smcounter_amd.cli", "--outPrefix", os.path.join(tmp, "one")] + common, cwd=ROOT,
                      stdout=subprocess.DEVNULL)
t1 = time.time() - t
env = dict(os.environ, SMC_SHARE_GPU="1")
t = time.time()
subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                       "--master-port", "29533", "-m", "smcounter_amd.cli", "--outPrefix", os.path.join(tmp, "two")] + common,
                      cwd=ROOT, env=env, stdout=subprocess.DEVNULL)
t2 = time.time() - t
same = all(open(os.path.join(tmp, "one" + e)).read() == open(os.path.join(tmp, "two" + e)).read()
           for e in (".smCounter.all.txt", ".smCounter.cut.txt"))
v1 = [l for l in open(os.path.join(tmp, "one.smCounter.cut.vcf")) if not l.startswith("#")]
v2 = [l for l in open(os.path.join(tmp, "two.smCounter.cut.vcf")) if not l.startswith("#")]
print("%d loci: one process %.2f s, two ranks on the one GPU %.2f s (torch + rendezvous start-up included); same all.txt / cut.txt / "
      "cut.vcf records: %s" % (len(loci), t1, t2, same and v1 == v2))
