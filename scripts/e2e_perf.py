"""End-to-end timing of the command line on a synthetic BAM of realistic shape (dev tool, GPU box):
BAM decode -> device batch -> kernels -> row strings -> writers, stage by stage.
usage: e2e_perf.py [n_loci] [depth] [reads_per_umi]"""
import os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from smcounter_amd import bamio, bedops, fasta, rows, writers, postfilter, abi
from smcounter_amd.params import VcParams

n_loci = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
rpu = int(sys.argv[3]) if len(sys.argv) > 3 else 60
RL = 120
rng = np.random.Generator(np.random.PCG64(11))
span = n_loci + 2 * RL
L = span + 2000
ref = "".join(rng.choice(list("ACGT"), size=L))
tmp = tempfile.mkdtemp()
fa = os.path.join(tmp, "ref.fa")
with open(fa, "w") as fh:
    fh.write(">chrE\n")
    for i in range(0, L, 60):
        fh.write(ref[i:i + 60] + "\n")
n_reads = depth * span // RL
n_umi = max(1, n_reads // rpu)
t0 = time.time()
recs = []
refb = np.frombuffer(ref.encode(), np.uint8)
for u in range(n_umi):
    c = int(rng.integers(1000 - RL, 1000 + n_loci))
    umi = "".join(rng.choice(list("ACGT"), size=12))
    for f in range(rpu // 2):
        start = max(0, c + int(rng.integers(-20, 20)))
        for mate in (0, 1):
            pos = start + (0 if mate == 0 else int(rng.integers(0, 30)))
            s = refb[pos:pos + RL].copy()
            err = rng.random(RL) < 1e-3
            s[err] = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, int(err.sum()))]
            recs.append(dict(tid=0, pos=pos, qname="i:1:r%d_%d:NN:%s:x" % (u, f, umi),
                             flag=(0x40 if mate == 0 else 0x80 | 0x10) | 1, mapq=60, cigar=[(0, RL)],
                             seq=s.tobytes().decode(), qual=rng.choice([25, 30, 37, 40], size=RL).astype(np.uint8).tolist(),
                             nm=int(err.sum())))
recs.sort(key=lambda r: r["pos"])
bam = os.path.join(tmp, "e2e.bam")
bamio.write_bam(bam, [("chrE", L)], recs)
bamio.write_bai(bam)
bed = os.path.join(tmp, "t.bed")
open(bed, "w").write("chrE\t1000\t%d\n" % (1000 + n_loci))
print("fixture: %d alignments, %d UMIs, %.1f s to write" % (len(recs), n_umi, time.time() - t0))

P = VcParams(mtDepth=max(1, depth // rpu), rpb=float(rpu), hpLen=8)
ref_f = fasta.FastaFile(fa)
loci = bedops.expand_loci(bed)
from smcounter_amd.engine import Engine
import gc
gc.disable()          # as cli.main does for the duration of a run
import torch
eng = Engine(0)
for rep in range(2):
    t_dec = t_gpu = t_fmt = 0.0
    n_rd = 0
    out = []
    t = time.time()
    it = bamio.iter_device_batches_native(bam, ref_f, loci, P, max_reads=8_000_000)
    while True:
        t1 = time.time()
        try:
            first, db = next(it)
        except StopIteration:
            break
        t2 = time.time()
        R = eng.call_batch_host(db, P)
        t3 = time.time()
        out.extend(rows.format_rows(R, db, P, ref_f))
        t4 = time.time()
        t_dec += t2 - t1; t_gpu += t3 - t2; t_fmt += t4 - t3; n_rd += db.n_reads
    t5 = time.time()
    final = postfilter.apply_repeat_filters(out, {}, {})
    writers.write_outputs(os.path.join(tmp, "o"), final, writers.pi_threshold(P.mtDepth, 0))
    t6 = time.time()
    print("pass %d: %d loci, %d pileup reads: decode+planes %.2fs (%.1f M reads/s) | H2D+kernels+D2H %.2fs | "
          "format rows %.2fs (%.0f loci/s) | post-filter+writers %.2fs | total %.2fs -> %.0f loci/s" % (
              rep, len(out), n_rd, t_dec, n_rd / t_dec / 1e6, t_gpu, t_fmt, len(out) / t_fmt, t6 - t5, t6 - t,
              len(out) / (t6 - t)))

# planes built on the device (the default of the command line): host decodes alignments, k_build_planes builds the planes
from smcounter_amd import cli, devplanes, vc
os.environ["SMC_DEVPLANES_TIMING"] = "1"
os.environ["SMC_BAM_TIMING"] = "1"
for rep in range(3):
    for k in devplanes._TIMES: devplanes._TIMES[k] = 0.0
    torch.cuda.synchronize()
    t = time.time()
    t_bld = t_run = 0.0
    out2, n_dev, n_host = cli._Rows(), 0, 0
    it = devplanes.iter_resident_batches(bam, ref_f, loci, P, eng, all_planes=False)
    while True:
        t1 = time.time()
        try:
            first, rb = next(it)
        except StopIteration:
            break
        torch.cuda.synchronize()
        t2 = time.time()
        out2.add(vc.vc_resident(rb, P, ref_f, eng))
        t3 = time.time()
        t_bld += t2 - t1; t_run += t3 - t2; n_dev += rb.n_device_runs; n_host += rb.n_host_runs
    t5 = time.time()
    out2.done()
    final2 = postfilter.apply_repeat_filters(out2, {}, {}, pred=out2.pred)
    writers.write_outputs(os.path.join(tmp, "o2"), final2, writers.pi_threshold(P.mtDepth, 0), pred=out2.pred)
    t6 = time.time()
    print("device planes, pass %d: %d loci (%d device runs, %d host runs): decode + upload + k_build_planes %.3fs | kernels + D2H + "
          "row strings %.3fs | post-filter+writers %.3fs | total %.3fs -> %.0f loci/s; rows equal to the host-planes pass: %s" % (
              rep, len(out2), n_dev, n_host, t_bld, t_run, t6 - t5, t6 - t, len(out2) / (t6 - t),
              list(out2) == out and all(open(os.path.join(tmp, "o" + e)).read() == open(os.path.join(tmp, "o2" + e)).read()
                                        for e in (".smCounter.all.txt", ".smCounter.cut.txt"))))
    print("   stages:", {k: round(v, 4) for k, v in devplanes._TIMES.items()})

os.environ.pop("SMC_BAM_TIMING", None)
# the command line itself, as a child process (interpreter start, imports, context creation included)
import subprocess
eng.close()
for rep in range(2):
    t = time.time()
    subprocess.check_call([sys.executable, "-m", "smcounter_amd.cli", "--outPrefix", os.path.join(tmp, "cli"), "--bamFile", bam,
                           "--bedTarget", bed, "--mtDepth", str(P.mtDepth), "--rpb", str(P.rpb), "--refGenome", fa,
                           "--logFile", os.path.join(tmp, "cli.log")], cwd=ROOT)
    dt = time.time() - t
    print("command line (child process), run %d: %.2f s wall for %d loci -> %.0f loci/s" % (rep, dt, len(loci), len(loci) / dt))
same = open(os.path.join(tmp, "cli.smCounter.all.txt")).read() == open(os.path.join(tmp, "o.smCounter.all.txt")).read()
print("its all.txt equals the in-process one:", same)
