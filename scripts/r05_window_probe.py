"""Is the allocation effect a matter of how much half-written output is in flight?  The walk into ten allocations of the read words
with its wavefronts per CU capped through an LDS pad (SMC_BP_LDS_PAD): all / 16 / 8 / 4 per CU (dev tool).
usage: r05_window_probe.py [n_loci]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["SMC_VMM_CHUNK_MB"] = "0"
from smcounter_amd import synth, engine
from smcounter_amd.engine import DevBuf
import bench_fa

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
pads = [0, 3072, 6144, 8192, 12288]
eng = engine.Engine(0)
cfg = synth.CONFIGS["C3"]
run = bench_fa.AlignmentRun(eng, cfg, synth.params_for(cfg), n, min(48, len(os.sched_getaffinity(0))), slots=1, place=0)
cands, spacers = [run.slots[0]["words"]], []
for i in range(9):
    spacers.append(DevBuf(eng, ((37, 301, 1024, 2500, 150, 4097, 611, 1777)[i % 8] << 20) + 4096))
    cands.append(DevBuf(eng, 4 * (run.ns + 64)))
print("allocation      " + "".join("%12s" % ("pad %d" % p) for p in pads))
for k, c in enumerate(cands):
    row = []
    for p in pads:
        os.environ["SMC_BP_LDS_PAD"] = str(p)
        run.slots[0]["words"] = c
        row.append(run._walk_ms(reps=4))
    print("hipMalloc #%-5d " % k + "".join("%12.3f" % x for x in row), flush=True)
os.environ["SMC_BP_LDS_PAD"] = "0"
# the same with parts of other sizes (rows of the sorted list per wavefront)
parts = [256]
print("allocation      " + "".join("%12s" % ("part %d" % p) for p in parts))
for k, c in enumerate(cands):
    row = []
    for p in parts:
        os.environ["SMC_BP_PART"] = str(p)
        run.slots[0]["words"] = c
        row.append(run._walk_ms(reps=4))
    print("hipMalloc #%-5d " % k + "".join("%12.3f" % x for x in row), flush=True)
