"""The walk into a dozen allocations of the read words with the (tile, part) wavefronts dealt to the XCDs (a) as contiguous eighths
of the list (the default) and (b) in groups of G consecutive wavefronts round-robin (SMC_BP_XCD_GROUP=G: ~ a tile's parts per
group) - is there a dealing whose time does not depend on the allocation?  (dev tool)  usage: r05_xcd_group.py [n_loci] [G,G,...]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["SMC_VMM_CHUNK_MB"] = "0"
from smcounter_amd import synth, engine
from smcounter_amd.engine import DevBuf
import bench_fa

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
groups = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "0,18,36,9,72").split(",")]
eng = engine.Engine(0)
cfg = synth.CONFIGS[os.environ.get("SWEEP_CFG", "C3")]
run = bench_fa.AlignmentRun(eng, cfg, synth.params_for(cfg), n, min(48, len(os.sched_getaffinity(0))), slots=1, place=0)
cands, spacers = [run.slots[0]["words"]], []
for i in range(9):
    spacers.append(DevBuf(eng, ((37, 301, 1024, 2500, 150, 4097, 611, 1777)[i % 8] << 20) + 4096))
    cands.append(DevBuf(eng, 4 * (run.ns + 64)))
print("allocation      " + "".join("%10s" % ("G=%d" % g) for g in groups))
for k, c in enumerate(cands):
    row = []
    for g in groups:
        os.environ["SMC_BP_XCD_GROUP"] = str(g)
        run.slots[0]["words"] = c
        row.append(run._walk_ms(reps=4))
    print("hipMalloc #%-5d " % k + "".join("%10.3f" % x for x in row), flush=True)
