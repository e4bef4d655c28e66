"""What distinguishes a fast allocation of the read words from a slow one?  One C3-shaped run; the walk
(k_bp_emit2) timed into the SAME 2.4 GB of read words placed in: an arena allocated FIRST in the process (carved at several
offsets), plain hipMalloc allocations made later (behind spacers), and HIP virtual-memory allocations (one physical handle, or
handles of 2 MB ... 1 GB, virtual range aligned to 2 MB / 1 GB).  With R05_PMC=1 the script ends with one walk into the fastest
and one into the slowest candidate (in that order) so that a `rocprofv3 --pmc` pass shows their counters side by side.
usage: r05_place_probe.py [n_loci] [n_plain]"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
n_plain = int(sys.argv[2]) if len(sys.argv) > 2 else 6
V = ctypes.CDLL(os.path.join(ROOT, "scripts", "libvmm_probe.so"))
vp = ctypes.c_void_p
V.vmm_alloc.argtypes = [ctypes.c_int, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t, ctypes.POINTER(vp)]
V.plain_alloc.argtypes = [ctypes.c_size_t, ctypes.POINTER(vp)]
V.plain_free.argtypes = [vp]


class Ptr(object):
    def __init__(self, p, label):
        self.p, self.label = int(p), label

    def data_ptr(self):
        return self.p

    def free(self):
        pass


# ---- the arena: the process's FIRST device allocation (before the library's tables, before the inputs)
ARENA = 24 << 30
a = vp()
t0 = time.perf_counter()
assert V.plain_alloc(ARENA, ctypes.byref(a)) == 0
print("arena of %d GB at 0x%x (%.1f ms)" % (ARENA >> 30, a.value, (time.perf_counter() - t0) * 1e3), flush=True)

from smcounter_amd import _lib, abi, synth, engine
from smcounter_amd.engine import DevBuf
import bench_fa

eng = engine.Engine(0)
L = eng.L
cfg = synth.CONFIGS["C3"]
run = bench_fa.AlignmentRun(eng, cfg, synth.params_for(cfg), n, min(48, len(os.sched_getaffinity(0))), slots=1, place=0)
need = 4 * (run.ns + 64)
gmin, grec = ctypes.c_size_t(), ctypes.c_size_t()
V.vmm_granularity(0, ctypes.byref(gmin), ctypes.byref(grec))
print("read words: %.2f GB; VMM granularity min %d, recommended %d" % (need / 1e9, gmin.value, grec.value), flush=True)

cands = [Ptr(run.slots[0]["words"].data_ptr(), "hipMalloc #0 (as the run allocated it, after the inputs)")]
step = (need + (1 << 21) - 1) >> 21 << 21
for k, off in enumerate((0, step, 3 * step + (64 << 20), 7 * step + (1 << 20))):
    if off + need <= ARENA:
        cands.append(Ptr(a.value + off, "arena +%.1f GB" % (off / 2**30)))
spacers = []
for i in range(n_plain):
    mb = (37, 301, 1024, 2500, 150, 4097, 611, 1777)[i % 8]
    s, p = vp(), vp()
    assert V.plain_alloc((mb << 20) + 4096, ctypes.byref(s)) == 0 and V.plain_alloc(need, ctypes.byref(p)) == 0
    spacers.append(s)
    cands.append(Ptr(p.value, "hipMalloc #%d (behind a %d MB spacer)" % (i + 1, mb)))
for chunk, align, label in ((0, 2 << 20, "VMM one handle, VA 2 MB"), (0, 1 << 30, "VMM one handle, VA 1 GB"), (1 << 30, 1 << 30, "VMM 1 GB handles, VA 1 GB"),
                            (64 << 20, 2 << 20, "VMM 64 MB handles"), (2 << 20, 2 << 20, "VMM 2 MB handles")):
    p = vp()
    rc = V.vmm_alloc(0, need, chunk, align, ctypes.byref(p))
    if rc == 0:
        cands.append(Ptr(p.value, label))
    else:
        print("  (%s: failed at step %d)" % (label, rc), flush=True)

res = []
for rnd in range(2):
    for c in cands:
        run.slots[0]["words"] = c
        ms = run._walk_ms(reps=4)
        if rnd == 0:
            res.append([ms])
        else:
            res[cands.index(c)].append(ms)
for c, r in zip(cands, res):
    print("%-62s 0x%012x  k_bp_emit2 %s ms" % (c.label, c.p, " ".join("%.3f" % x for x in r)), flush=True)
# the same words whatever the placement
run.slots[0]["words"] = cands[0]
order = sorted(range(len(cands)), key=lambda i: min(res[i]))
if os.environ.get("R05_PMC"):
    for i in (order[0], order[-1]):
        run.slots[0]["words"] = cands[i]
        run.step(slot=0)
        L.smc_device_sync(eng.ctx)
    print("FINAL two walks: fastest = %s (%.3f), slowest = %s (%.3f)" % (cands[order[0]].label, min(res[order[0]]), cands[order[-1]].label,
                                                                         min(res[order[-1]])), flush=True)
# the step as a whole on the best arena placement vs the default allocation
for i in (0, 1):
    run.slots[0]["words"] = cands[i]
    for _ in range(3):
        run.step(slot=0)
    L.smc_device_sync(eng.ctx)
    t0 = time.perf_counter()
    for _ in range(20):
        run.step(slot=0)
    L.smc_device_sync(eng.ctx)
    print("step, one at a time, words in [%s]: %.3f ms" % (cands[i].label, (time.perf_counter() - t0) / 20 * 1e3), flush=True)
