"""Why does the walk's time depend on the allocation of the read words?  (1) the walk into a dozen allocations; (2) the fastest,
a middle one and the slowest again with the tiles dealt to the XCDs in each of the eight rotations (SMC_BP_XCD_ROT); (3) for the
fastest and the slowest: stream-write (and read) time of every 64 MB region from every single XCD (dev tool).
usage: r05_xcd_probe.py [n_loci]"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from smcounter_amd import _lib, abi, synth, engine
import bench_fa

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
V = ctypes.CDLL(os.path.join(ROOT, "scripts", "libvmm_probe.so"))
vp = ctypes.c_void_p
V.vmm_alloc.argtypes = [ctypes.c_int, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t, ctypes.POINTER(vp)]
V.plain_alloc.argtypes = [ctypes.c_size_t, ctypes.POINTER(vp)]
V.region_matrix.argtypes = [vp, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int, ctypes.POINTER(ctypes.c_float)]


class Ptr(object):
    def __init__(self, p, label):
        self.p, self.label = int(p), label

    def data_ptr(self):
        return self.p

    def free(self):
        pass


os.environ["SMC_VMM_CHUNK_MB"] = "0"          # (the run's own arrays: plain hipMalloc, as before this round)
eng = engine.Engine(0)
L = eng.L
cfg = synth.CONFIGS["C3"]
params = synth.params_for(cfg)
run = bench_fa.AlignmentRun(eng, cfg, params, n, min(48, len(os.sched_getaffinity(0))), slots=1, place=0)
need = 4 * (run.ns + 64)
cands = [Ptr(run.slots[0]["words"].data_ptr(), "hipMalloc (the run's own)")]
for rep in range(4):
    for mb in (64, 256):
        p = vp()
        if V.vmm_alloc(0, need, mb << 20, 2 << 20, ctypes.byref(p)) == 0:
            cands.append(Ptr(p.value, "VMM %4d MB handles #%d" % (mb, rep)))
    p = vp()
    assert V.plain_alloc(need, ctypes.byref(p)) == 0
    cands.append(Ptr(p.value, "hipMalloc #%d" % (rep + 1)))


def walk(c, rot=0):
    os.environ["SMC_BP_XCD_ROT"] = str(rot)
    run.slots[0]["words"] = c
    return run._walk_ms(reps=4)


ms = [walk(c) for c in cands]
for c, m in zip(cands, ms):
    print("%-28s 0x%012x  k_bp_emit2 %.3f ms" % (c.label, c.p, m), flush=True)
order = sorted(range(len(cands)), key=lambda i: ms[i])
picks = [order[0], order[len(order) // 2], order[-1]]
print("\nthe tiles dealt to the XCDs in the eight rotations (XCD x takes eighth (x + rot) & 7 of the tiles):")
for i in picks:
    print("%-28s %s" % (cands[i].label, "  ".join("%.3f" % walk(cands[i], r) for r in range(8))), flush=True)
os.environ["SMC_BP_XCD_ROT"] = "0"
REG = 64 << 20
nr = need // REG
for wr in (1, 0):
    for i in (picks[0], picks[-1]):
        out = (ctypes.c_float * (nr * 9))()
        rc = V.region_matrix(vp(cands[i].p), need, REG, wr, out)
        m = np.frombuffer(out, np.float32).reshape(nr, 9)
        print("\n%s of every 64 MB region of [%s] (walk %.3f ms) from ONE XCD at a time (columns: XCD 0-7, all XCDs), us; rc %d" % (
            "stream-WRITE" if wr else "stream-READ", cands[i].label, ms[i], rc))
        for r in range(nr):
            print("  region %2d: %s" % (r, " ".join("%6.0f" % (1e3 * x) for x in m[r])))
        print("  mean     : %s" % " ".join("%6.0f" % (1e3 * x) for x in m.mean(axis=0)))
