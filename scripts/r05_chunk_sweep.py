"""The walk (k_bp_emit2) and the locus kernel (k_call_v2) with the read words in HIP virtual-memory allocations backed by physical
handles of 4 MB ... 1 GB, three allocations per size, beside plain hipMalloc allocations - which backing is robustly fast?
(dev tool.)  usage: r05_chunk_sweep.py [n_loci]"""
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


class Ptr(object):
    def __init__(self, p, label):
        self.p, self.label = int(p), label

    def data_ptr(self):
        return self.p

    def free(self):
        pass


eng = engine.Engine(0)
L = eng.L
cfg = synth.CONFIGS[os.environ.get("SWEEP_CFG", "C3")]
params = synth.params_for(cfg)
run = bench_fa.AlignmentRun(eng, cfg, params, n, min(48, len(os.sched_getaffinity(0))), slots=1, place=0)
need = 4 * (run.ns + 64)
cands = [Ptr(run.slots[0]["words"].data_ptr(), "hipMalloc (the run's own)")]
sizes = [4, 16, 32, 64, 128, 256, 512, 1024]
for rep in range(3):
    for mb in sizes:
        p = vp()
        if V.vmm_alloc(0, need, mb << 20, 2 << 20, ctypes.byref(p)) == 0:
            cands.append(Ptr(p.value, "VMM %4d MB handles #%d" % (mb, rep)))
    p = vp()
    assert V.plain_alloc(need, ctypes.byref(p)) == 0
    cands.append(Ptr(p.value, "hipMalloc #%d" % (rep + 1)))


def call_ms(words):
    run.slots[0]["words"] = words
    plan = run.step(keep_plan=True, slot=0)
    L.smc_device_sync(eng.ctx)
    plan.set_timing(6)
    for _ in range(6):
        plan.run([words, run.slots[0]["uaux"][0]], params, run.slots[0]["rows"], stream=0)
    ms = plan.kernel_ms()[0]
    plan.close()
    return ms


def step_ms(words, reps=15):
    run.slots[0]["words"] = words
    for _ in range(3):
        run.step(slot=0)
    L.smc_device_sync(eng.ctx)
    t0 = time.perf_counter()
    for _ in range(reps):
        run.step(slot=0)
    L.smc_device_sync(eng.ctx)
    return (time.perf_counter() - t0) / reps * 1e3


print("%-28s %10s %10s %10s" % ("read words in", "k_bp_emit2", "k_call_v2", "step"))
for c in cands:
    run.slots[0]["words"] = c
    w = run._walk_ms(reps=4)
    print("%-28s %10.3f %10.3f %10.3f" % (c.label, w, call_ms(c), step_ms(c)), flush=True)
# the other big arrays of a step in chunked backing as well: umi_start (+ the two aux arrays), the input pool
best = min((c for c in cands if "VMM   64" in c.label), key=lambda c: 0, default=None)
if best is not None:
    S = run.slots[0]
    base = step_ms(best)
    ua = []
    for k in range(3):
        p = vp()
        assert V.vmm_alloc(0, 4 * (run.ns + run.nl + 64), 64 << 20, 2 << 20, ctypes.byref(p)) == 0
        ua.append(Ptr(p.value, "uaux"))
    old = S["uaux"]
    S["uaux"] = ua
    print("words in [%s]: step %.3f ms; umi_start / u_gid / u_finc chunked as well: %.3f ms" % (best.label, base, step_ms(best)), flush=True)
    S["uaux"] = old
