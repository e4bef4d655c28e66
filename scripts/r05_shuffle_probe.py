"""Is a SCRAMBLED physical layout what makes an allocation of the read words fast?  Virtual ranges whose physical handles (2 ... 64 MB)
are created in order and mapped in a shuffled order, beside the same handle sizes mapped in order and plain hipMalloc (dev tool)."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["SMC_VMM_CHUNK_MB"] = "0"
import numpy as np
from smcounter_amd import synth, engine
import bench_fa
V = ctypes.CDLL(os.path.join(ROOT, "scripts", "libvmm_probe.so"))
vp = ctypes.c_void_p
V.vmm_alloc.argtypes = [ctypes.c_int, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t, ctypes.POINTER(vp)]
V.vmm_alloc_shuffled.argtypes = [ctypes.c_int, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_uint64, ctypes.POINTER(vp)]
V.plain_alloc.argtypes = [ctypes.c_size_t, ctypes.POINTER(vp)]


class Ptr(object):
    def __init__(self, p, label):
        self.p, self.label = int(p), label

    def data_ptr(self):
        return self.p

    def free(self):
        pass


eng = engine.Engine(0)
eng.alloc_tries = 1
L = eng.L
L.smc_mem_write_probe.argtypes = [vp, vp, ctypes.c_int64, ctypes.POINTER(ctypes.c_float)]
cfg = synth.CONFIGS[os.environ.get("SWEEP_CFG", "C3")]
run = bench_fa.AlignmentRun(eng, cfg, synth.params_for(cfg), cfg.n_loci, 48, slots=1, place=0)
need = 4 * (run.ns + 64)
cands = []
for rep in range(3):
    for mb in (2, 8, 32, 64):
        p = vp()
        t0 = time.perf_counter()
        rc = V.vmm_alloc_shuffled(0, need, mb << 20, 1000 * rep + mb, ctypes.byref(p))
        dt = time.perf_counter() - t0
        if rc == 0:
            cands.append(Ptr(p.value, "shuffled %2d MB handles #%d (%.0f ms)" % (mb, rep, dt * 1e3)))
        else:
            print("shuffled %d MB: failed at step %d" % (mb, rc))
    for mb in (2, 64):
        p = vp()
        if V.vmm_alloc(0, need, mb << 20, 2 << 20, ctypes.byref(p)) == 0:
            cands.append(Ptr(p.value, "in order %2d MB handles #%d" % (mb, rep)))
    p = vp()
    assert V.plain_alloc(need, ctypes.byref(p)) == 0
    cands.append(Ptr(p.value, "hipMalloc #%d" % rep))
os.environ["SMC_PROBE_SPIN"] = "0"
for c in cands:
    run.slots[0]["words"] = c
    w = run._walk_ms(reps=3)
    ms = ctypes.c_float()
    L.smc_mem_write_probe(eng.ctx, vp(c.p), need, ctypes.byref(ms))
    print("%-40s walk %.3f ms   probe %.3f ms" % (c.label, w, ms.value), flush=True)
