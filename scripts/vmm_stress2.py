"""A plain hipMalloc block that follows the FREE of a virtual-memory block: does it keep its data?  (dev tool, round 5)"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from smcounter_amd import engine, _lib
n_it = int(sys.argv[1]) if len(sys.argv) > 1 else 40
mb = 576
eng = engine.Engine(0)
L = eng.L
pat = (np.arange(1 << 18, dtype=np.uint32) * 2654435761 + 12345).astype(np.uint32)
bad = 0
for it in range(n_it):
    os.environ["SMC_VMM_CHUNK_MB"] = "64"
    v = ctypes.c_void_p()
    _lib.check(L.smc_mem_alloc(eng.ctx, mb << 20, ctypes.byref(v)), "alloc vmm")
    L.smc_mem_h2d(eng.ctx, v.value, pat.ctypes.data, pat.nbytes)
    L.smc_mem_free(eng.ctx, v)
    os.environ["SMC_VMM_CHUNK_MB"] = "0"
    p = ctypes.c_void_p()
    _lib.check(L.smc_mem_alloc(eng.ctx, mb << 20, ctypes.byref(p)), "alloc plain")
    offs = [0, (mb << 20) // 2, (mb << 20) - pat.nbytes]
    for o in offs:
        L.smc_mem_h2d(eng.ctx, p.value + o, pat.ctypes.data, pat.nbytes)
    for delay in (0.0, 0.01, 0.03, 0.1):
        time.sleep(delay)
        for o in offs:
            got = np.empty_like(pat)
            L.smc_mem_d2h(eng.ctx, got.ctypes.data, p.value + o, got.nbytes)
            if not (got == pat).all():
                bad += 1
    L.smc_mem_free(eng.ctx, p)
print("vmm_stress2: %d iterations (vmm block freed, then a plain block written and read back): %d bad read-backs" % (n_it, bad))
