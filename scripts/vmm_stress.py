"""Does a freshly mapped virtual-memory block keep what is written to it right away?  (round 5: on some boxes the first batch's
umi_start - a fresh 576 MB block of smc_mem_alloc - read back as zeros some tens of milliseconds after the builder had written it;
dev tool)  usage: vmm_stress.py [iterations] [MB]"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from smcounter_amd import engine, _lib
n_it = int(sys.argv[1]) if len(sys.argv) > 1 else 40
mb = int(sys.argv[2]) if len(sys.argv) > 2 else 576
os.environ.setdefault("SMC_VMM_CHUNK_MB", "64")       # (smc_mem_alloc backs nothing by virtual memory any more unless told to; 0: plain hipMalloc)
eng = engine.Engine(0)
L = eng.L
pat = (np.arange(1 << 18, dtype=np.uint32) * 2654435761 + 12345).astype(np.uint32)        # 1 MB
bad = 0
for it in range(n_it):
    p = ctypes.c_void_p()
    _lib.check(L.smc_mem_alloc(eng.ctx, mb << 20, ctypes.byref(p)), "alloc")
    offs = [0, (mb << 20) // 2, (mb << 20) - pat.nbytes]
    for o in offs:
        _lib.check(L.smc_mem_h2d(eng.ctx, p.value + o, pat.ctypes.data, pat.nbytes), "h2d")
    other = ctypes.c_void_p()
    _lib.check(L.smc_mem_alloc(eng.ctx, 300 << 20, ctypes.byref(other)), "alloc2")
    ms = ctypes.c_float()
    for delay in (0.0, 0.01, 0.03, 0.1):
        time.sleep(delay)
        L.smc_mem_write_probe(eng.ctx, other, 300 << 20, ctypes.byref(ms))       # (kernels run in between, as in the product)
        for o in offs:
            got = np.empty_like(pat)
            _lib.check(L.smc_mem_d2h(eng.ctx, got.ctypes.data, p.value + o, got.nbytes), "d2h")
            if not (got == pat).all():
                bad += 1
                print("iteration %d, offset %d, after %.0f ms: %d of %d words differ, %d of them zero" % (
                    it, o, 1e3 * delay, int((got != pat).sum()), len(pat), int((got[got != pat] == 0).sum())))
    L.smc_mem_free(eng.ctx, other)
    L.smc_mem_free(eng.ctx, p)
print("vmm_stress: %d iterations of %d MB, SMC_VMM_CHUNK_MB=%s: %d bad read-backs" % (n_it, mb, os.environ.get("SMC_VMM_CHUNK_MB", "(default)"), bad))
