"""Counters of the library's write-only probe (k_mem_write_probe) in a fast and in a slow allocation: sixteen candidate blocks
(engine.DevBuf: virtual-memory backing), each probed; then one more probe into the fastest and one into the slowest (the LAST two
dispatches of the kernel: scripts/pmc_last_dispatches.py).  (dev tool)"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from smcounter_amd import engine
from smcounter_amd.engine import DevBuf
eng = engine.Engine(0)
eng.alloc_tries = 1
L = eng.L
need = 4 * 600_000_000
bufs, ms = [], []
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 16):
    os.environ["SMC_VMM_CHUNK_MB"] = str((64, 32, 128, 64)[i % 4])
    b = DevBuf(eng, need)
    t = ctypes.c_float()
    assert L.smc_mem_write_probe(eng.ctx, ctypes.c_void_p(b.data_ptr()), need, ctypes.byref(t)) == 0
    bufs.append(b); ms.append(t.value)
order = sorted(range(len(bufs)), key=lambda i: ms[i])
res = []
for i in (order[0], order[-1]):
    t = ctypes.c_float()
    L.smc_mem_write_probe(eng.ctx, ctypes.c_void_p(bufs[i].data_ptr()), need, ctypes.byref(t))
    res.append(t.value)
print("probe times of the candidates:", " ".join("%.3f" % x for x in ms))
print("FINAL eight dispatches: four into the fastest (%.3f ms now), four into the slowest (%.3f ms now)" % (res[0], res[1]), flush=True)
