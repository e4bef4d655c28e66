"""The memory side of the walk's store pattern, by the write-only probe with other shapes: the same 2.4 GB written as runs of
41 / 82 / 164 words, by parts of 4 / 2 / 1 batches - what do longer runs and runs written side by side buy?  (dev tool)"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from smcounter_amd import engine
from smcounter_amd.engine import DevBuf
eng = engine.Engine(0)
eng.alloc_tries = 1
L = eng.L
need = 4 * 600_000_000
shapes = [(18, 4, 41), (72, 1, 41), (36, 2, 41), (9, 4, 82), (18, 2, 82), (36, 1, 82), (9, 2, 164), (18, 1, 164), (4, 4, 164), (1, 4, 656), (4, 1, 656)]
bufs = [DevBuf(eng, need) for _ in range(4)]
print("%-34s %s" % ("parts, batches, words per run", "  ".join("alloc %d" % k for k in range(len(bufs)))))
for sh in shapes:
    os.environ["SMC_PROBE_SHAPE"] = "%d,%d,%d" % sh
    row = []
    for b in bufs:
        ms = ctypes.c_float()
        assert L.smc_mem_write_probe(eng.ctx, ctypes.c_void_p(b.data_ptr()), need, ctypes.byref(ms)) == 0
        row.append(ms.value)
    print("%-34s %s   (%.2f TB/s)" % ("%d parts x %d batches x %d words" % sh, "  ".join("%7.3f" % x for x in row), need * (sh[0] * sh[1] * sh[2] / 3000.0) / min(row) / 1e9), flush=True)
