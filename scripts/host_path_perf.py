"""PCIe-inclusive rate of the hot path: `smc_call_batch_host` on host (numpy, pageable) buffers - H2D of the planes the
kernel reads, kernels, D2H of the rows (dev tool, GPU box).  usage: host_path_perf.py [CFG] [n_loci]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from smcounter_amd import synth, engine
cfg = synth.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "C3"]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
P = synth.params_for(cfg)
db = synth.generate_native(cfg, 0, n, P)
eng = engine.Engine(0)
ts = []
for rep in range(6):
    t = time.perf_counter()
    R = eng.call_batch_host(db, P)
    ts.append(time.perf_counter() - t)
best = min(ts[1:])
moved = (db.meta.nbytes + db.frag.nbytes + db.umi_start.nbytes + db.loci.nbytes + R.nbytes) / 1e9
print("%s: %d loci x %d reads from host buffers: %.1f ms best of 5 (first %.1f ms) -> %.2f M loci/s, %.2f GB over PCIe (%.1f GB/s incl. kernels)"
      % (cfg.name, n, cfg.depth, best * 1e3, ts[0] * 1e3, n / best / 1e6, moved, moved / best))
