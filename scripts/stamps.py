"""Diagnostic: per-phase cycle shares of k_call_loci (s_memtime stamps, separate -DSMC_STAMPS build).
Shares only - the stamped build's own run time is not a performance number."""
import ctypes, os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from smcounter_amd import build, synth, _lib
so = os.path.join(ROOT, "smcounter_amd", "libsmcounter_hip_stamps.so")
subprocess.check_call([build.hipcc_path()] + build.HIPCC_FLAGS + ["-DSMC_STAMPS", "-o", so, os.environ.get("SMC_SRC", build.SRC)])
_lib.LIB_PATH = so
from smcounter_amd import engine
cfgname = sys.argv[1] if len(sys.argv) > 1 else "C3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
cfg = synth.CONFIGS[cfgname]; P = synth.params_for(cfg)
db = synth.generate_native(cfg, 0, n)
eng = engine.Engine(0)
planes = eng.upload(db); plan = eng.make_plan(db.loci); rows = plan.alloc_rows()
plan.run(planes, P, rows); torch.cuda.synchronize()
st = (ctypes.c_ulonglong * 16)()
eng.L.smc_debug_stamps(st, 1)
for _ in range(3):
    plan.run(planes, P, rows)
torch.cuda.synchronize()
eng.L.smc_debug_stamps(st, 1)
v = np.array(list(st)[:10], float)
names = ["S0 init", "P1 scan", "S2 barcodes", "-", "R merge", "(downsample)", "U0 simple barcodes", "U1 queued barcodes", "-", "E rank+write"]
for k, nm in enumerate(names):
    print("%-16s %6.1f%%  %8.0f cycles/locus" % (nm, 100 * v[k] / v.sum(), v[k] / (3 * n)))
print("total %.0f cycles/locus (thread-0 clock64 ticks)" % (v.sum() / (3 * n)))
