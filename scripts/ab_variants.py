"""A/B of compile-time variants of the HIP library in ONE process, interleaved rounds (dev tool).
usage: ab_variants.py CFG NLOCI lib1.so lib2.so ..."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from smcounter_amd import synth, abi, _lib
cfg = synth.CONFIGS[sys.argv[1]]; n = int(sys.argv[2]); libs = sys.argv[3:]
P = synth.params_for(cfg)
db = synth.generate_native(cfg, 0, n)
dev = torch.device("cuda", 0)
planes = [torch.from_numpy(np.ascontiguousarray(x).view(np.int32)).to(dev) for x in (db.meta, db.umi, db.frag, db.dist, db.umi_start)]
rows = torch.empty(n * abi.ROW_DTYPE.itemsize, dtype=torch.uint8, device=dev)
vp = ctypes.c_void_p
H = []
for path in libs:
    L = ctypes.CDLL(os.path.join(ROOT, "smcounter_amd", path))
    L.smc_create.argtypes = [ctypes.c_int, ctypes.POINTER(vp)]
    L.smc_plan_create.argtypes = [vp, vp, ctypes.c_int64, ctypes.POINTER(vp)]
    L.smc_plan_run_words.argtypes = [vp, ctypes.POINTER(abi.SmcParams), vp, vp, vp, vp]
    L.smc_pack_words.argtypes = [vp, vp, vp, vp, vp]
    L.smc_last_error.restype = ctypes.c_char_p
    ctx, plan = vp(), vp()
    assert L.smc_create(0, ctypes.byref(ctx)) == 0, L.smc_last_error()
    loci = np.ascontiguousarray(db.loci)
    assert L.smc_plan_create(ctx, loci.ctypes.data, n, ctypes.byref(plan)) == 0, L.smc_last_error()
    H.append((path, L, plan))
cp = abi.c_params(P)
st = torch.cuda.current_stream()
words = torch.empty_like(planes[0])
assert H[0][1].smc_pack_words(H[0][2], planes[0].data_ptr(), planes[2].data_ptr(), words.data_ptr(), vp(st.cuda_stream)) == 0
torch.cuda.synchronize()
def run(L, plan):
    rc = L.smc_plan_run_words(plan, ctypes.byref(cp), words.data_ptr(), planes[4].data_ptr(), rows.data_ptr(), vp(st.cuda_stream))
    assert rc == 0, L.smc_last_error()
res = {p: [] for p, _, _ in H}
ref = None
for rnd in range(8):
    for path, L, plan in H:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); run(L, plan); b.record(); torch.cuda.synchronize()
        if rnd: res[path].append(a.elapsed_time(b))
        if rnd == 0:
            r = rows.cpu().numpy().tobytes()
            if ref is None: ref = r
            print(path, "rows identical to first variant:", r == ref)
bpl = 16 * cfg.depth + 360
for path in res:
    v = np.array(res[path])
    print("%-34s median %.3f ms  min %.3f ms  -> %.2f M loci/s, %.1f%% of 8 TB/s" % (
        path, np.median(v), v.min(), n / np.median(v) / 1e3, n * bpl / np.median(v) / 1e6 / 80))
