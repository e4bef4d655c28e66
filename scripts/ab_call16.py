"""A/B of compile-time variants of the HIP library on the LOCUS KERNELS with 16-bit read words, in ONE process, interleaved rounds
(dev tool; the 32-bit twin is ab_variants.py).  usage: ab_call16.py CFG NLOCI lib1.so lib2.so ...   (libraries under smcounter_amd/)"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from smcounter_amd import synth, abi, devplanes
cfg = synth.CONFIGS[sys.argv[1]]; n = int(sys.argv[2]); libs = sys.argv[3:]
P = synth.params_for(cfg)
db = synth.generate_native(cfg, 0, n)
dev = torch.device("cuda", 0)
w32 = devplanes.pack_words_host(db.meta, db.frag, db.loci)
w16 = devplanes.words16_from_32(w32)
assert w16 is not None
words = torch.from_numpy(w16.view(np.int16)).to(dev)
ustart = torch.from_numpy(np.ascontiguousarray(db.umi_start).view(np.int32)).to(dev)
rows = torch.empty(n * abi.ROW_DTYPE.itemsize, dtype=torch.uint8, device=dev)
vp = ctypes.c_void_p
H = []
for path in libs:
    L = ctypes.CDLL(os.path.join(ROOT, "smcounter_amd", path))
    L.smc_create.argtypes = [ctypes.c_int, ctypes.POINTER(vp)]
    L.smc_plan_create.argtypes = [vp, vp, ctypes.c_int64, ctypes.POINTER(vp)]
    L.smc_plan_run_words16.argtypes = [vp, ctypes.POINTER(abi.SmcParams), vp, vp, vp, vp]
    L.smc_last_error.restype = ctypes.c_char_p
    ctx, plan = vp(), vp()
    assert L.smc_create(0, ctypes.byref(ctx)) == 0, L.smc_last_error()
    loci = np.ascontiguousarray(db.loci)
    assert L.smc_plan_create(ctx, loci.ctypes.data, n, ctypes.byref(plan)) == 0, L.smc_last_error()
    H.append((path, L, plan))
cp = abi.c_params(P)
st = torch.cuda.current_stream()
def run(L, plan):
    rc = L.smc_plan_run_words16(plan, ctypes.byref(cp), words.data_ptr(), ustart.data_ptr(), rows.data_ptr(), vp(st.cuda_stream))
    assert rc == 0, L.smc_last_error()
res = {p: [] for p, _, _ in H}
ref = None
for rnd in range(10):
    for path, L, plan in (H if rnd % 2 == 0 else H[::-1]):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); run(L, plan); b.record(); torch.cuda.synchronize()
        if rnd: res[path].append(a.elapsed_time(b))
        if rnd == 0:
            r = rows.cpu().numpy().tobytes()
            if ref is None: ref = r
            print(path, "rows identical to first variant:", r == ref)
for path in res:
    v = np.array(res[path])
    print("%-34s median %.4f ms  min %.4f ms" % (path, np.median(v), v.min()))
