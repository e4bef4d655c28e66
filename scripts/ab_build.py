"""A/B of builds of the HIP library on smc_build_planes in ONE process, interleaved rounds: the same resident run of C3-shaped
alignments, the same output arrays, every library its own context (dev tool; the walk's time moves by +-15 % between processes on the
same box, so variants are only comparable side by side).
usage: ab_build.py NLOCI lib1.so lib2.so ...      (paths under smcounter_amd/)"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from smcounter_amd import _lib, abi, synth, engine
import bench_fa as fa_leg

n = int(sys.argv[1]); libs = sys.argv[2:]
cfgname = os.environ.get("AB_CFG", "C3")
eng = engine.Engine(0)
cfg = synth.CONFIGS[cfgname]
run = fa_leg.AlignmentRun(eng, cfg, synth.params_for(cfg), n, 8)
vp = ctypes.c_void_p
H = []
for spec in libs:
    path = spec.split("@")[0]                                      # lib.so[@LDS_PAD]: SMC_BP_LDS_PAD for that entry's launches
    L = ctypes.CDLL(os.path.join(ROOT, "smcounter_amd", path))
    L.smc_create.argtypes = [ctypes.c_int, ctypes.POINTER(vp)]
    L.smc_last_error.restype = ctypes.c_char_p
    L.smc_build_planes.argtypes = [vp, ctypes.POINTER(abi.SmcParams), ctypes.POINTER(abi.SmcBuildIn), ctypes.c_uint32, ctypes.c_uint32,
                                   vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, ctypes.c_int64, vp, vp]
    L.smc_build_set_timing.argtypes = [vp, ctypes.c_int]
    L.smc_build_kernel_ms.argtypes = [vp, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int32)]
    L.smc_device_sync.argtypes = [vp]
    ctx = vp()
    assert L.smc_create(0, ctypes.byref(ctx)) == 0, L.smc_last_error()
    H.append((spec, L, ctx))


def build(L, ctx, spec=""):
    os.environ["SMC_BP_LDS_PAD"] = spec.split("@")[1] if "@" in spec else "0"
    rc = L.smc_build_planes(ctx, ctypes.byref(run.cp), ctypes.byref(run.bi), 0, 0, run.words.data_ptr(), None, None, None, None,
                            run.uaux[0].data_ptr(), run.uaux[1].data_ptr(), run.uaux[2].data_ptr(), run.d_loci.data_ptr(),
                            run.d_x.data_ptr(), run.xcap, run.d_cnt.data_ptr(), None)
    assert rc == 0, L.smc_last_error()


REPS = 6
sig = {}
res = {p: ([], []) for p, _, _ in H}
for rnd in range(5):
    for path, L, ctx in H:
        build(L, ctx, path); L.smc_device_sync(ctx)
        L.smc_build_set_timing(ctx, REPS)
        t0 = time.perf_counter()
        for _ in range(REPS):
            build(L, ctx, path)
        L.smc_device_sync(ctx)
        wall = (time.perf_counter() - t0) / REPS * 1e3
        k_ms, k_n = ctypes.c_float(), ctypes.c_int32()
        L.smc_build_kernel_ms(ctx, ctypes.byref(k_ms), ctypes.byref(k_n))
        L.smc_build_set_timing(ctx, 0)
        if rnd:
            res[path][0].append(k_ms.value); res[path][1].append(wall)
        if rnd == 0:   # the planes every library wrote (same numbering rules -> same bytes)
            m = run.words.download(np.uint32, min(run.ns, 1 << 24))
            sig[path] = (int(m.astype(np.uint64).sum()), run.status())
for path in libs:
    e, w = res[path]
    print("%-24s k_bp_emit %s ms (median %.3f)   whole build %.3f ms   planes checksum %s" % (
        path, " ".join("%.3f" % x for x in e), sorted(e)[len(e) // 2], sorted(w)[len(w) // 2], sig[path]))
