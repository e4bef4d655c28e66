"""Experiment: the from-alignments step in S chunks of loci, alternating over T host threads, each with its own context and
stream - the builder of chunk c+1 beside the locus kernel of chunk c.  usage: python scripts/chunk_probe.py [NLOCI] [S,S,...] [T]"""
import ctypes, os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from smcounter_amd import _lib, abi, synth
from smcounter_amd.engine import Engine
import bench_fa


def main():
    nl = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
    Ss = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1,2,4,8").split(",")]
    T = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    cfg = synth.CONFIGS["C3"]
    P = synth.params_for(cfg)
    eng = Engine(0)
    run = bench_fa.AlignmentRun(eng, cfg, P, nl, 8)
    for _ in range(3):
        run.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        run.step()
    torch.cuda.synchronize()
    print("whole run: %.3f ms/step" % ((time.perf_counter() - t0) * 100))
    want = run.rows.download(abi.ROW_DTYPE, run.nl).tobytes()
    engs = [eng] + [Engine(0) for _ in range(T - 1)]
    streams = [torch.cuda.Stream(device=0) for _ in range(T)]
    L = eng.L
    loc = run.loc_host
    rows2 = bench_fa.DevBuf(eng, run.nl * abi.ROW_DTYPE.itemsize)
    cnt = bench_fa.DevBuf(eng, 8 * 64)
    for S in Ss:
        edges = [run.nl * k // S // 64 * 64 for k in range(S)] + [run.nl]
        bis = []
        for c in range(S):
            c0, c1 = edges[c], edges[c + 1]
            b = abi.SmcBuildIn(run.bi.aln, run.bi.cig, run.bi.bq, run.bi.loc + 16 * c0, run.bi.refseq + c0,
                               run.lo + c0, c1 - c0, run.bi.n_bc, run.bi.n_pair, run.bi.max_depth, run.bi.n_aln,
                               run.loc_host.ctypes.data + 16 * c0)
            bis.append((c0, c1, b))
        xper = run.xcap // S

        def work(t):
            e, st = engs[t], ctypes.c_void_p(streams[t].cuda_stream)
            for c in range(t, S, T):
                c0, c1, b = bis[c]
                _lib.check(L.smc_build_planes(e.ctx, ctypes.byref(run.cp), ctypes.byref(b), 0, c0, run.words.data_ptr(), None, None,
                                              None, None, run.uaux[0].data_ptr(), run.uaux[1].data_ptr(), run.uaux[2].data_ptr(),
                                              run.d_loci.data_ptr() + c0 * bench_fa.LOCUS_DTYPE.itemsize,
                                              run.d_x.data_ptr() + 20 * xper * c, xper, cnt.data_ptr() + 8 * c, st), "build")
                h = ctypes.c_void_p()
                _lib.check(L.smc_plan_create_dev(e.ctx, run.d_loci.data_ptr() + c0 * bench_fa.LOCUS_DTYPE.itemsize, c1 - c0, st,
                                                 ctypes.byref(h)), "plan")
                _lib.check(L.smc_plan_run_words(h, ctypes.byref(run.cp), run.words.data_ptr(), run.uaux[0].data_ptr(),
                                                rows2.data_ptr() + c0 * abi.ROW_DTYPE.itemsize, st), "run")
                L.smc_plan_destroy(h)

        def step():
            th = [threading.Thread(target=work, args=(t,)) for t in range(1, T)]
            for x in th:
                x.start()
            work(0)
            for x in th:
                x.join()

        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 100
        got = rows2.download(abi.ROW_DTYPE, run.nl).tobytes()
        st_bits = cnt.download(np.uint32, 2 * S).reshape(S, 2)[:, 1]
        print("S=%d T=%d: %.3f ms/step (%.1f M loci/s)  rows identical: %s  status %s" % (S, T, ms, run.nl / ms / 1e3, got == want,
                                                                                       st_bits.tolist()))


main()
