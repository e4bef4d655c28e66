"""bench.py's `from_alignments` leg: the hot path timed from where the reference's hot loop starts (smCounter.py:316) - a run's
ALIGNMENTS resident in HBM (what the BAM decoder hands over) -> smc_build_planes (sort, count, scan, the walk that writes the
planes: csrc/k_build_planes.inc) -> smc_plan_create_dev (launch plan made where the descriptors are) -> smc_plan_run_words -> rows in HBM.

Not part of the product path: a measurement harness (and its parity check against the decoder + host builder + CPU
restatement on a bounded sample).
"""
from __future__ import annotations

import ctypes
import os
import time

import numpy as np

from . import _lib, abi, synth
from .engine import DevBuf
from .features import LOCUS_DTYPE

HBM_PEAK_GBS = 8000.0


class AlignmentRun(object):
    """A run of synthetic alignments resident in HBM + the output arrays of smc_build_planes (only what the locus kernels read:
    the read words)."""

    def __init__(self, eng, cfg, params, n_loci, nthreads):
        self.eng, self.cfg, self.params = eng, cfg, params
        t0 = time.time()
        kw = {}
        if os.environ.get("SMC_FA_INDEL_RATE"):                              # (experiments: the share of alignments with an insertion / a deletion)
            kw = dict(p_ins_aln=float(os.environ["SMC_FA_INDEL_RATE"]), p_del_aln=float(os.environ["SMC_FA_INDEL_RATE"]))
        A = synth.generate_alignments(cfg, n_loci, params, nthreads=nthreads, **kw)
        self.t_gen = time.time() - t0
        self.A = A
        self.nl, self.ns, self.lo = A["nl"], A["n_slots"], int(A["start0"])
        self.reads = int(A["reads"])
        up = lambda a: DevBuf(eng, a.nbytes + 64).upload(a.view(np.uint8).reshape(-1))
        self.d_in = [up(A[k]) for k in ("aln", "cig", "seq", "qual", "loc")]
        run_ref = synth.aln_ref_fetch(self.lo, self.lo + self.nl)
        self.d_ref = up(np.frombuffer(run_ref.encode(), np.uint8).copy())
        self.words = DevBuf(eng, 4 * (self.ns + 64))
        self.uaux = [DevBuf(eng, 4 * (self.ns + self.nl + 64)) for _ in range(3)]
        self.d_loci = DevBuf(eng, self.nl * LOCUS_DTYPE.itemsize)
        self.xcap = 4 * self.nl + 4096
        self.d_x = DevBuf(eng, 20 * self.xcap)
        self.d_cnt = DevBuf(eng, 8)
        self.loc_host = np.ascontiguousarray(A["loc"])
        self.bi = abi.SmcBuildIn(self.d_in[0].data_ptr(), self.d_in[1].data_ptr(), self.d_in[2].data_ptr(), self.d_in[3].data_ptr(),
                                 self.d_in[4].data_ptr(), self.d_ref.data_ptr(), self.lo, self.nl, A["n_bc"], A["n_pair"],
                                 int(A["loc"]["n"].max()), len(A["aln"]), self.loc_host.ctypes.data)
        self.cp = abi.c_params(params)
        self.rows = DevBuf(eng, self.nl * abi.ROW_DTYPE.itemsize)
        self.lc = np.empty(self.nl, LOCUS_DTYPE)
        self.host_plan = bool(os.environ.get("SMC_FA_HOST_PLAN"))      # (measurement: descriptors back to the host, smc_plan_create)
        self.t = {"build_issue": 0.0, "descriptors_d2h": 0.0, "plan_create": 0.0, "run_issue": 0.0, "n": 0}

    def input_bytes(self):
        A = self.A
        return int(A["aln"].nbytes + A["cig"].nbytes + A["seq"].nbytes + A["qual"].nbytes + A["loc"].nbytes)

    def step(self, keep_plan=False):
        """build -> descriptors -> plan -> run; everything the product path does between the decoder and the rows."""
        eng, L = self.eng, self.eng.L
        t0 = time.perf_counter()
        _lib.check(L.smc_build_planes(eng.ctx, ctypes.byref(self.cp), ctypes.byref(self.bi), 0, 0, self.words.data_ptr(), None, None,
                                      None, None, self.uaux[0].data_ptr(), self.uaux[1].data_ptr(),
                                      self.uaux[2].data_ptr(), self.d_loci.data_ptr(), self.d_x.data_ptr(), self.xcap,
                                      self.d_cnt.data_ptr(), ctypes.c_void_p(0)), "smc_build_planes")
        t1 = time.perf_counter()
        if self.host_plan:
            self.d_loci.download(LOCUS_DTYPE, self.nl, out=self.lc)   # (behind the kernels on the default stream)
            t2 = time.perf_counter()
            plan = eng.make_plan(self.lc)
        else:
            t2 = t1
            plan = eng.make_plan_dev(self.d_loci, self.nl)            # binned where the descriptors are (waits for the builder)
        t3 = time.perf_counter()
        plan.run([self.words, self.uaux[0]], self.params, self.rows, stream=0)
        t4 = time.perf_counter()
        T = self.t
        T["build_issue"] += t1 - t0; T["descriptors_d2h"] += t2 - t1; T["plan_create"] += t3 - t2; T["run_issue"] += t4 - t3; T["n"] += 1
        if keep_plan:
            return plan
        plan.close()

    def status(self):
        return self.d_cnt.download(np.uint32, 2).tolist()

    def close(self):
        for b in self.d_in + [self.d_ref, self.words, self.d_loci, self.d_x, self.d_cnt, self.rows] + self.uaux:
            b.free()
        self.A = None


def parity_sample(run: AlignmentRun, n_check: int, nthreads: int, tmpdir: str):
    """The first `n_check` loci: the same alignments written as a BAM, through the real decoder, the host plane builder and
    the CPU restatement (oracle/smc_oracle.c), against the rows the device path has just produced for them."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "oracle"))
    import oracle_lib
    from . import bamio, fasta
    n_check = min(n_check, run.nl)
    bam, fa_path = os.path.join(tmpdir, "fa_sample.bam"), os.path.join(tmpdir, "fa_sample.fa")
    chrom, p0, p1 = synth.alignments_to_bam(run.A, bam, 0, n_check, fa_path)
    fa = fasta.FastaFile(fa_path)
    loci = [(chrom, str(p)) for p in range(p0, p1 + 1)]
    hb = [b for _, b in bamio.iter_device_batches_native(bam, fa, loci, run.params, max_reads=1 << 40, nthreads=nthreads)]
    assert len(hb) == 1 and hb[0].n_loci == n_check
    want, fragile, pi_all = oracle_lib.call_batch_mt(hb[0], abi.c_params(run.params), abi.ROW_DTYPE, nthreads, return_fragile=True,
                                                     return_pi_all=True)
    got = run.rows.download(abi.ROW_DTYPE, n_check)
    rep = abi.parity_report(got, want, fragile, pi_all)
    rep["detail"] = rep["detail"][:3]
    rep["checked_against"] = ("the same alignments as a BAM -> smc_bam_planes (host decoder + builder) -> oracle/smc_oracle.c, first %d "
                              "loci of the run" % n_check)
    return rep


def run_leg(eng, cfg_name: str, n_loci: int, steps: int, warmup: int, blocks: int, nthreads: int, parity_loci: int = 0,
            tmpdir: str = "/tmp"):
    cfg = synth.CONFIGS[cfg_name]
    params = synth.params_for(cfg)
    L = eng.L
    run = AlignmentRun(eng, cfg, params, n_loci, nthreads)
    for _ in range(max(1, warmup)):
        run.step()
    L.smc_device_sync(eng.ctx)
    st = run.status()
    for k in run.t:
        run.t[k] = 0
    _lib.check(L.smc_build_set_timing(eng.ctx, min(256, steps * blocks)), "smc_build_set_timing")
    times = []
    for _ in range(max(1, blocks)):
        L.smc_device_sync(eng.ctx)
        t0 = time.perf_counter()
        for _ in range(steps):
            run.step()
        L.smc_device_sync(eng.ctx)
        times.append(time.perf_counter() - t0)
    k_ms, k_n = ctypes.c_float(), ctypes.c_int32()
    _lib.check(L.smc_build_kernel_ms(eng.ctx, ctypes.byref(k_ms), ctypes.byref(k_n)), "smc_build_kernel_ms")
    L.smc_build_set_timing(eng.ctx, 0)
    # the locus kernels of the same planes, timed alone with the last plan
    plan = run.step(keep_plan=True)
    L.smc_device_sync(eng.ctx)
    plan.set_timing(8)
    for _ in range(8):
        plan.run([run.words, run.uaux[0]], params, run.rows, stream=0)
    c_ms = plan.kernel_ms()[0]
    plan.close()
    el = sorted(times)[len(times) // 2]
    slots = int(run.ns)
    # bytes the walk has to move: per pileup read one base + one quality in, the two plane words out; the alignment records
    # and CIGARs once per tile they touch (counted once here); umi_start and the descriptor per locus
    need = 2.0 * run.reads + 4.0 * slots + 36.0 * len(run.A["aln"]) + 4.0 * run.A["cig"].nbytes / 4 + 36.0 * run.nl
    n = max(1, run.t["n"])
    # HBM bytes per launch of the walk from the committed PMC passes (profiles/traffic.json), over THIS run's kernel time
    traffic, traffic_src = None, None
    tpath = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "traffic.json")
    if os.path.exists(tpath):
        import json
        rec = json.load(open(tpath)).get("fa:%s:%d" % (cfg_name, run.nl))
        if rec:
            traffic = rec["hbm_bytes_per_launch"] / (k_ms.value * 1e-3) / 1e9
            traffic_src = "profiles/traffic.json <- " + rec.get("source", "rocprofv3 --pmc")
    out = {
        "workload": "%s-shaped alignments: %d loci, %d alignments (%d barcodes, %d fragments), %d pileup reads, depth %.0f; "
                    "resident in HBM (%.2f GB)" % (cfg_name, run.nl, len(run.A["aln"]), run.A["n_bc"], run.A["n_pair"], run.reads,
                                                     run.reads / run.nl, run.input_bytes() / 1e9),
        "step": "smc_build_planes (read words) -> smc_plan_create_dev (binning on the device) -> smc_plan_run_words -> rows in HBM",
        "value": run.nl * steps / el, "unit": "loci/s", "ms_per_step": el / steps * 1e3,
        "blocks_ms_per_step": [round(t / steps * 1e3, 3) for t in times],
        "pileup_reads_per_s": run.reads * steps / el,
        "host_ms_per_step": {k: round(v / n * 1e3, 3) for k, v in run.t.items() if k != "n"},
        "k_call_v2_ms": c_ms,
        "roofline": {"bound": "hbm", "kernel": "k_bp_emit (the walk that writes the planes)", "kernel_ms": k_ms.value,
                     "kernel_samples": k_n.value, "needed_bytes_per_launch": need,
                     "achieved": need / (k_ms.value * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": need / (k_ms.value * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "frac_basis": "needed bytes: 2 B in + 4 B out (the read word) per pileup read, alignment records and CIGARs once",
                     "reads_per_s_kernel": run.reads / (k_ms.value * 1e-3), "traffic": traffic, "traffic_source": traffic_src},
        "builder_status": st,
        "generate_s": round(run.t_gen, 1),
    }
    if parity_loci:
        run.step()
        L.smc_device_sync(eng.ctx)
        out["parity"] = parity_sample(run, parity_loci, nthreads, tmpdir)
    run.close()
    return out


if __name__ == "__main__":
    import argparse
    import json
    from .engine import Engine
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C3")
    ap.add_argument("--loci", type=int, default=0)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--blocks", type=int, default=3)
    ap.add_argument("--parity-loci", type=int, default=512)
    a = ap.parse_args()
    eng = Engine(0)
    cfg = synth.CONFIGS[a.config]
    print(json.dumps(run_leg(eng, a.config, a.loci or cfg.n_loci, a.steps, a.warmup, a.blocks, len(os.sched_getaffinity(0)), a.parity_loci)))
