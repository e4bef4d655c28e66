#!/usr/bin/env python3
"""bench.py - throughput of the per-locus hot path on synthetic pileups of stated depth.

Metric (BASELINE.json): loci/s at fixed read-depth x reads-per-UMI.  Workload at every N:
BASELINE.json configs[2] "synthetic 200k loci, 3000x depth, 50 UMIs/locus, 60 rpb" PER GPU (weak
scaling: rank r calls loci [r*200k, (r+1)*200k) of the same seeded config), inputs resident in HBM
before the timed region.  A step = one pass of the hot path (k_call_v2 bins + k_filter_loci) over
the rank's batch, followed, for N > 1, by the gather of the rows to rank 0 (RCCL; the gather of a step
overlaps the next step's kernels, two row buffers per rank - every step's rows are gathered inside the
timed region).  `--scaling strong --config C4`: BASELINE's configs[3], 1 M loci IN TOTAL sharded over
the ranks (it fits one MI355X as well: the N = 1 anchor of that curve).

The timed region is a block of EXACTLY --steps steps between barrier + synchronize on both sides; the
block is repeated --blocks times (default 5) and `value` / `ms_per_step` come from the MEDIAN block
(every block's ms_per_step is listed under `blocks`): one 20-step block is ~30 ms, too short to trust
to a percent.

Prints ONE JSON line on rank 0 (see the task contract): value = loci of all ranks / max-over-ranks
time; roofline = the bytes the dominant kernel has to move per launch over its mean HIP-event duration,
against 8 TB/s (`frac`, with `frac_basis`; SURVEY.md 8d's 16 B/read figure beside it as
`frac_survey_8d`); cpu_baseline = CPU legs timed on a bounded sample of the same workload (rank 0,
N = 1 only; physical cores stated, the single-process rate beside the pool's); parity = EVERY row of the
run against the CPU restatement with the number of loci whose order-dependent fields were excused, the
loci that reached filterVariants, the Fisher tests run and the largest p-value difference;
other_configs = the other single-GPU shapes (C2, C5, X3 - 30 % of the loci with a candidate -, EX - the
statistics of the reference's own example run), each timed the same way in the same process;
from_alignments = the same C3 depth shape timed from where the reference's hot loop starts
(smCounter.py:316): alignments resident in HBM -> plane builder -> launch plan -> hot path
(smcounter_amd/fa_leg.py), with the plane-writing kernel's own roofline.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# heavy imports happen inside main(): the CPU leg's spawned workers re-import this module
HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def needed_bytes(loci) -> float:
    """HBM bytes one launch of the hot path has to move for these loci: the read words (one uint32 per read slot: allele,
    quality, fragment start, class), umi_start, the 32-byte descriptor + 4-byte launch-order entry, and the row written."""
    import numpy as np
    from smcounter_amd import abi
    slots = ((loci["n_reads"].astype(np.int64) + 3) // 4 * 4).sum()
    return float(4 * slots + 4 * (loci["n_umi"].astype(np.int64) + 1).sum() + (32 + 4 + abi.ROW_DTYPE.itemsize) * len(loci))


class Resident(object):
    """One config's batch resident in HBM: what the kernels read - the read words (packed on the device from the synthetic
    batch's meta and frag planes, smc_pack_words, before anything is timed) and umi_start - + the plan; optionally the CPU
    restatement's rows of every chunk (all host cores), kept for the parity pass."""

    def __init__(self, eng, cfg, params, lo, hi, chunk, nthreads, dev, oracle=None):
        import numpy as np
        import torch
        from smcounter_amd import abi, synth
        n_loc = hi - lo
        stride = (cfg.depth + 3) // 4 * 4
        meta = torch.empty(n_loc * stride, dtype=torch.int32, device=dev)
        frag = torch.empty(n_loc * stride, dtype=torch.int32, device=dev)
        self.umi_start = torch.empty(n_loc * (cfg.n_umi + 1), dtype=torch.int32, device=dev)
        loci_parts, self.want = [], []
        for c0 in range(lo, hi, chunk):
            c1 = min(hi, c0 + chunk)
            db = synth.generate_native(cfg, c0, c1, params, nthreads=nthreads)
            off = (c0 - lo) * stride
            # (the umi and dist planes hold the raw fields the CPU restatement reads; the kernels do not: smcounter_hip.h)
            meta[off:off + db.n_slots].copy_(torch.from_numpy(db.meta.view(np.int32)))
            frag[off:off + db.n_slots].copy_(torch.from_numpy(db.frag.view(np.int32)))
            uoff = (c0 - lo) * (cfg.n_umi + 1)
            self.umi_start[uoff:uoff + len(db.umi_start)].copy_(torch.from_numpy(db.umi_start.view(np.int32)))
            if oracle is not None:
                self.want.append(oracle.call_batch_mt(db, abi.c_params(params), abi.ROW_DTYPE, nthreads, return_fragile=True,
                                                      return_pi_all=True))
            loc = db.loci.copy()
            loc["read_off4"] += off // 4
            loc["umi_off"] += uoff
            loci_parts.append(loc)
        self.loci = np.concatenate(loci_parts)
        self.plan = eng.make_plan(self.loci)
        self.words = torch.empty(n_loc * stride, dtype=torch.int32, device=dev)
        self.plan.pack_words(meta, frag, self.words)
        torch.cuda.synchronize()
        del meta, frag
        self.planes = [self.words, self.umi_start]
        self.params = params

    def run(self, rows):
        return self.plan.run(self.planes, self.params, rows)

    def parity(self, rows):
        import numpy as np
        from smcounter_amd import abi
        got = self.plan.download(rows)
        tot = {"loci": 0, "mismatches": 0, "fragile_skipped": 0, "near_tie_skipped": 0, "pi_max_abs_diff": 0.0, "loci_filtered": 0,
               "fisher_tests_run": 0, "p_max_abs_diff": 0.0, "detail": []}
        lo = 0
        for want, fragile, pi_all in self.want:
            rep = abi.parity_report(got[lo:lo + len(want)], want, fragile, pi_all)
            for k in ("loci", "mismatches", "fragile_skipped", "near_tie_skipped", "loci_filtered", "fisher_tests_run"):
                tot[k] += rep[k]
            for k in ("pi_max_abs_diff", "p_max_abs_diff"):
                tot[k] = max(tot[k], rep[k])
            tot["detail"] += rep["detail"]
            lo += len(want)
        from smcounter_amd import rows as _rows
        tot["pi_boundary_loci"] = int(len(_rows.pi_boundary_loci(got)))      # PI within 1e-8 of a printing / gating boundary
        tot["detail"] = tot["detail"][:3]
        tot["checked_against"] = "oracle/smc_oracle.c on all host cores, every locus of the run"
        return tot

    def close(self):
        self.plan.close()
        self.words = self.umi_start = self.planes = None


def roofline_block(plan_loci, k_ms, k_n, k_loci, k_reads, cfg_key):
    """`achieved` / `frac`: the bytes the dominant kernel HAS to move per launch (the read words - 4 B per read slot -,
    umi_start, descriptors, rows) over its mean HIP-event duration, against the 8 TB/s peak.  SURVEY.md 8d's algorithmic figure
    (16 B per read + 360 B per locus: the four raw-field planes of the layout, which the plane builder digests into the one
    word per read the kernels load) is kept beside it as achieved_survey_8d / frac_survey_8d; it can exceed what a copy
    achieves on this part and is not a fraction of anything the kernel does."""
    alg_bytes = 16.0 * k_reads + 360.0 * k_loci              # per launch of the dominant kernel
    need = needed_bytes(plan_loci)
    achieved = need / (k_ms * 1e-3) / 1e9
    # HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/traffic.json), expressed like `achieved`:
    # GB/s over THIS run's measured kernel duration.  A constant read from the committed profile, not measured here.
    traffic, traffic_bytes, src = None, None, None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        rec = json.load(open(tpath)).get(cfg_key)
        if rec:
            traffic_bytes = rec["hbm_bytes_per_launch"]
            traffic = traffic_bytes / (k_ms * 1e-3) / 1e9
            src = "profiles/traffic.json <- " + rec.get("source", "rocprofv3 --pmc")
    return {"bound": "hbm", "kernel": "k_call_v2", "achieved": achieved, "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "frac_basis": "needed bytes per launch (4 B per read slot: the read word; + umi_start + descriptor + row) / kernel time / 8 TB/s",
            "traffic": traffic, "traffic_source": src,
            "kernel_ms": k_ms, "kernel_samples": k_n, "loci_per_launch": k_loci,
            "needed_bytes_per_launch": need, "hbm_bytes_per_launch_pmc": traffic_bytes,
            # the same launch charged the 8 B per read slot it moved until round 3 (meta + frag words): for comparison with
            # earlier rounds' `frac` only - the kernel no longer moves those bytes
            "frac_on_round2_bytes": (need + 4.0 * float(((plan_loci["n_reads"].astype("int64") + 3) // 4 * 4).sum())) / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "alg_bytes_per_launch_survey_8d": alg_bytes, "achieved_survey_8d": alg_bytes / (k_ms * 1e-3) / 1e9,
            "frac_survey_8d": alg_bytes / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "note": "frac_survey_8d counts SURVEY 8d's 16 B/read (four raw-field planes); the kernels load 4 B/read - the read word "
                    "the plane builder folds them into (the from_alignments leg times it; rounds 1-3 loaded 8 B/read: meta + frag)"}


def physical_cores():
    """Physical cores among the CPUs this process may run on (/proc/cpuinfo: distinct (physical id, core id))."""
    try:
        allowed = os.sched_getaffinity(0)
        cores, cpu, phys, core = set(), None, None, None
        for line in open("/proc/cpuinfo"):
            if line.startswith("processor"):
                cpu = int(line.split(":")[1]); phys = core = None
            elif line.startswith("physical id"):
                phys = int(line.split(":")[1])
            elif line.startswith("core id"):
                core = int(line.split(":")[1])
                if cpu in allowed:
                    cores.add((phys, core))
        return len(cores) or len(allowed)
    except Exception:
        return os.cpu_count() or 1


def main():
    global np, torch, dist, abi, engine, synth, smcdist
    import numpy as np
    from smcounter_amd import abi, synth
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--blocks", type=int, default=5, help="timed blocks of --steps steps each; value = the median block")
    ap.add_argument("--config", default="C3", help="synthetic config (C2, C3, C5); C3 is the metric's")
    ap.add_argument("--loci-per-gpu", type=int, default=0, help="override (default: the config's size)")
    ap.add_argument("--chunk", type=int, default=25000, help="loci generated/uploaded per chunk")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the every-row check against the CPU restatement")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the C2 / C5 / X3 / EX one-liners")
    ap.add_argument("--no-from-alignments", action="store_true", help="skip the leg that starts from alignments (plane builder + hot path)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak: the config's loci PER GPU (default, the driver's scaling run); strong: the config's loci in total, "
                         "sharded over the ranks (e.g. --config C4: 1 M loci; fits one MI355X too)")
    ap.add_argument("--rows", choices=("gather", "resident"), default="gather",
                    help="N > 1: gather every step's rows to rank 0 (default; overlapped with the next step) or leave "
                         "them in each rank's HBM (diagnostic: isolates the collective)")
    ap.add_argument("--wire", choices=("packed", "full"), default="packed",
                    help="N > 1: what travels to rank 0 - the packed wire rows (default) or the full 432-byte rows")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch N>1 through torch.distributed.run)"
                         % (a.gpus, world))
    # The CPU leg runs FIRST, before this process touches the GPU: it starts worker processes.
    cpu = None
    if world == 1 and not a.no_cpu_baseline:
        cpu = cpu_leg(a)
    import torch
    import torch.distributed as dist
    from smcounter_amd import engine
    from smcounter_amd import dist as smcdist
    # (SMC_BENCH_SHARE_GPU=1: every rank on GPU 0 with the gloo backend - a FUNCTIONAL check of the N > 1 path on a box with
    # one GPU, where RCCL refuses two ranks on a device; its numbers are not a measurement of anything)
    share_gpu = bool(os.environ.get("SMC_BENCH_SHARE_GPU"))
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    # (SMC_BENCH_FORCE_DIST=1 under a 1-process torch.distributed.run exercises the collective path on one GPU)
    use_dist = world > 1 or bool(os.environ.get("SMC_BENCH_FORCE_DIST"))
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if share_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    cfg = synth.CONFIGS[a.config]
    params = synth.params_for(cfg)
    n_loc = a.loci_per_gpu or cfg.n_loci
    total_loci = n_loc * world if a.scaling == "weak" else n_loc
    lo, hi = smcdist.shard_range(total_loci, rank, world)         # contiguous, equal shares of the ordered locus list
    n_mine = hi - lo
    eng = engine.Engine(local_rank)
    dev = torch.device("cuda", local_rank)
    oracle = None
    if world == 1 and not a.no_parity:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle_lib as oracle

    # ---- build the rank's batch in HBM, chunk by chunk (host RAM stays bounded)
    t0 = time.time()
    nthreads = max(1, len(os.sched_getaffinity(0)) // max(1, world))
    res = Resident(eng, cfg, params, lo, hi, a.chunk, nthreads, dev, oracle)
    plan = res.plan
    rows = plan.alloc_rows()
    torch.cuda.synchronize()
    t_build = time.time() - t0

    # N > 1: the rows of step i travel to rank 0 (RCCL, its own stream) while step i + 1 computes - two buffers per rank,
    # a buffer is reused only after its gather has completed (dist.RowPipeline).  What travels is the packed wire row
    # (smc_pack_rows: every printed column, a third of the bytes) unless --wire full.
    gather = use_dist and a.rows == "gather"
    packed = gather and a.wire == "packed"
    if packed:
        wires = [plan.alloc_wire(), plan.alloc_wire()]

        def produce(buf):
            res.run(rows)
            plan.pack(rows, buf)
        pipe = smcdist.RowPipeline(wires, collective=True)
    else:
        def produce(buf):
            res.run(buf)
        pipe = smcdist.RowPipeline([rows, plan.alloc_rows()] if gather else [rows], collective=gather)

    def step():
        pipe.step(produce)

    for _ in range(a.warmup):
        step()
    pipe.drain()
    torch.cuda.synchronize()
    blocks = []
    plan.set_timing(min(a.steps * a.blocks, 256))
    for _ in range(max(1, a.blocks)):
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        t_start = time.perf_counter()
        for _ in range(a.steps):
            step()
        pipe.drain()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        elapsed = time.perf_counter() - t_start
        if use_dist:
            t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if share_gpu else dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        blocks.append(elapsed)
    k_ms, k_n, k_loci, k_reads = plan.kernel_ms()
    plan.set_timing(0)
    elapsed = sorted(blocks)[len(blocks) // 2]                   # the median block

    out = None
    if rank == 0:
        out = {
            "metric": "loci/sec at fixed read-depth x rpb", "value": total_loci * a.steps / elapsed,
            "unit": "loci/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True, "scaling": a.scaling,
            "vs_baseline": None, "dtype": "u8/u32 scan + f64 posterior", "data": "synthetic",
            "config": {"workload": "%s: %d loci%s x %d reads (%d UMIs x %d rpb), seed %d"
                       % (cfg.name, n_loc, "/GPU" if a.scaling == "weak" else " in total", cfg.depth, cfg.n_umi, cfg.rpb, cfg.seed),
                       "loci_total": total_loci,
                       "parallelism": "loci sharded x%d, %s" % (world, ("%s rows gathered to rank 0" % ("packed wire" if packed else "full"))
                                                                 if gather else ("single GPU" if world == 1 else
                                                                                 "rows left in each rank's HBM (--rows resident)")),
                       "build_s": round(t_build, 1), **({"note": "SMC_BENCH_SHARE_GPU: all ranks on one GPU through gloo - a functional "
                                                                         "check of the N > 1 path, not a measurement"} if share_gpu else {})},
            "blocks": {"n": len(blocks), "steps_each": a.steps, "ms_per_step": [round(b / a.steps * 1e3, 4) for b in blocks],
                       "value_from": "median block", "spread_pct": round(100.0 * (max(blocks) - min(blocks)) / elapsed, 2)},
            "roofline": roofline_block(res.loci, k_ms, k_n, k_loci, k_reads, "%s:%d" % (a.config, n_mine)),
            "host_buffers": "inputs resident in HBM when the timed region starts; handing host buffers instead "
                            "(smc_call_batch_host: H2D of 8 B/read + kernels + D2H of the rows) measured 1.94 M loci/s on C3 "
                            "(DESIGN.md section 5) - PCIe-bound, never `value`",
        }
        if cpu is not None:
            out["cpu_baseline"], out["cpu_baseline_c"] = cpu["python_pool"], cpu["c_port"]
            out["cpu_baseline_c_all_cores"] = cpu["c_port_all_cores"]
            out["cpu_baseline_single_process"] = cpu["python_single"]
            out["cpu_baseline_pool_chunked"] = cpu["python_pool_chunked"]
        if oracle is not None:
            res.run(rows)
            torch.cuda.synchronize()
            out["parity"] = res.parity(rows)
    res.close()
    del rows, pipe
    if rank == 0 and world == 1 and not a.no_other_configs:
        out["other_configs"] = {}
        for name in ("C2", "C5", "X3", "EX"):           # X3: 30 % of the loci reach the filters; EX: the example run's statistics
            if name != a.config:
                out["other_configs"][name] = other_config(eng, name, a, dev, nthreads, oracle)
        if not a.no_from_alignments:
            from smcounter_amd import fa_leg
            torch.cuda.empty_cache()
            out["from_alignments"] = fa_leg.run_leg(eng, "C3", a.loci_per_gpu or synth.CONFIGS["C3"].n_loci, max(3, a.steps // 4),
                                                    2, 3, nthreads, parity_loci=0 if a.no_parity else 512)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def other_config(eng, name, a, dev, nthreads, oracle):
    """The other single-GPU BASELINE configs, same procedure (resident inputs, warm-up, blocks of --steps steps, median
    block, HIP-event time of the dominant kernel, every row checked), reported as one short object each."""
    cfg = synth.CONFIGS[name]
    params = synth.params_for(cfg)
    torch.cuda.empty_cache()
    res = Resident(eng, cfg, params, 0, cfg.n_loci, max(1, min(a.chunk, 2_000_000_00 // (16 * cfg.depth))), nthreads, dev, oracle)
    rows = res.plan.alloc_rows()
    for _ in range(a.warmup):
        res.run(rows)
    blocks = []
    res.plan.set_timing(min(a.steps * a.blocks, 256))
    for _ in range(max(1, a.blocks)):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            res.run(rows)
        torch.cuda.synchronize()
        blocks.append(time.perf_counter() - t0)
    k_ms, k_n, k_loci, k_reads = res.plan.kernel_ms()
    res.plan.set_timing(0)
    el = sorted(blocks)[len(blocks) // 2]
    rf = roofline_block(res.loci, k_ms, k_n, k_loci, k_reads, "%s:%d" % (name, cfg.n_loci))
    o = {"workload": "%s: %d loci x %d reads (%d UMIs x %d rpb), seed %d" % (name, cfg.n_loci, cfg.depth, cfg.n_umi, cfg.rpb, cfg.seed),
         "value": cfg.n_loci * a.steps / el, "unit": "loci/s", "ms_per_step": el / a.steps * 1e3,
         "blocks_ms_per_step": [round(b / a.steps * 1e3, 4) for b in blocks],
         "roofline": {k: rf[k] for k in ("achieved", "frac", "frac_basis", "kernel_ms", "loci_per_launch", "needed_bytes_per_launch",
                                         "achieved_survey_8d", "frac_survey_8d")}}
    if oracle is not None:
        res.run(rows)
        torch.cuda.synchronize()
        o["parity"] = res.parity(rows)
    res.close()
    return o


def cpu_leg(a):
    """CPU baselines on a bounded sample of the same workload (first chunk of the config):
    * python_pool - oracle/vc_port.py, the pure-Python restatement of vc(), driven like the reference's
      main(): multiprocessing.Pool(all host cores), one task per locus, on the first 2000+ loci;
    * c_port - oracle/smc_oracle.c on one core over the whole first chunk;
    * c_port_all_cores - the same C code on every host core (threads over contiguous locus ranges)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_lib
    import vc_port
    cfg = synth.CONFIGS[a.config]
    params = synth.params_for(cfg)
    n_loc = a.loci_per_gpu or cfg.n_loci
    sample = synth.generate_native(cfg, 0, min(a.chunk, n_loc), params)
    n = sample.n_loci
    t = time.perf_counter()
    ref_rows = oracle_lib.call_batch(sample, abi.c_params(params), abi.ROW_DTYPE)
    dt_c = time.perf_counter() - t
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    reps = 5
    t = time.perf_counter()
    for _ in range(reps):
        mt_rows = oracle_lib.call_batch_mt(sample, abi.c_params(params), abi.ROW_DTYPE, cores)
    dt_cmt = (time.perf_counter() - t) / reps
    assert mt_rows.tobytes() == ref_rows.tobytes()
    phys = physical_cores()
    # one process, no pool: what one core does with the pure-Python restatement (no task pickling, no result pipes)
    n_one = min(n, 120)
    t = time.perf_counter()
    vc_port.call_batch(sample, params, n_cpu=1, loci=range(n_one))
    dt_one = time.perf_counter() - t
    n_py = min(n, max(2000, 40 * cores))
    pool = vc_port.make_pool(cores)                 # started and warmed outside the timed region
    vc_port.call_config(a.config, params, range(cores), pool)          # (first import of the generator in every worker)
    t = time.perf_counter()
    got = vc_port.call_config(a.config, params, range(n_py), pool)
    dt_py = time.perf_counter() - t
    n_ch = min(n, 40 * cores)
    t = time.perf_counter()
    vc_port.call_config_chunked(a.config, params, 0, n_ch, pool, 20)
    dt_ch = time.perf_counter() - t
    pool.close()
    pool.join()
    want = vc_port.call_batch(sample, params, n_cpu=1, loci=range(4))
    assert [r["cvg"] for r in got[:4]] == [r["cvg"] for r in want] and [r["pi"] for r in got[:4]] == [r["pi"] for r in want]
    per_core_pool = n_py / dt_py / phys
    return {
        "python_pool": {"value": n_py / dt_py, "unit": "loci/s", "cores": phys, "logical_cpus": cores, "kind": "port",
                        "per_physical_core": per_core_pool,
                        "sample": "first %d loci of the same workload, pure-Python port oracle/vc_port.py under "
                                  "multiprocessing.Pool(%d), one task per locus as smCounter.py:683-685, every worker making its own "
                                  "locus's input (the reference's worker reads its own BAM region), %.1f s (pool already started)"
                                  % (n_py, cores, dt_py)},
        "python_pool_chunked": {"value": n_ch / dt_ch, "unit": "loci/s", "cores": phys, "logical_cpus": cores, "kind": "port",
                                "sample": "first %d loci, the same pool with 20 consecutive loci per task, %.1f s: without the parent's "
                                          "per-task round trip (~ 0.4 ms, invisible at the reference's 0.02 - 2 s per locus)" % (n_ch, dt_ch)},
        "python_single": {"value": n_one / dt_one, "unit": "loci/s", "cores": 1, "kind": "port",
                          "sample": "first %d loci, oracle/vc_port.py in this process (no pool), %.1f s; SURVEY.md section 6 timed the "
                                    "imported reference at ~51 loci/s on one core of the build container for this shape"
                                    % (n_one, dt_one)},
        "c_port": {"value": n / dt_c, "unit": "loci/s", "cores": 1, "kind": "port",
                   "sample": "first %d loci, C restatement oracle/smc_oracle.c, 1 thread, %.1f s" % (n, dt_c)},
        "c_port_all_cores": {"value": n / dt_cmt, "unit": "loci/s", "cores": phys, "logical_cpus": cores, "kind": "port",
                             "sample": "first %d loci, C restatement oracle/smc_oracle.c on %d threads (contiguous locus "
                                       "ranges), mean of %d passes, %.3f s each" % (n, cores, reps, dt_cmt)}}


if __name__ == "__main__":
    main()
