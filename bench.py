#!/usr/bin/env python3
"""bench.py - throughput of the per-locus hot path on synthetic pileups of stated depth.

Metric (BASELINE.json): loci/s at fixed read-depth x reads-per-UMI.  Workload at every N:
BASELINE.json configs[2] "synthetic 200k loci, 3000x depth, 50 UMIs/locus, 60 rpb" PER GPU (weak
scaling: rank r calls loci [r*200k, (r+1)*200k) of the same seeded config), inputs resident in HBM
before the timed region.  A step = one pass of the hot path (k_call_loci bins + k_filter_loci) over
the rank's batch, followed, for N > 1, by the gather of the fixed-width rows to rank 0 (RCCL; the gather of a
step overlaps the next step's kernels, two row buffers per rank - every step's rows are gathered inside the timed
region).

Prints ONE JSON line on rank 0 (see the task contract): value = loci of all ranks / max-over-ranks
time; roofline = algorithmic bytes (16 B/read + 360 B/locus) of the dominant kernel
over its mean HIP-event duration, against 8 TB/s; cpu_baseline = the C restatement under oracle/
timed on one host core on a bounded sample of the same workload (rank 0, N = 1 only).
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


def main():
    global np, torch, dist, abi, engine, synth, smcdist
    import numpy as np
    from smcounter_amd import abi, synth
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="C3", help="synthetic config (C2, C3, C5); C3 is the metric's")
    ap.add_argument("--loci-per-gpu", type=int, default=0, help="override (default: the config's size)")
    ap.add_argument("--chunk", type=int, default=25000, help="loci generated/uploaded per chunk")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--rows", choices=("gather", "resident"), default="gather",
                    help="N > 1: gather every step's rows to rank 0 (default; overlapped with the next step) or leave "
                         "them in each rank's HBM (diagnostic: isolates the collective)")
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
    torch.cuda.set_device(local_rank)
    # (SMC_BENCH_FORCE_DIST=1 under a 1-process torch.distributed.run exercises the collective path on one GPU)
    use_dist = world > 1 or bool(os.environ.get("SMC_BENCH_FORCE_DIST"))
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    cfg = synth.CONFIGS[a.config]
    params = synth.params_for(cfg)
    n_loc = a.loci_per_gpu or cfg.n_loci
    lo, hi = smcdist.shard_range(n_loc * world, rank, world)      # contiguous, equal (weak scaling)
    eng = engine.Engine(local_rank)
    dev = torch.device("cuda", local_rank)

    # ---- build the rank's batch in HBM, chunk by chunk (host RAM stays bounded)
    t0 = time.time()
    stride = (cfg.depth + 3) // 4 * 4
    planes = [torch.empty(n_loc * stride, dtype=torch.int32, device=dev) for _ in range(4)]
    planes.append(torch.empty(n_loc * (cfg.n_umi + 1), dtype=torch.int32, device=dev))      # umi_start
    loci_parts, sample = [], None
    nthreads = max(1, (os.cpu_count() or 1) // max(1, world))
    for c0 in range(lo, hi, a.chunk):
        c1 = min(hi, c0 + a.chunk)
        db = synth.generate_native(cfg, c0, c1, params, nthreads=nthreads)
        off = (c0 - lo) * stride
        for pl, src in zip(planes, (db.meta, db.umi, db.frag, db.dist)):
            pl[off:off + db.n_slots].copy_(torch.from_numpy(src.view(np.int32)))
        uoff = (c0 - lo) * (cfg.n_umi + 1)
        planes[4][uoff:uoff + len(db.umi_start)].copy_(torch.from_numpy(db.umi_start.view(np.int32)))
        loc = db.loci.copy()
        loc["read_off4"] += off // 4
        loc["umi_off"] += uoff
        loci_parts.append(loc)
        if sample is None:
            sample = db
    loci = np.concatenate(loci_parts)
    plan = eng.make_plan(loci)
    rows = plan.alloc_rows()
    torch.cuda.synchronize()
    t_build = time.time() - t0

    # N > 1: the rows of step i travel to rank 0 (RCCL, its own stream) while step i + 1 computes - two row
    # buffers per rank, a buffer is reused only after its gather has completed (dist.RowPipeline)
    gather = use_dist and a.rows == "gather"
    pipe = smcdist.RowPipeline([rows, plan.alloc_rows()] if gather else [rows], collective=gather)

    def step():
        pipe.step(lambda buf: plan.run(planes, params, buf))

    drain = pipe.drain

    for _ in range(a.warmup):
        step()
    drain()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    plan.set_timing(min(a.steps, 64))
    torch.cuda.synchronize()
    t_start = time.perf_counter()
    for _ in range(a.steps):
        step()
    drain()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t_start
    k_ms, k_n, k_loci, k_reads = plan.kernel_ms()
    plan.set_timing(0)
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    out = None
    if rank == 0:
        total_loci = n_loc * world
        ms_per_step = elapsed / a.steps * 1e3
        alg_bytes = 16.0 * k_reads + 360.0 * k_loci              # per launch of the dominant kernel
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9
        # HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/traffic.json), expressed
        # like `achieved`: GB/s over this run's measured kernel duration
        traffic, traffic_bytes = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            rec = json.load(open(tpath)).get("%s:%d" % (a.config, n_loc))
            if rec:
                traffic_bytes = rec["hbm_bytes_per_launch"]
                traffic = traffic_bytes / (k_ms * 1e-3) / 1e9
        out = {
            "metric": "loci/sec at fixed read-depth x rpb", "value": total_loci * a.steps / elapsed,
            "unit": "loci/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8/u32 scan + f64 posterior", "data": "synthetic",
            "config": {"workload": "%s: %d loci/GPU x %d reads (%d UMIs x %d rpb), seed %d"
                       % (cfg.name, n_loc, cfg.depth, cfg.n_umi, cfg.rpb, cfg.seed),
                       "loci_total": total_loci, "parallelism": "loci sharded x%d, %s" % (world, "rows gathered to rank 0" if (gather or world == 1) else
                                                                  "rows left in each rank's HBM (--rows resident)"),
                       "build_s": round(t_build, 1)},
            "roofline": {"bound": "hbm", "kernel": "k_call_loci", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel_ms": k_ms, "kernel_samples": k_n, "loci_per_launch": k_loci,
                         "alg_bytes_per_launch": alg_bytes,
                         "hbm_bytes_per_launch_pmc": traffic_bytes},
        }
        if cpu is not None:
            out["cpu_baseline"], out["cpu_baseline_c"] = cpu["python_pool"], cpu["c_port"]
            out["cpu_baseline_c_all_cores"] = cpu["c_port_all_cores"]
            gpu_rows = plan.download(rows)[:len(cpu["rows"])]
            bad = abi.compare_rows(gpu_rows, cpu["rows"], fragile=cpu["fragile"])
            out["parity_sample"] = {"loci": len(gpu_rows), "mismatches": len(bad), "detail": bad[:3]}
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def cpu_leg(a):
    """CPU baselines on a bounded sample of the same workload (first chunk of the config):
    * python_pool - oracle/vc_port.py, the pure-Python restatement of vc(), driven like the reference's
      main(): multiprocessing.Pool(all host cores), one task per locus, on the first 2000 loci;
    * c_port - oracle/smc_oracle.c on one core over the whole first chunk; its rows are kept to check
      the GPU rows of that chunk field by field after the timed run;
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
    ref_rows, fragile = oracle_lib.call_batch(sample, abi.c_params(params), abi.ROW_DTYPE, return_fragile=True)
    dt_c = time.perf_counter() - t
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    reps = 5
    t = time.perf_counter()
    for _ in range(reps):
        mt_rows = oracle_lib.call_batch_mt(sample, abi.c_params(params), abi.ROW_DTYPE, cores)
    dt_cmt = (time.perf_counter() - t) / reps
    assert mt_rows.tobytes() == ref_rows.tobytes()
    n_py = min(n, max(2000, 40 * cores))
    pool = vc_port.make_pool(cores)                 # started and warmed outside the timed region
    t = time.perf_counter()
    vc_port.call_batch(sample, params, n_cpu=cores, loci=range(n_py), pool=pool)
    dt_py = time.perf_counter() - t
    pool.close()
    pool.join()
    return {
        "python_pool": {"value": n_py / dt_py, "unit": "loci/s", "cores": cores, "kind": "port",
                        "sample": "first %d loci of the same workload, pure-Python port oracle/vc_port.py under "
                                  "multiprocessing.Pool(%d), one task per locus, %.1f s (pool already started)"
                                  % (n_py, cores, dt_py)},
        "c_port": {"value": n / dt_c, "unit": "loci/s", "cores": 1, "kind": "port",
                   "sample": "first %d loci, C restatement oracle/smc_oracle.c, 1 thread, %.1f s" % (n, dt_c)},
        "c_port_all_cores": {"value": n / dt_cmt, "unit": "loci/s", "cores": cores, "kind": "port",
                             "sample": "first %d loci, C restatement oracle/smc_oracle.c on %d threads (contiguous locus "
                                       "ranges), mean of %d passes, %.3f s each" % (n, cores, reps, dt_cmt)},
        "rows": ref_rows, "fragile": fragile}


if __name__ == "__main__":
    main()
