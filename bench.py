#!/usr/bin/env python3
"""bench.py - throughput of the per-locus hot path on synthetic pileups of stated depth.

Metric (BASELINE.json): loci/s at fixed read-depth x reads-per-UMI.  Workload at every N: BASELINE.json configs[2]
"synthetic 200k loci, 3000x depth, 50 UMIs/locus, 60 rpb" PER GPU (weak scaling: rank r takes its own stretch of the seeded
config), inputs resident in HBM before the timed region.

A STEP is the whole device path from where the reference's hot loop starts (smCounter.py:316, the pileup over the alignments):
the run's ALIGNMENTS (what the BAM decoder hands over: one record per alignment, CIGARs, bases, qualities; ~ 1.4 GB for C3) ->
smc_build_planes (the per-pileup-read work of vc(): sort, count, the walk that writes one word per read) -> smc_plan_create_dev ->
smc_plan_run_words (k_call_v2 bins + k_filter_loci) -> rows in HBM; for N > 1 followed by the gather of the packed rows to rank 0
(RCCL; the gather of a step overlaps the next step's kernels, two row buffers per rank - every step's rows are gathered inside the
timed region).  `--scaling strong --config C4`: BASELINE's configs[3], 1 M loci IN TOTAL sharded over the ranks.
(Rounds 1-3 quoted `value` on the second half of that path alone - the locus kernels over read words already resident; that
figure is kept as `consumer_only`.)

`python3 bench.py --gpus N` without a launcher starts its own ranks: a child `python -m torch.distributed.run --nnodes 1
--nproc-per-node N bench.py <same arguments>`, rank 0's JSON line relayed, the child's exit code returned.

The timed region is a block of EXACTLY --steps steps between barrier + synchronize on both sides; the block is repeated
--blocks times (default 5) and `value` / `ms_per_step` come from the MEDIAN block (all are listed under `blocks`).

Prints ONE JSON line on rank 0: value = loci of all ranks / max-over-ranks time; roofline = the bytes the step's dominant kernel
(the walk that writes the read words) has to move per launch over its mean HIP-event duration, against 8 TB/s; cpu_baseline =
CPU legs timed on a bounded sample of the same workload (rank 0, N = 1 only; inputs made before the clock starts); parity = EVERY
row of the run against the CPU (oracle/aln_planes.c + oracle/smc_oracle.c from the same alignments), with the number of loci
whose order-dependent fields were excused, the loci that reached filterVariants, the Fisher tests run and the largest p-value
difference; from_alignments = the SAME step on the other single-GPU shapes, variants under the reads where the config has them
(C5 = BASELINE's configs[4], X3 - 30 % of the loci with a candidate -, EX - the statistics of the reference's own example run -, C2),
every row checked the same way: where the p-value half of the metric is exercised; consumer_only = the locus kernels alone over
resident read words (C3) and other_configs = the other shapes that way.
The read words of a run live in a block the LIBRARY chose (engine.DevBuf(walk_output=True) -> smc_mem_alloc_best: step_breakdown
.allocation says what its probe saw); --place N > 0 adds round 4's bench-side trials for comparison.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# heavy imports happen inside main(): the CPU leg's spawned workers re-import this module
HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--blocks", type=int, default=25, help="timed blocks of EXACTLY --steps steps each (barrier + synchronize on both sides of "
                                                           "every block); value = the median block.  25 blocks of 20 steps are 1.3 s of device work: a median "
                                                           "over more blocks, and long enough for an outside observer of the GPU to see it busy")
    ap.add_argument("--place", type=int, default=0,
                    help="(round 4's bench-side trials, kept for comparison) extra allocations of the read words tried by the BENCH at "
                         "set-up, each timed with the real walk, the fastest kept.  Default 0: the library places the array itself - "
                         "engine.DevBuf(walk_output=True) -> smc_mem_alloc_best, a write-pattern probe over up to SMC_ALLOC_TRIES "
                         "(12) allocations, as every caller of the product path gets it (DESIGN.md section 5)")
    ap.add_argument("--slots", type=int, default=2,
                    help="sets of output arrays (each with a stream of its own) consecutive steps alternate between: with 2 the builder "
                         "of step i + 1 is enqueued while the locus kernels of step i run; 1: one step at a time")
    ap.add_argument("--config", default="C3", help="synthetic config (C2, C3, C5, C4); C3 is the metric's")
    ap.add_argument("--loci-per-gpu", type=int, default=0, help="override (default: the config's size)")
    ap.add_argument("--chunk", type=int, default=25000, help="loci generated / checked per chunk")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the every-row checks against the CPU")
    ap.add_argument("--no-other-configs", action="store_true", help="skip consumer_only and the C2 / C5 / X3 / EX one-liners")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak: the config's loci PER GPU (default, the driver's scaling run); strong: the config's loci in total, "
                         "sharded over the ranks (e.g. --config C4: 1 M loci; fits one MI355X too)")
    ap.add_argument("--rows", choices=("gather", "resident"), default="gather",
                    help="N > 1: gather every step's rows to rank 0 (default; overlapped with the next step) or leave "
                         "them in each rank's HBM (diagnostic: isolates the collective)")
    ap.add_argument("--wire", choices=("packed", "full"), default="packed",
                    help="N > 1: what travels to rank 0 - the packed wire rows (default) or the full 432-byte rows")
    return ap.parse_args(argv)


def self_launch(a) -> int:
    """--gpus N without WORLD_SIZE: start the ranks as a CHILD process (never re-exec a process that may touch the GPU), relay
    rank 0's JSON line, return the child's exit code."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for ln in proc.stdout:
        if ln.lstrip().startswith("{") and '"metric"' in ln:
            line = ln.strip()
        else:
            sys.stderr.write(ln)
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    elif rc == 0:
        rc = 1
    return rc


def needed_bytes(loci) -> float:
    """HBM bytes one launch of the locus kernels has to move for these loci: the read words (one uint32 per read slot: allele,
    quality, fragment start, class), umi_start, the 32-byte descriptor + 4-byte launch-order entry, and the row written."""
    import numpy as np
    from smcounter_amd import abi
    slots = ((loci["n_reads"].astype(np.int64) + 3) // 4 * 4).sum()
    return float(4 * slots + 4 * (loci["n_umi"].astype(np.int64) + 1).sum() + (32 + 4 + abi.ROW_DTYPE.itemsize) * len(loci))


class Resident(object):
    """One config's batch of READ WORDS resident in HBM (packed on the device from the synthetic batch's meta and frag planes
    before anything is timed) and umi_start, + the plan: the input of the locus kernels alone (`consumer_only`,
    `other_configs`); optionally the CPU restatement's rows of every chunk (all host cores), kept for the parity pass."""

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


def call_roofline(plan_loci, k_ms, k_n, k_loci, k_reads, cfg_key):
    """k_call_v2 alone: the bytes it HAS to move per launch (the read words - 4 B per read slot -, umi_start, descriptors, rows)
    over its mean HIP-event duration, against the 8 TB/s peak.  (SURVEY.md 8d's figure - 16 B per read + 360 B per locus - counts
    the four raw-field planes the plane builder digests: it is charged to the WHOLE step, `whole_step_on_survey_8d`, not to a kernel
    that moves a quarter of it.)"""
    import bench_fa
    need = needed_bytes(plan_loci)
    achieved = need / (k_ms * 1e-3) / 1e9
    rec, why = bench_fa.traffic_record(cfg_key)
    return {"kernel": "k_call_v2", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "frac_basis": "needed bytes per launch (4 B per read slot: the read word; + umi_start + descriptor + row) / kernel time / 8 TB/s",
            "traffic": rec["hbm_bytes_per_launch"] / (k_ms * 1e-3) / 1e9 if rec else None, "traffic_measured_in_run": False,
            **({"traffic_note": why} if why else {}),
            "kernel_ms": k_ms, "kernel_samples": k_n, "loci_per_launch": k_loci, "needed_bytes_per_launch": need}


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


def dry_main(a, rank, world):
    """SMC_BENCH_DRY=1: the launch plumbing without a GPU (CPU test of `--gpus N`): the ranks meet over gloo, run the timed
    loop's barriers and the row gather on dummy rows, rank 0 prints the line (value null)."""
    import torch
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group("gloo")
    rows = torch.full((64, 42), float(rank))
    t0 = time.perf_counter()
    for _ in range(a.steps):
        if world > 1:
            got = [torch.empty_like(rows) for _ in range(world)] if rank == 0 else None
            dist.gather(rows, got, dst=0)
            if rank == 0:
                assert all(float(g[0, 0]) == r for r, g in enumerate(got))
    if world > 1:
        dist.barrier()
    el = time.perf_counter() - t0
    if rank == 0:
        emit({"metric": "loci/sec at fixed read-depth x rpb", "value": None, "unit": "loci/s", "n_gpus": world,
              "steps": a.steps, "warmup": a.warmup, "ms_per_step": el / max(1, a.steps) * 1e3, "higher_is_better": True,
              "scaling": a.scaling, "vs_baseline": None, "dtype": "u8/u32 scan + f64 posterior", "data": "synthetic",
              "config": {"workload": "dry run: launch plumbing only (SMC_BENCH_DRY=1), no GPU work"}, "dry_run": True})
    if world > 1:
        dist.destroy_process_group()


def main():
    global np, torch, dist, abi, engine, synth, smcdist
    a = parse_args()
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        raise SystemExit(self_launch(a))               # (before anything here imports torch or touches a GPU)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (a.gpus, world))
    if os.environ.get("SMC_BENCH_DRY"):
        return dry_main(a, rank, world)
    import numpy as np
    from smcounter_amd import abi, synth
    # The CPU leg runs FIRST, before this process touches the GPU: it starts worker processes.
    cpu = None
    if world == 1 and not a.no_cpu_baseline:
        cpu = cpu_leg(a)
    import torch
    import torch.distributed as dist
    import bench_fa
    from smcounter_amd import _lib, engine
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
    L = eng.L
    dev = torch.device("cuda", local_rank)
    nthreads = max(1, len(os.sched_getaffinity(0)) // max(1, world))

    # ---- the rank's run of alignments, resident in HBM
    t0 = time.time()
    # (two sets of output arrays, each with a stream of its own: the builder of step i + 1 is enqueued while the locus kernels of
    # step i run - as the runs of a BAM follow each other; --slots 1: one step at a time on one stream)
    run = bench_fa.AlignmentRun(eng, cfg, params, n_mine, nthreads, shard=rank, slots=a.slots, place=a.place)
    torch.cuda.synchronize()
    t_build = time.time() - t0
    rows = torch.empty(n_mine * abi.ROW_DTYPE.itemsize, dtype=torch.uint8, device=dev)
    # (buffer k of the pipeline is filled by step(slot=k), which runs on slot k % active_slots' stream - the placement trials may
    # have left ONE active slot: the pipeline must order its waits and gathers against the stream that really fills the buffer,
    # ADVICE r4)
    slot_streams = [run.stream_of(k % run.active_slots) for k in range(a.slots)] if a.slots > 1 else None

    # N > 1: the rows of step i travel to rank 0 (RCCL, its own stream) while step i + 1 computes - two buffers per rank,
    # a buffer is reused only after its gather has completed (dist.RowPipeline).  What travels is the packed wire row
    # (smc_pack_rows: every printed column, 168 of the 432 bytes) unless --wire full.
    gather = use_dist and a.rows == "gather"
    packed = gather and a.wire == "packed"
    wire_bytes = L.smc_wire_row_size()
    if packed:
        wires = [torch.empty(n_mine * wire_bytes, dtype=torch.uint8, device=dev) for _ in range(max(2, a.slots))]

        def produce(buf, b=None):
            import ctypes as _c
            run.step(slot=b)                                         # (rows into the slot's own row buffer, on the slot's stream)
            S = run.slots[run.last_slot]
            sp = _c.c_void_p(S["stream"].cuda_stream) if S["stream"] is not None else None
            _lib.check(L.smc_pack_rows(eng.ctx, S["rows"].data_ptr(), n_mine, buf.data_ptr(), sp), "smc_pack_rows")
        pipe = smcdist.RowPipeline(wires, collective=True, streams=slot_streams)
    else:
        bufs = [rows] + [torch.empty_like(rows) for _ in range(max(2 if gather else 1, a.slots) - 1)]

        def produce(buf, b=None):
            run.step(rows=buf, slot=b)
        pipe = smcdist.RowPipeline(bufs, collective=gather, streams=slot_streams)

    def step():
        pipe.step(produce)

    for _ in range(max(1, a.warmup)):
        step()
    pipe.drain()
    torch.cuda.synchronize()
    status = run.status()
    for k in run.t:
        run.t[k] = 0
    blocks = []
    _lib.check(L.smc_build_set_timing(eng.ctx, min(a.steps * a.blocks, 256)), "smc_build_set_timing")
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
    import ctypes
    k_ms, k_n = ctypes.c_float(), ctypes.c_int32()
    _lib.check(L.smc_build_kernel_ms(eng.ctx, ctypes.byref(k_ms), ctypes.byref(k_n)), "smc_build_kernel_ms")
    L.smc_build_set_timing(eng.ctx, 0)
    elapsed = sorted(blocks)[len(blocks) // 2]                   # the median block
    # what rank 0 received in the last step against what every rank sent (outside the timed region): a 64-bit sum over each
    # rank's last buffer, compared with the sum over the block rank 0 holds for that rank
    gather_check = None
    if gather:
        b_last = (pipe.n - 1) % len(pipe.bufs)
        csum = lambda t: t.view(torch.int32).sum(dtype=torch.int64).reshape(1)
        mine = csum(pipe.bufs[b_last])
        mine = mine.cpu() if share_gpu else mine
        sums = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
        dist.all_gather(sums, mine)
        if rank == 0:
            got_sums = [int(csum(t).item()) for t in pipe.recv[b_last]]
            sent = [int(t.item()) for t in sums]
            gather_check = {"ranks": len(sent), "blocks_equal_what_was_sent": got_sums == sent, "bytes_per_rank": int(pipe.bufs[b_last].numel())}
            if got_sums != sent:
                raise SystemExit("bench.py: the rows rank 0 gathered differ from what the ranks sent: %r vs %r" % (got_sums, sent))
    # the locus kernels of the same planes, timed alone with one more plan (outside the timed region)
    if slot_streams is not None:
        torch.cuda.set_stream(torch.cuda.default_stream(dev))
    # one step at a time (every step on slot 0's stream): what a step takes when nothing of the next one runs beside it
    torch.cuda.synchronize()
    serial_ms = run.serial_ms(a.steps, rows=rows)
    # the same run with the read words in the other width (32 bits), the same way: what the 16-bit word bought on this box
    other_width = None
    if world == 1 and not a.no_other_configs and not os.environ.get("SMC_BENCH_NO_WIDTH_AB"):
        try:
            other_width = run.measure_other_word_width(a.steps, rows=rows)
        except Exception as e:                                       # (memory short, ...: the headline does not depend on it)
            other_width = {"error": str(e)}
    plan = run.step(keep_plan=True, rows=rows, slot=0)
    torch.cuda.synchronize()
    plan.set_timing(8)
    for _ in range(8):
        plan.run([run.words, run.uaux[0]], params, rows, stream=0)
    c_ms = plan.kernel_ms()[0]
    plan.close()
    torch.cuda.synchronize()

    out = None
    if rank == 0:
        n_steps = max(1, run.t["n"])
        out = {
            "metric": "loci/sec at fixed read-depth x rpb", "value": total_loci * a.steps / elapsed,
            "unit": "loci/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True, "scaling": a.scaling,
            "vs_baseline": None, "dtype": "u8/u32 scan + f64 posterior", "data": "synthetic",
            "config": {"workload": "%s: %d loci%s x %d reads (%d UMIs x %d rpb), seed %d; as ALIGNMENTS resident in HBM: %s"
                       % (cfg.name, n_loc, "/GPU" if a.scaling == "weak" else " in total", cfg.depth, cfg.n_umi, cfg.rpb, cfg.seed,
                          bench_fa.describe(run, a.config)),
                       "step": ("smc_build_planes_w16 (sort, count, the walk that writes one 16-bit word per pileup read) -> smc_plan_create_dev_spec -> "
                                "smc_plan_run_words16 (k_call_v2 + k_filter_loci) -> rows in HBM" if run.word_bits == 16 else
                                "smc_build_planes (sort, count, the walk that writes one word per pileup read) -> smc_plan_create_dev_spec -> "
                                "smc_plan_run_words (k_call_v2 + k_filter_loci) -> rows in HBM") + ((" -> %s rows gathered to rank 0"
                               % ("packed wire" if packed else "full")) if gather else ""),
                       "loci_total": total_loci, "read_word_bits": run.word_bits,
                       "parallelism": "loci sharded x%d, %s" % (world, ("%s rows gathered to rank 0" % ("packed wire" if packed else "full"))
                                                                 if gather else ("single GPU" if world == 1 else
                                                                                 "rows left in each rank's HBM (--rows resident)")),
                       "build_s": round(t_build, 1), **({"note": "SMC_BENCH_SHARE_GPU: all ranks on one GPU through gloo - a functional "
                                                                         "check of the N > 1 path, not a measurement"} if share_gpu else {})},
            "blocks": {"n": len(blocks), "steps_each": a.steps, "ms_per_step": [round(b / a.steps * 1e3, 4) for b in blocks],
                       "value_from": "median block", "spread_pct": round(100.0 * (max(blocks) - min(blocks)) / elapsed, 2)},
            "roofline": dict(bench_fa.roofline_block(run, k_ms.value, k_n.value, a.config), **{
                # SURVEY.md 8d's per-locus figure (16 B per pileup read + 360 B per locus) charged to the WHOLE step - what VERDICT r3
                # holds against the north star's "40 % of the HBM roofline" (this rank's share of the step)
                "whole_step_on_survey_8d": {
                    "bytes_per_step": 16.0 * run.reads + 360.0 * run.nl,
                    "achieved": (16.0 * run.reads + 360.0 * run.nl) / (elapsed / a.steps) / 1e9, "unit": "GB/s",
                    "frac": (16.0 * run.reads + 360.0 * run.nl) / (elapsed / a.steps) / 1e9 / HBM_PEAK_GBS},
                # the step one at a time with the read words in the other width, same process, same inputs, right after the headline
                **({"other_read_word_width_same_run": dict(other_width, ms_per_step_one_at_a_time_this_width=serial_ms)} if other_width else {})}),
            "step_breakdown": {"slots": a.slots, "slot_choice": run.slot_choice, "ms_per_step_one_at_a_time": serial_ms, "placement": run.placement,
                               # what the library's write-pattern probe saw when it chose the blocks of the read words (one per slot)
                               "allocation": {"chosen_by": "smc_mem_alloc_best (engine.DevBuf(walk_output=True))", "tries": eng.alloc_tries,
                                              "blocks": list(eng.alloc_log)},
                               "k_bp_emit2_ms": k_ms.value, "k_call_v2_ms": c_ms,
                               "host_ms_per_step": {k: round(v / n_steps * 1e3, 3) for k, v in run.t.items() if k != "n"},
                               "pileup_reads_per_s": run.reads * a.steps / elapsed, "builder_status": status,
                               "plans": bench_fa.plans_record(eng)},
            "p_value_note": "C3 has no locus that reaches filterVariants (parity.loci_filtered 0): the p-value half of the metric is "
                            "carried by from_alignments X3 / EX / C5 (the same step as the headline, variants under the reads) and by "
                            "other_configs X3 / EX / C5 (read words resident)",
            "host_buffers": "inputs resident in HBM when the timed region starts; handing host buffers of planes instead "
                            "(smc_call_batch_host: H2D of 8 B/read + kernels + D2H of the rows) measured 1.94 M loci/s on C3 "
                            "(DESIGN.md section 5) - PCIe-bound, never `value`",
        }
        if gather_check is not None:
            out["gather_check"] = gather_check
        if cpu is not None:
            out["cpu_baseline"], out["cpu_baseline_c"] = cpu["python_pool"], cpu["c_port"]
            out["cpu_baseline_c_all_cores"] = cpu["c_port_all_cores"]
            out["cpu_baseline_c_from_alignments"] = cpu["c_from_alignments"]
            out["cpu_baseline_object_adapter"] = cpu["object_adapter"]
            out["cpu_baseline_single_process"] = cpu["python_single"]
            out["cpu_baseline_pool_chunked"] = cpu["python_pool_chunked"]
        if world == 1 and not a.no_parity:
            run.step(rows=rows, slot=0)
            torch.cuda.synchronize()
            run.rows = _TensorBuf(rows)
            out["parity"] = bench_fa.parity_full(run, nthreads, chunk=a.chunk)
    run.close()
    del rows, pipe
    if rank == 0 and world == 1 and not a.no_other_configs:
        oracle = None
        if not a.no_parity:
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import oracle_lib as oracle
        torch.cuda.empty_cache()
        out["consumer_only"] = other_config(eng, "C3", a, dev, nthreads, oracle)
        out["consumer_only"]["what"] = ("the locus kernels alone (k_call_v2 bins + k_filter_loci) over read words already resident - the "
                                        "second half of the step; rounds 1-3 quoted this as `value`")
        out["other_configs"] = {}
        for name in ("C2", "C5", "X3", "EX"):           # X3: 30 % of the loci reach the filters; EX: the example run's statistics
            out["other_configs"][name] = other_config(eng, name, a, dev, nthreads, oracle)
        # the same shapes through the WHOLE device path, as the headline goes: alignments (variants under the reads where the
        # config has them: smc_synth_alignments' alt_locus_frac / alt_af) -> smc_build_planes -> smc_plan_create_dev ->
        # smc_plan_run_words, every row against oracle/aln_planes.c + oracle/smc_oracle.c - BASELINE's configs[4] (C5), loci that
        # reach filterVariants and its Fisher tests (X3, EX), the reference's own depth (EX) and the small batch (C2)
        out["from_alignments"] = {}
        for name in ("C5", "X3", "EX", "C2"):
            torch.cuda.empty_cache()
            eng.trim()
            o = bench_fa.run_leg(eng, name, synth.CONFIGS[name].n_loci, a.steps, a.warmup, a.blocks, nthreads,
                                 parity_loci=0 if a.no_parity else -1, slots=a.slots, place=0)
            for k in ("placement", "generate_s"):
                o.pop(k, None)
            out["from_alignments"][name] = o
    if rank == 0:
        emit(out)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


LINE_LIMIT = 4096            # the driver reads a bounded tail of stdout: a line beyond it is not a record (BENCH_r05: parsed null)


def _r(x, sig=6):
    """Numbers of the headline line at `sig` significant digits (the sidecar keeps them in full)."""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    if isinstance(x, float):
        return float("%.*g" % (sig, x)) if x == x and abs(x) != float("inf") else None
    if isinstance(x, dict):
        return {k: _r(v, sig) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_r(v, sig) for v in x]
    return x


def _pick(d, keys):
    return {k: d[k] for k in keys if d is not None and k in d}


def compact(out: dict) -> dict:
    """The ONE line the driver records: the contract's keys, `roofline` and `cpu_baseline`, the parity counts and one short
    list per other leg - everything else of `out` (the prose, the allocation log, the extra CPU legs, the per-leg rooflines)
    goes to the sidecar `bench_detail.json` and to stderr."""
    cfgk = out.get("config", {})
    line = _pick(out, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                       "vs_baseline", "dtype", "data"))
    line["config"] = _pick(cfgk, ("workload", "step", "loci_total", "read_word_bits", "parallelism"))
    for k in ("workload", "step"):                       # (the sidecar has the long form)
        if k in line["config"] and len(line["config"][k]) > 260:
            line["config"][k] = line["config"][k][:257] + "..."
    if "blocks" in out:
        b = sorted(out["blocks"]["ms_per_step"])                 # (the sidecar lists every block)
        line["blocks_ms_per_step"] = {"n": len(b), "min": b[0], "median": b[len(b) // 2], "max": b[-1]}
    rf = out.get("roofline")
    if rf is not None:
        line["roofline"] = _pick(rf, ("bound", "kernel", "kernel_ms", "kernel_samples", "needed_bytes_per_launch", "achieved", "peak",
                                      "unit", "frac", "traffic", "hbm_bytes_per_launch_pmc"))
        line["roofline"]["kernel"] = str(line["roofline"].get("kernel", "")).split(" (")[0]
        if "whole_step_on_survey_8d" in rf:
            line["roofline"]["whole_step_on_survey_8d"] = _pick(rf["whole_step_on_survey_8d"], ("bytes_per_step", "frac"))
    sb = out.get("step_breakdown")
    if sb is not None:
        line["step_breakdown"] = _pick(sb, ("ms_per_step_one_at_a_time", "k_bp_emit2_ms", "k_call_v2_ms", "host_ms_per_step"))
    cb = out.get("cpu_baseline")
    if cb is not None:
        line["cpu_baseline"] = _pick(cb, ("value", "unit", "cores", "kind"))
        line["cpu_baseline"]["sample"] = cb.get("sample_short") or str(cb.get("sample", ""))[:160]
    for k in ("cpu_baseline_c_from_alignments", "cpu_baseline_object_adapter"):
        if out.get(k) is not None:
            line[k] = _pick(out[k], ("value", "cores"))
    if out.get("parity") is not None:
        line["parity"] = _pick(out["parity"], ("loci", "mismatches", "pi_max_abs_diff", "p_max_abs_diff", "near_tie_skipped",
                                               "fragile_skipped", "loci_filtered", "fisher_tests_run"))
    if out.get("gather_check") is not None:
        line["gather_check"] = out["gather_check"]

    def leg(o):          # [loci/s, ms per step, mismatches, Fisher tests run, fraction of the roofline (leg's own basis)]
        par = o.get("parity") or {}
        frac = (o.get("whole_step_on_survey_8d") or {}).get("frac")
        if frac is None:
            frac = (o.get("roofline") or {}).get("frac")
        return [o.get("value"), o.get("ms_per_step"), par.get("mismatches"), par.get("fisher_tests_run"), frac]
    if out.get("from_alignments"):
        line["from_alignments"] = {k: leg(v) for k, v in out["from_alignments"].items()}
        line["legs_are"] = "[loci/s, ms_per_step, parity mismatches, Fisher tests run, roofline frac]"
    if out.get("consumer_only"):
        line["consumer_only"] = leg(out["consumer_only"])
    if out.get("other_configs"):
        line["other_configs"] = {k: leg(v) for k, v in out["other_configs"].items()}
    if out.get("dry_run"):
        line["dry_run"] = True
    line["detail"] = "bench_detail.json"
    return _r(line)


def emit(out: dict):
    """Everything to the sidecar (and to stderr), the compact line LAST on stdout."""
    text = json.dumps(out)
    for path in (os.path.join(ROOT, "bench_detail.json"), os.path.join(ROOT, "gpurun_out", "bench_detail.json")):
        try:
            if os.path.isdir(os.path.dirname(path)):
                with open(path, "w") as fh:
                    fh.write(text + "\n")
        except OSError as e:                               # (a read-only checkout: the line is still printed)
            sys.stderr.write("bench.py: %s not written: %s\n" % (path, e))
    sys.stderr.write(text + "\n")
    sys.stderr.flush()
    line = json.dumps(compact(out), separators=(",", ":"))
    if len(line) > LINE_LIMIT:                             # never print a line the driver cannot record: drop the optional legs
        c = compact(out)
        for k in ("other_configs", "consumer_only", "step_breakdown", "blocks_ms_per_step", "from_alignments"):
            c.pop(k, None)
            line = json.dumps(c, separators=(",", ":"))
            if len(line) <= LINE_LIMIT:
                break
    assert len(line) <= LINE_LIMIT, len(line)
    sys.stdout.flush()
    print(line, flush=True)


class _TensorBuf(object):
    """A torch tensor where bench_fa expects a DevBuf (download / free)."""

    def __init__(self, t):
        self.t = t

    def download(self, dtype, count, offset_bytes=0):
        import numpy as np
        return self.t.cpu().numpy()[offset_bytes:offset_bytes + count * np.dtype(dtype).itemsize].view(dtype).copy()

    def data_ptr(self):
        return self.t.data_ptr()

    def free(self):
        self.t = None


def other_config(eng, name, a, dev, nthreads, oracle):
    """The locus kernels alone on a single-GPU shape (resident read words, warm-up, blocks of --steps steps, median block,
    HIP-event time of the dominant kernel, every row checked), reported as one short object."""
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
    o = {"workload": "%s: %d loci x %d reads (%d UMIs x %d rpb), seed %d; read words resident in HBM"
                     % (name, cfg.n_loci, cfg.depth, cfg.n_umi, cfg.rpb, cfg.seed),
         "value": cfg.n_loci * a.steps / el, "unit": "loci/s", "ms_per_step": el / a.steps * 1e3,
         "blocks_ms_per_step": [round(b / a.steps * 1e3, 4) for b in blocks],
         "roofline": call_roofline(res.loci, k_ms, k_n, k_loci, k_reads, "%s:%d" % (name, cfg.n_loci))}
    if oracle is not None:
        res.run(rows)
        torch.cuda.synchronize()
        o["parity"] = res.parity(rows)
    res.close()
    return o


def cpu_leg(a):
    """CPU baselines on a bounded sample of the same workload (first chunk of the config), every input made BEFORE its clock
    starts:
    * python_pool - oracle/vc_port.py, the pure-Python restatement of vc(), driven like the reference's main():
      multiprocessing.Pool(all host cores), one task per locus (smCounter.py:683-685) - the loci's pileups generated into
      each worker before the timed pass (the reference's worker reads its own BAM region: that read is not what is compared);
    * python_single - the same port in this process, no pool; `ideal_all_cores` = that rate x physical cores;
    * c_port / c_port_all_cores - oracle/smc_oracle.c on one core / on every host core."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_lib
    import vc_port
    from smcounter_amd import abi, synth
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
    # the C restatement FROM ALIGNMENTS - the work the timed GPU step does: oracle/aln_planes.c (pileup, allele,
    # barcode / fragment bookkeeping, smCounter.py:316-471) + oracle/smc_oracle.c on every host core, alignments made before the clock
    A = synth.generate_alignments(cfg, n, params, nthreads=min(48, cores))
    t = time.perf_counter()
    db_a = oracle_lib.aln_planes(A, params, 0, n, n_threads=cores)
    dt_planes = time.perf_counter() - t
    oracle_lib.call_batch_mt(db_a, abi.c_params(params), abi.ROW_DTYPE, cores)
    dt_fa = time.perf_counter() - t
    del A, db_a
    # one process, no pool: what one core does with the pure-Python restatement (no task pickling, no result pipes)
    n_one = min(n, 120)
    t = time.perf_counter()
    vc_port.call_batch(sample, params, n_cpu=1, loci=range(n_one))
    dt_one = time.perf_counter() - t
    n_py = min(n, max(2000, 40 * cores))
    pool = vc_port.make_pool(cores)                 # started and warmed outside the timed region
    shared = vc_port.share_batch(sample, n_py)      # the loci's pileups, made before the clock starts, mapped by every worker
    vc_port.call_shared(shared, params, range(min(n_py, 4 * cores)), pool)   # (every worker has imported and mapped)
    t = time.perf_counter()
    got = vc_port.call_shared(shared, params, range(n_py), pool)
    dt_py = time.perf_counter() - t
    vc_port.unshare_batch(shared)
    # the port fed the way the REFERENCE's worker is fed (SURVEY.md 8d "through an object adapter"): pysam-like objects per pileup read,
    # the read name split and joined, the tag list walked for NM, the CIGAR walked, string allele keys (oracle/vc_port_objects.py,
    # smCounter.py:316-479) - every worker maps the run's alignments, makes the objects of its own loci (not timed: the reference's
    # worker reads them from its BAM) and times its pass; the rate is all loci over the slowest worker's pass
    import vc_port_objects
    per_w = 6
    A_o = synth.generate_alignments(cfg, min(n, cores * per_w), params, nthreads=min(48, cores))
    ref_o = synth.aln_ref_fetch(int(A_o["start0"]), int(A_o["start0"]) + int(A_o["nl"]) + 256)
    vc_port_objects.timed_pass(A_o, params, ref_o, min(cores, 8), 1, pool)      # (imports in the workers)
    n_one_o, dt_one_o, _, _ = vc_port_objects.timed_pass(A_o, params, ref_o, 1, per_w, pool)   # one worker alone: what a core does with it
    n_obj, dt_obj, w_obj, _ = vc_port_objects.timed_pass(A_o, params, ref_o, cores, per_w, pool)
    del A_o
    vc_port.call_config(a.config, params, range(cores), pool)          # (first import of the generator in every worker)
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
                        "sample_short": "first %d loci of the workload, oracle/vc_port.py under multiprocessing.Pool(%d), one task per locus "
                                        "(smCounter.py:683-685), %.1f s" % (n_py, cores, dt_py),
                        "per_physical_core": per_core_pool,
                        "ideal_all_cores": n_one / dt_one * phys,
                        "sample": "first %d loci of the same workload, pure-Python port oracle/vc_port.py under "
                                  "multiprocessing.Pool(%d), one task per locus as smCounter.py:683-685, %s, %.1f s (pool already "
                                  "started); ideal_all_cores = the single-process rate x %d physical cores"
                                  % (n_py, cores, "the loci's pileups made BEFORE the clock starts and mapped by the workers", dt_py, phys)},
        "object_adapter": {"value": n_obj / dt_obj, "unit": "loci/s", "cores": phys, "logical_cpus": cores, "kind": "port",
                           "per_physical_core": n_obj / dt_obj / phys, "one_worker_alone": n_one_o / dt_one_o,
                           "sample": "first %d loci AS pysam-like objects (oracle/vc_port_objects.py: qname split, NM tag, CIGAR scan, string "
                                     "allele keys - smCounter.py:316-479 - then the port's :482-600), %d workers x %d loci each, objects made "
                                     "before each worker's clock starts; rate = loci / the slowest worker's pass (%.2f s); one worker alone: %.1f loci/s"
                                     % (n_obj, w_obj, per_w, dt_obj, n_one_o / dt_one_o)},
        "python_pool_chunked": {"value": n_ch / dt_ch, "unit": "loci/s", "cores": phys, "logical_cpus": cores, "kind": "port",
                                "sample": "first %d loci, the same pool with 20 consecutive loci per task, every worker generating its own "
                                          "loci inside the clock (rounds 2-3's variant), %.1f s" % (n_ch, dt_ch)},
        "python_single": {"value": n_one / dt_one, "unit": "loci/s", "cores": 1, "kind": "port",
                          "sample": "first %d loci, oracle/vc_port.py in this process (no pool), %.1f s; SURVEY.md section 6 timed the "
                                    "imported reference at ~51 loci/s on one core of the build container for this shape"
                                    % (n_one, dt_one)},
        "c_port": {"value": n / dt_c, "unit": "loci/s", "cores": 1, "kind": "port",
                   "sample": "first %d loci, C restatement oracle/smc_oracle.c, 1 thread, %.1f s" % (n, dt_c)},
        "c_from_alignments": {"value": n / dt_fa, "unit": "loci/s", "cores": phys, "logical_cpus": cores, "kind": "port",
                              "sample": "first %d loci AS ALIGNMENTS (what the timed GPU step starts from): oracle/aln_planes.c %.2f s + "
                                        "oracle/smc_oracle.c %.2f s on %d threads" % (n, dt_planes, dt_fa - dt_planes, cores)},
        "c_port_all_cores": {"value": n / dt_cmt, "unit": "loci/s", "cores": phys, "logical_cpus": cores, "kind": "port",
                             "sample": "first %d loci, C restatement oracle/smc_oracle.c on %d threads (contiguous locus "
                                       "ranges), mean of %d passes, %.3f s each" % (n, cores, reps, dt_cmt)}}


if __name__ == "__main__":
    main()
