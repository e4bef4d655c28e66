"""`python3 bench.py --gpus N` without a launcher (the shape of the driver's command) starts its own ranks as a child
`torch.distributed.run` and relays rank 0's line.  Here without a GPU: SMC_BENCH_DRY=1 keeps the launch, the rendezvous, the
barriers and a gather of dummy rows over gloo, and skips the device work."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT


@pytest.mark.timeout(300)
@pytest.mark.parametrize("n", [2, 4, 8])
def test_bench_starts_its_own_ranks(n):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["SMC_BENCH_DRY"] = "1"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "3", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=280)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == n and d["steps"] == 3 and d["metric"].startswith("loci/sec") and d["dry_run"] is True
    # the line is the LAST thing on stdout and short enough for the driver's bounded tail (BENCH_r05: a 21 KB line was not recorded)
    assert p.stdout.rstrip().splitlines()[-1] == lines[0] and len(lines[0]) <= 4096


def _fat_record():
    """A record shaped like bench.py's full result with every optional block present and long prose in it."""
    prose = "x" * 700
    rf = {"bound": "hbm", "kernel": "k_bp_emit2 (the walk that writes the read words: the step's dominant kernel)", "kernel_ms": 1.0577949285507202,
          "kernel_samples": 100, "needed_bytes_per_launch": 2751318844.0, "achieved": 2600.994549831666, "peak": 8000.0, "unit": "GB/s",
          "frac": 0.32512431872895825, "frac_basis": prose, "traffic": 3072.123456789, "traffic_note": prose, "hbm_bytes_per_launch_pmc": 3.25e9,
          "whole_step_on_survey_8d": {"bytes_per_step": 9652571168.0, "achieved": 3686.9, "unit": "GB/s", "frac": 0.46086858351215537},
          "other_read_word_width_same_run": {"note": prose}}
    par = {"loci": 200000, "mismatches": 0, "fragile_skipped": 0, "near_tie_skipped": 54, "underflow_skipped": 0, "pi_max_abs_diff": 1.3380372365645599e-09,
           "loci_filtered": 0, "fisher_tests_run": 0, "p_max_abs_diff": 0.0, "detail": [prose] * 3, "checked_against": prose}
    leg = {"workload": prose, "step": prose, "value": 28715612.123, "unit": "loci/s", "ms_per_step": 3.4824312, "roofline": dict(rf), "parity": dict(par),
           "whole_step_on_survey_8d": {"bytes_per_step": 1.2e10, "frac": 0.4586}, "host_ms_per_step": {"plan_create": 0.1}}
    cpu = {"value": 3209.07123, "unit": "loci/s", "cores": 128, "logical_cpus": 256, "kind": "port", "sample": prose}
    return {"metric": "loci/sec at fixed read-depth x rpb", "value": 76393100.123456, "unit": "loci/s", "n_gpus": 1, "steps": 20, "warmup": 5,
            "ms_per_step": 2.6180412345, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8/u32 scan + f64 posterior",
            "data": "synthetic", "config": {"workload": prose, "step": prose, "loci_total": 200000, "read_word_bits": 16, "parallelism": "loci sharded x1, single GPU"},
            "blocks": {"n": 5, "ms_per_step": [2.6294, 2.6173, 2.618, 2.6165, 2.6198]}, "roofline": rf,
            "step_breakdown": {"ms_per_step_one_at_a_time": 2.65, "k_bp_emit2_ms": 1.05, "k_call_v2_ms": 0.85, "allocation": {"blocks": [prose] * 12},
                               "host_ms_per_step": {"build_issue": 0.108, "descriptors_d2h": 0.0, "plan_create": 2.476, "run_issue": 0.013}},
            "p_value_note": prose, "host_buffers": prose, "cpu_baseline": cpu, "cpu_baseline_c": cpu, "cpu_baseline_c_all_cores": cpu,
            "cpu_baseline_c_from_alignments": cpu, "cpu_baseline_object_adapter": cpu, "cpu_baseline_single_process": cpu, "parity": par,
            "consumer_only": leg, "other_configs": {k: leg for k in ("C2", "C5", "X3", "EX")},
            "from_alignments": {k: leg for k in ("C5", "X3", "EX", "C2")}}


def test_the_headline_line_is_short_and_round_trips(tmp_path, capsys, monkeypatch):
    """What `bench.py` prints last on stdout is ONE JSON object of at most 4 KB carrying the contract's keys, `roofline` and
    `cpu_baseline`; the full record goes to the sidecar and to stderr."""
    import bench
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    rec = _fat_record()
    assert len(json.dumps(rec)) > 30000
    bench.emit(rec)
    cap = capsys.readouterr()
    line = cap.out.rstrip().splitlines()[-1]
    assert len(line) <= bench.LINE_LIMIT == 4096
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline", "parity"):
        assert k in d, k
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(d["roofline"]) and d["roofline"]["kernel"] == "k_bp_emit2"
    assert abs(d["roofline"]["frac"] - d["roofline"]["achieved"] / d["roofline"]["peak"]) < 1e-5
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(d["cpu_baseline"])
    assert d["config"]["workload"] and "model" not in d["config"]
    assert d["from_alignments"]["C5"][:4] == [28715600.0, 3.48243, 0, 0]
    assert abs(d["value"] - rec["value"]) / rec["value"] < 1e-5
    full = json.loads((tmp_path / "bench_detail.json").read_text())
    assert full == json.loads(json.dumps(rec)) and json.loads(cap.err.strip().splitlines()[-1]) == full


def test_bench_refuses_a_world_that_does_not_match():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", SMC_BENCH_DRY="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "WORLD_SIZE=3" in (p.stderr + p.stdout)


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_rccl_runs_the_gather_path_with_one_rank():
    """RCCL on hardware: `bench.py --gpus 1` with SMC_BENCH_FORCE_DIST=1 under a CHILD
    `torch.distributed.run --nproc-per-node 1` on the `nccl` backend - init, the packed wire rows gathered to rank 0 inside every
    step, barrier, all_reduce of the block time, all_gather of the checksums - and the JSON line it prints: the rows rank 0
    received are the rows that were sent, and every row of the run equals the oracle's."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(SMC_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--blocks", "2",
           "--loci-per-gpu", "20000", "--place", "0", "--no-cpu-baseline", "--no-other-configs"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=850)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-3000:])
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["value"] > 0
    assert len(lines[0]) <= 4096 and p.stdout.rstrip().splitlines()[-1] == lines[0]
    assert "gathered to rank 0" in d["config"]["parallelism"]
    assert d["gather_check"] == {"ranks": 1, "blocks_equal_what_was_sent": True, "bytes_per_rank": 20000 * 168}
    assert d["parity"]["loci"] == 20000 and d["parity"]["mismatches"] == 0


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_two_ranks_on_one_gpu_gather_what_they_sent_when_one_rank_uses_one_slot():
    """The N > 1 bench path end to end on the one GPU of the test box (SMC_BENCH_SHARE_GPU: both ranks on GPU 0, gloo between them -
    RCCL refuses two ranks on one device): every step's packed rows gathered to rank 0 through dist.RowPipeline while the next step
    computes.  Rank 1 is made to step through ONE slot (its two row buffers are then filled by the same stream) while rank 0
    alternates between two - the configuration in which the gather once raced with the stream that fills the buffer;
    the bench compares what rank 0 received with what every rank sent."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(SMC_BENCH_SHARE_GPU="1", SMC_FA_ONE_SLOT_ON_RANK="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--blocks", "2",
                        "--loci-per-gpu", "12000"], env=env, capture_output=True, text=True, timeout=850)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-3000:])
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    assert len(lines) == 1 and len(lines[0]) <= 4096
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0
    assert d["gather_check"] == {"ranks": 2, "blocks_equal_what_was_sent": True, "bytes_per_rank": 12000 * 168}
