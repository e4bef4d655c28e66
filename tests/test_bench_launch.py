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
@pytest.mark.parametrize("n", [2, 4])
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
    assert "gathered to rank 0" in d["config"]["parallelism"]
    assert d["gather_check"] == {"ranks": 1, "blocks_equal_what_was_sent": True, "bytes_per_rank": 20000 * 168}
    assert d["parity"]["loci"] == 20000 and d["parity"]["mismatches"] == 0
