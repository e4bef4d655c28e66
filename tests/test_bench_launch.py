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
