"""The N > 1 path on CPU: two gloo ranks shard the locus list, each produces its rows (here with
the CPU restatement standing in for the kernel - this test is about sharding and the gather), rank
0 gathers and must equal the single-process result in submission order."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as tdist
import torch.multiprocessing as mp

from conftest import ROOT


def _worker(rank, world, port, n_loci, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    tdist.init_process_group("gloo", rank=rank, world_size=world)
    from smcounter_amd import abi, dist, synth
    import oracle_lib
    cfg = synth.CONFIGS["C2"]
    P = synth.params_for(cfg)
    lo, hi = dist.shard_range(n_loci, rank, world)
    db = synth.generate_native(cfg, lo, hi, P, nthreads=1)
    rows = oracle_lib.call_batch(db, abi.c_params(P), abi.ROW_DTYPE)
    t = torch.from_numpy(rows.view(np.uint8).copy())
    counts = [b - a for a, b in (dist.shard_range(n_loci, r, world) for r in range(world))]
    out = dist.gatherv_rows(t, counts, abi.ROW_DTYPE.itemsize, dst=0)
    if rank == 0:
        q.put(out.numpy().tobytes())
    tdist.barrier()
    tdist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gather_restores_submission_order():
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from smcounter_amd import abi, synth
    import oracle_lib
    n_loci, world = 37, 2          # odd: ranks get unequal blocks -> exercises the padding
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_loci, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = np.frombuffer(q.get(timeout=240), np.uint8).view(abi.ROW_DTYPE)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    cfg = synth.CONFIGS["C2"]
    P = synth.params_for(cfg)
    want = oracle_lib.call_batch(synth.generate_native(cfg, 0, n_loci, P, nthreads=1), abi.c_params(P), abi.ROW_DTYPE)
    assert got.tobytes() == want.tobytes()
