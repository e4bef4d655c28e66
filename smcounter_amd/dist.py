"""Multi-GPU sharding of the locus list and the row gather.

The reference's only parallelism is a process pool over loci with results collected in
submission order (smCounter.py:683-685).  Loci share nothing, so here rank r of N calls a
contiguous range of the ordered locus list on its own GPU and the fixed-width rows
(include/smcounter_hip.h: smc_row) are gathered to rank 0 in rank order, which restores the
submission order.  One exchange per batch: equal shares (the bench) as a gather, shares of different
sizes (the command line's split by reads) as grouped sends / receives at their own sizes - the
gatherv RCCL does not have as a collective.  Works with the `nccl` (= RCCL) backend on GPUs and
with `gloo` on CPU tensors (used by the tests).
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import numpy as np


def shard_range(n_loci: int, rank: int, world: int):
    """Contiguous, near-equal split by locus count."""
    base, rem = divmod(n_loci, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_by_reads(n_reads: Sequence[int], world: int) -> List[int]:
    """Contiguous split balanced by the number of pileup reads (depth varies several-fold across a
    real panel).  Returns world+1 boundaries into the locus list."""
    c = np.concatenate([[0], np.cumsum(np.asarray(n_reads, np.int64))])
    total = int(c[-1])
    cuts = [0]
    for r in range(1, world):
        cuts.append(int(np.searchsorted(c, total * r / world, side="left")))
    cuts.append(len(n_reads))
    for i in range(1, len(cuts)):
        cuts[i] = max(cuts[i], cuts[i - 1])
    return cuts


def gather_rows(rows, gather_list: Optional[list], dst: int = 0):
    """Equal-size gather of a rank's row bytes to `dst` (weak-scaling bench path)."""
    import torch.distributed as dist
    dist.gather(rows, gather_list if dist.get_rank() == dst else None, dst=dst)


def _gatherv(payload, sizes: Sequence[int], dst: int):
    """Blocks of `sizes[r]` elements from every rank r to `dst` - each at its own size, nothing padded: the ranks other than `dst`
    post one send, `dst` one receive per peer that has something, as ONE group (torch.distributed.batch_isend_irecv: with the
    `nccl` backend ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd - the gatherv RCCL does not have as a collective, SURVEY.md
    8e; every peer's block travels its own xGMI link to `dst`).  -> list of tensors in rank order on `dst`, None elsewhere."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(), dist.get_rank()
    ops, out = [], None
    if rank == dst:
        out = [payload if r == dst else torch.empty(int(sizes[r]), dtype=payload.dtype, device=payload.device) for r in range(world)]
        ops = [dist.P2POp(dist.irecv, out[r], r) for r in range(world) if r != dst and sizes[r] > 0]
    elif sizes[rank] > 0:
        ops = [dist.P2POp(dist.isend, payload, dst)]
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    return out


def gatherv_rows(rows, counts: Sequence[int], row_bytes: int, dst: int = 0):
    """Variable-size gather of row bytes: rank r hands over counts[r] rows (known to every rank: the shares of the locus list);
    `dst` returns the concatenation in rank order (a uint8 tensor), other ranks return None.  Nothing is padded (round 6: an uneven
    shard_by_reads split no longer ships the largest share's size from every rank)."""
    import torch
    import torch.distributed as dist
    rank = dist.get_rank()
    sizes = [int(c) * row_bytes for c in counts]
    assert rows.numel() >= sizes[rank]
    parts = _gatherv(rows[:sizes[rank]].contiguous(), sizes, dst)
    if parts is None:
        return None
    return torch.cat(parts) if parts else rows[:0]


def gather_strings(local: Sequence[str], dst: int = 0):
    """Row strings of every rank, concatenated in rank order on `dst` (None elsewhere).  (Round 2's hand-over of the command
    line - pickled strings; the command line now gathers packed wire rows + an allele table: pack_shard / gatherv_bytes.)"""
    import torch.distributed as dist
    world, rank = dist.get_world_size(), dist.get_rank()
    parts = [None] * world if rank == dst else None
    dist.gather_object(list(local), parts, dst=dst)
    if rank != dst:
        return None
    return [s for part in parts for s in part]


def pack_shard(wire: np.ndarray, ref: Sequence[str], alleles: Sequence[Sequence[str]]) -> np.ndarray:
    """One rank's hand-over to the writing rank as ONE byte block: the packed wire rows (abi.WIRE_DTYPE, 168 B per locus: every
    printed number), the reference letters, and the rank's allele-string table - only the keys beyond the six fixed ones
    (indel alleles; SURVEY.md 8e: variable-length strings travel as a per-rank table beside the fixed-width rows).
    Layout: int64 n_rows, int64 blob_bytes | wire rows | JSON blob [ref letters, {locus: [extra keys]}]."""
    import json
    wire = np.ascontiguousarray(wire)
    n = len(wire)
    assert len(ref) == n and len(alleles) == n
    one = all(len(r) == 1 for r in ref)
    extras = {str(i): list(t[6:]) for i, t in enumerate(alleles) if len(t) > 6}
    blob = json.dumps(["".join(ref) if one else list(ref), extras]).encode()
    head = np.array([n, len(blob)], np.int64)
    return np.concatenate([head.view(np.uint8), wire.view(np.uint8).reshape(-1), np.frombuffer(blob, np.uint8)])


def unpack_shard(block: np.ndarray):
    """-> (wire rows, ref letters, allele tables) of pack_shard."""
    import json
    from .abi import WIRE_DTYPE
    from .pileup import BASE_ALLELES
    block = np.ascontiguousarray(block, np.uint8)
    n, nb = (int(x) for x in block[:16].view(np.int64))
    w_end = 16 + n * WIRE_DTYPE.itemsize
    wire = block[16:w_end].view(WIRE_DTYPE)
    ref, extras = json.loads(bytes(block[w_end:w_end + nb]).decode())
    ref = list(ref)
    base = list(BASE_ALLELES)
    alleles = [list(base) for _ in range(n)]
    for k, t in extras.items():
        alleles[int(k)] = base + list(t)
    return wire, ref, alleles


def gatherv_bytes(payload, dst: int = 0):
    """Byte blocks of different sizes from every rank to `dst` (a list of uint8 tensors in rank order there, None elsewhere):
    sizes first (one small all_gather), then the blocks as grouped sends / receives, each at its own size (_gatherv).  `payload`: a
    uint8 tensor on the device the backend moves (CPU for gloo, the rank's GPU for nccl)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    size = torch.tensor([payload.numel()], dtype=torch.int64, device=payload.device)
    sizes = [torch.zeros_like(size) for _ in range(world)]
    dist.all_gather(sizes, size)
    sizes = [int(t.item()) for t in sizes]
    return _gatherv(payload, sizes, dst)


def all_gather_status(err: Optional[str]) -> List[Optional[str]]:
    """Every rank's error message (None = fine), on every rank: the agreement step that precedes a collective so
    that one rank's failure ends the whole job at once instead of stranding its peers (smCounter.py:689-694 raises
    in the parent for any failed worker)."""
    import torch.distributed as dist
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, err)
    return out


def init_from_env():
    """(rank, local_rank, world) of a `python -m torch.distributed.run` launch, process group initialised
    (`nccl` = RCCL when a GPU is visible, else `gloo`; SMC_DIST_BACKEND overrides); (0, 0, 1) when not launched
    that way."""
    import os
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return 0, 0, 1
    import torch
    import torch.distributed as dist
    rank, local_rank = int(os.environ["RANK"]), int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("SMC_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    if os.environ.get("SMC_SHARE_GPU"):
        # every rank on GPU 0, gloo between them: a functional run of the multi-rank path on a box with one GPU
        local_rank, backend = 0, "gloo"
    if not dist.is_initialized():
        if backend == "nccl":
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    return rank, local_rank, world


class RowPipeline(object):
    """Double-buffered "compute rows, gather them to rank 0" loop: the gather of step i (asynchronous collective)
    overlaps the computation of step i + 1; a row buffer is reused only after its gather has completed.
    `bufs`: this rank's row buffers (>= 1 tensors of equal size); `produce(buf)` enqueues the work that fills one.
    With `world == 1` (no process group) it degenerates to calling `produce`.
    `streams` (optional, one per buffer): buffer b is produced on streams[b] - made the current stream BEFORE the wait for the
    buffer's previous gather and left current for the gather that follows, so that both order against the stream that fills the
    buffer; `produce(buf, b)` is then called with the buffer's index."""

    def __init__(self, bufs, collective: bool, dst: int = 0, streams=None):
        self.bufs = list(bufs)
        self.streams = list(streams) if streams else None
        self.collective = collective
        self.dst = dst
        self.pending = [None] * len(self.bufs)
        self.n = 0
        self.recv = None
        self.host_staged, self.recv_host = False, None
        if collective:
            import torch
            import torch.distributed as dist
            self.rank, self.world = dist.get_rank(), dist.get_world_size()
            self.host_staged = dist.get_backend() == "gloo" and self.bufs[0].is_cuda
            if self.rank == dst:
                self.recv = [[torch.empty_like(b) for _ in range(self.world)] for b in self.bufs]

    def step(self, produce):
        b = self.n % len(self.bufs)
        self.n += 1
        if self.streams is not None and self.streams[b] is not None:
            import torch
            torch.cuda.set_stream(self.streams[b])
        if self.pending[b] is not None:
            self.pending[b].wait()
        if self.streams is not None:
            produce(self.bufs[b], b)
        else:
            produce(self.bufs[b])
        if self.collective:
            import torch.distributed as dist
            if self.host_staged:
                # (gloo: the collective runs on host copies - a functional path for boxes where RCCL cannot be used, e.g. two
                # ranks sharing one GPU; not a performance path)
                h = self.bufs[b].cpu()
                if self.rank == self.dst and self.recv_host is None:
                    import torch
                    self.recv_host = [torch.empty_like(h) for _ in range(self.world)]
                dist.gather(h, self.recv_host if self.rank == self.dst else None, dst=self.dst)
                if self.rank == self.dst:
                    for r, t in enumerate(self.recv_host):
                        self.recv[b][r].copy_(t)
                return b
            self.pending[b] = dist.gather(self.bufs[b], self.recv[b] if self.rank == self.dst else None, dst=self.dst,
                                          async_op=True)
        return b

    def drain(self):
        for k, w in enumerate(self.pending):
            if w is not None:
                w.wait()
                self.pending[k] = None
