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


def _wire_worker(rank, world, port, path, q):
    """Packed wire rows (168 B, abi.WIRE_DTYPE) instead of full rows: each rank packs its block (numpy mirror of
    k_pack_rows - the GPU suite checks the kernel against it byte for byte), rank 0 gathers, unpacks (smc_unpack_rows) and
    prints."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    tdist.init_process_group("gloo", rank=rank, world_size=world)
    from smcounter_amd import abi, dist
    import oracle_lib
    from conftest import load_golden
    pb, db, P, refp, expected = load_golden(path)
    rows = oracle_lib.call_batch(db, abi.c_params(P), abi.ROW_DTYPE)
    lo, hi = dist.shard_range(len(rows), rank, world)
    wire = abi.pack_wire(rows[lo:hi])
    counts = [b - a for a, b in (dist.shard_range(len(rows), r, world) for r in range(world))]
    out = dist.gatherv_rows(torch.from_numpy(wire.view(np.uint8).copy()), counts, abi.WIRE_DTYPE.itemsize, dst=0)
    if rank == 0:
        q.put(out.numpy().tobytes())
    tdist.barrier()
    tdist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gather_of_packed_wire_rows_prints_the_same_strings():
    """the gather moves 168-byte wire rows; unpacked on rank 0 they print the same 45-field strings
    as the full 432-byte rows (golden stress vectors: filters, bi-allelic loci, zero coverage, indel alleles)."""
    from conftest import golden_files, load_golden
    from smcounter_amd import abi, rows as rowsmod
    import oracle_lib
    path = [p for p in golden_files() if "stress2" in p][0]
    pb, db, P, refp, expected = load_golden(path)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_wire_worker, args=(r, 2, port, path, q)) for r in range(2)]
    for p in procs:
        p.start()
    wire = np.frombuffer(q.get(timeout=240), np.uint8).view(abi.WIRE_DTYPE)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    full = oracle_lib.call_batch(db, abi.c_params(P), abi.ROW_DTYPE)
    assert len(wire) == len(full) and abi.WIRE_DTYPE.itemsize == 168 <= 200
    got = abi.unpack_wire(wire)
    assert rowsmod.format_rows(got, db, P, refp) == rowsmod.format_rows(full, db, P, refp)
    assert rowsmod.format_rows(got, db, P, refp, native=False) == rowsmod.format_rows(full, db, P, refp, native=False)
    assert (got["cand"]["flt_applied"].any() and got["biallelic"].any() and (got["status"] == 1).any())
    # every field the wire carries survives bit for bit
    for f in ("status", "cvg", "all_frag", "all_mt", "used_frag", "used_mt", "mt3", "mt5", "mt7", "mt10", "dp", "umt", "vsm",
              "biallelic"):
        assert np.array_equal(got[f], full[f]), f
    assert got["pi"].tobytes() == full["pi"].tobytes()
    for f in ("allele", "flt", "flt_applied", "vmf_lt_099", "vdp", "vmt", "vsm"):
        assert np.array_equal(got["cand"][f], full["cand"][f]), f
    assert got["cand"]["pi"].tobytes() == full["cand"]["pi"].tobytes()


def _cli_worker(rank, world, port, tmp, q, weights=None):
    """One rank of the distributed command line; the GPU call is swapped for the CPU restatement (this test is
    about sharding, the gather of packed wire rows + allele tables, and printing + the writers on rank 0).  `weights`:
    locus weights that replace the BAI-derived ones (to force uneven and EMPTY shares)."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                      WORLD_SIZE=str(world), SMC_DIST_BACKEND="gloo")
    from smcounter_amd import abi, bamio, cli, fasta
    import oracle_lib

    def cpu_call_shard_rows(args, params, loci, device):
        if not len(loci):
            return np.zeros(0, abi.ROW_DTYPE), [], []
        ref = fasta.FastaFile(args.refGenome)
        parts, refs, tabs = [], [], []
        for _, db in bamio.iter_device_batches_native(args.bamFile, ref, loci, params, max_reads=args.batchReads):
            parts.append(oracle_lib.call_batch(db, abi.c_params(params), abi.ROW_DTYPE)); refs += list(db.ref); tabs += list(db.alleles)
        return np.concatenate(parts), refs, tabs
    cli.call_shard_rows = cpu_call_shard_rows
    if weights is not None:
        bamio.locus_weights = lambda path, loci: np.asarray(weights(len(loci)), np.float64)
    prefix = os.path.join(tmp, "dist")
    thr = cli.main(dict(outPrefix=prefix, bamFile=os.path.join(tmp, "case.bam"), bedTarget=os.path.join(tmp, "case.bed"),
                        mtDepth=12, rpb=3.0, hpLen=8, refGenome=os.path.join(tmp, "case.fa"), threshold=10,
                        batchReads=300))
    if rank == 0:
        q.put(thr)


def _skewed_weights(n):
    """All the weight on the first locus and the last third: of four ranks, two get NOTHING and the others unequal shares."""
    w = [0.0] * n
    w[0] = 100.0
    for i in range(2 * n // 3, n):
        w[i] = 1.0
    return w


def _single_process_expectation(tmp_path):
    import bam_fixture  # noqa: F401
    from smcounter_amd import abi, bamio, bedops, fasta, postfilter, rows
    from smcounter_amd.params import VcParams
    import oracle_lib
    P = VcParams(mtDepth=12, rpb=3.0, hpLen=8)
    fa = fasta.FastaFile(str(tmp_path / "case.fa"))
    loci = bedops.expand_loci(str(tmp_path / "case.bed"))
    want = []
    for _, db in bamio.iter_device_batches_native(str(tmp_path / "case.bam"), fa, loci, P):
        want.extend(rows.format_rows(oracle_lib.call_batch(db, abi.c_params(P), abi.ROW_DTYPE), db, P, fa))
    return postfilter.apply_repeat_filters(want, {}, {}), loci


@pytest.mark.timeout(300)
def test_four_ranks_uneven_shares_and_empty_ranks_write_the_single_process_files(tmp_path):
    """Four ranks, shares forced to (1 locus, none, none, the rest): the ranks without loci take part in the status agreement
    and the gather with an empty block; rank 0 prints every row from the packed wire rows + allele tables - the same all.txt
    as one process (the fixture has indel alleles: the tables matter)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import bam_fixture
    from smcounter_amd import dist
    case = bam_fixture.make_case(str(tmp_path))
    for k, name in (("bam", "case.bam"), ("bed", "case.bed"), ("fasta", "case.fa")):
        os.replace(case[k], str(tmp_path / name))
    if os.path.exists(case["fasta"] + ".fai"):
        os.replace(case["fasta"] + ".fai", str(tmp_path / "case.fa.fai"))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_cli_worker, args=(r, 4, port, str(tmp_path), q, _skewed_weights)) for r in range(4)]
    for p in procs:
        p.start()
    assert q.get(timeout=240) == 10
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    want, loci = _single_process_expectation(tmp_path)
    cuts = dist.shard_by_reads(_skewed_weights(len(loci)), 4)
    assert cuts[1] == cuts[2] == cuts[3] and 0 < cuts[1] < len(loci)          # two ranks without a locus
    got = open(str(tmp_path / "dist.smCounter.all.txt")).read().split("\n")[1:-1]
    assert got == want and len(got) == len(loci)
    assert any("INS|" in r or "DEL|" in r or len(r.split("\t")[2]) > 1 or len(r.split("\t")[3]) > 1 for r in got)


def _eight_way_weights(n):
    """Eight ranks (integer weights, total 80, a cut every 10): locus 0 alone on rank 0, nothing for ranks 1-3, two loci for
    rank 4, one each for ranks 5 and 6, the rest for rank 7."""
    w = [0] * n
    w[0] = 40
    w[1] = w[2] = 5
    w[3] = w[4] = w[5] = 10
    return w


@pytest.mark.timeout(420)
def test_eight_ranks_with_empty_single_and_large_shares_write_the_single_process_files(tmp_path):
    """The launch shape of the driver's 8-GPU run (one process per rank, one gather to rank 0) with shares of 0, 1, 2 and many
    loci: the same all.txt as one process writes."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import bam_fixture
    from smcounter_amd import dist
    case = bam_fixture.make_case(str(tmp_path))
    for k, name in (("bam", "case.bam"), ("bed", "case.bed"), ("fasta", "case.fa")):
        os.replace(case[k], str(tmp_path / name))
    if os.path.exists(case["fasta"] + ".fai"):
        os.replace(case["fasta"] + ".fai", str(tmp_path / "case.fa.fai"))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_cli_worker, args=(r, 8, port, str(tmp_path), q, _eight_way_weights)) for r in range(8)]
    for p in procs:
        p.start()
    assert q.get(timeout=360) == 10
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    want, loci = _single_process_expectation(tmp_path)
    cuts = dist.shard_by_reads(_eight_way_weights(len(loci)), 8)
    shares = [cuts[r + 1] - cuts[r] for r in range(8)]
    assert shares[:7] == [1, 0, 0, 0, 2, 1, 1] and shares[7] == len(loci) - 5 > 2, shares
    got = open(str(tmp_path / "dist.smCounter.all.txt")).read().split("\n")[1:-1]
    assert got == want and len(got) == len(loci)


@pytest.mark.timeout(300)
def test_two_rank_command_line_writes_the_single_process_files(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import bam_fixture
    from smcounter_amd import abi, bamio, bedops, fasta, postfilter, rows, writers
    from smcounter_amd.params import VcParams
    import oracle_lib
    case = bam_fixture.make_case(str(tmp_path))
    for k, name in (("bam", "case.bam"), ("bed", "case.bed"), ("fasta", "case.fa")):
        os.replace(case[k], str(tmp_path / name))
    if os.path.exists(case["fasta"] + ".fai"):
        os.replace(case["fasta"] + ".fai", str(tmp_path / "case.fa.fai"))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_cli_worker, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    assert q.get(timeout=240) == 10
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # single-process expectation
    P = VcParams(mtDepth=12, rpb=3.0, hpLen=8)
    fa = fasta.FastaFile(str(tmp_path / "case.fa"))
    loci = bedops.expand_loci(str(tmp_path / "case.bed"))
    want = []
    for _, db in bamio.iter_device_batches_native(str(tmp_path / "case.bam"), fa, loci, P):
        want.extend(rows.format_rows(oracle_lib.call_batch(db, abi.c_params(P), abi.ROW_DTYPE), db, P, fa))
    want = postfilter.apply_repeat_filters(want, {}, {})
    got = open(str(tmp_path / "dist.smCounter.all.txt")).read().split("\n")[1:-1]
    assert got == want and len(got) == len(loci)
    assert os.path.getsize(str(tmp_path / "dist.smCounter.cut.vcf")) > 0


def _failing_cli_worker(rank, world, port, tmp, q):
    """Rank 1's shard raises (a failing locus / decoder error); both ranks must end promptly with the same error."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                      WORLD_SIZE=str(world), SMC_DIST_BACKEND="gloo")
    from smcounter_amd import cli

    def shard(args, params, loci, device):
        if rank == 1:
            raise ValueError("decoder broke on purpose")
        from smcounter_amd import abi
        return np.zeros(len(loci), abi.ROW_DTYPE), ["A"] * len(loci), [list("ATGCN") + ["DEL"]] * len(loci)
    cli.call_shard_rows = shard
    try:
        cli.main(dict(outPrefix=os.path.join(tmp, "fail"), bamFile=os.path.join(tmp, "case.bam"),
                      bedTarget=os.path.join(tmp, "case.bed"), mtDepth=12, rpb=3.0, hpLen=8,
                      refGenome=os.path.join(tmp, "case.fa"), threshold=10))
        q.put((rank, "returned"))
    except RuntimeError as e:
        q.put((rank, str(e)))


@pytest.mark.timeout(300)
def test_one_failing_rank_ends_every_rank_together(tmp_path):
    """a rank that raises before the gather used to strand its peers in the collective."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import bam_fixture
    case = bam_fixture.make_case(str(tmp_path))
    for k, name in (("bam", "case.bam"), ("bed", "case.bed"), ("fasta", "case.fa")):
        os.replace(case[k], str(tmp_path / name))
    if os.path.exists(case["bam"] + ".bai"):
        os.replace(case["bam"] + ".bai", str(tmp_path / "case.bam.bai"))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_failing_cli_worker, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for r in (0, 1):
        assert "failed on 1 of 2 ranks" in got[r] and "rank 1: ValueError: decoder broke on purpose" in got[r]
    assert not os.path.exists(str(tmp_path / "fail.smCounter.all.txt"))


def _pipe_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    tdist.init_process_group("gloo", rank=rank, world_size=world)
    from smcounter_amd import dist
    bufs = [torch.zeros(64, dtype=torch.uint8), torch.zeros(64, dtype=torch.uint8)]
    pipe = dist.RowPipeline(bufs, collective=True)
    seen = []
    for i in range(5):
        b = pipe.step(lambda buf, i=i: buf.fill_(10 * i + rank))
        seen.append(b)
    pipe.drain()
    if rank == 0:
        # buffers alternate; after the drain rank 0 holds every rank's rows of the last two steps
        got = {b: [int(t[0]) for t in pipe.recv[b]] for b in (0, 1)}
        q.put((seen, got))
    tdist.barrier()
    tdist.destroy_process_group()


@pytest.mark.timeout(300)
def test_row_pipeline_overlaps_and_keeps_order():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_pipe_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    seen, got = q.get(timeout=240)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert seen == [0, 1, 0, 1, 0]
    assert got == {0: [40, 41], 1: [30, 31]}            # step 4 went through buffer 0, step 3 through buffer 1
