"""Command line of the MI355X build: the reference's flags, inputs and output files
(smCounter.py:616-640 argParseInit, :645-909 main), with the per-locus worker pool replaced by batched
launches on one GPU.

    python -m smcounter_amd.cli --outPrefix example --bamFile example.bam --bedTarget example.bed \\
        --mtDepth 3612 --rpb 8.6 --refGenome hg19.fasta [...]

Differences that do not change results: `--nCPU` is accepted and ignored (the pool is gone);
`--bedtoolsPath` is accepted and ignored (merge/sort/intersect run in-process, bedops.py); when the two
repeat BEDs are absent the repeat flags are skipped with a note instead of failing inside bedtools.
"""
from __future__ import annotations

import argparse
import datetime
import os
import sys

from . import _lib, bamio, bedops, fasta, postfilter, runlog, vc, writers
from . import dist as smcdist
from .params import VcParams


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="Variant calling using molecular barcodes (MI355X build)",
                                fromfile_prefix_chars="@")
    p.add_argument("--outPrefix", default=None, required=True, help="prefix for output files")
    p.add_argument("--bamFile", default=None, required=True, help="BAM file")
    p.add_argument("--bedTarget", default=None, required=True, help="BED file for target region")
    p.add_argument("--mtDepth", default=None, required=True, type=int, help="Mean MT depth")
    p.add_argument("--rpb", default=None, required=True, type=float, help="Mean read pairs per MT")
    p.add_argument("--nCPU", type=int, default=1, help="ignored: loci are batched onto the GPU")
    p.add_argument("--minBQ", type=int, default=20, help="minimum base quality allowed for analysis")
    p.add_argument("--minMQ", type=int, default=30, help="minimum mapping quality allowed for analysis")
    p.add_argument("--hpLen", type=int, default=10, help="Minimum length for homopolymers")
    p.add_argument("--mismatchThr", type=float, default=6.0, help="average number of mismatches per 100 bases allowed")
    p.add_argument("--mtDrop", type=int, default=0, help="Drop MTs with lower than or equal to X reads.")
    p.add_argument("--maxMT", type=int, default=0, help="Randomly downsample to X MTs; 0 = 2.0 * mean MT depth")
    p.add_argument("--primerDist", type=int, default=2, help="filter variants that are within X bases to primer")
    p.add_argument("--threshold", type=int, default=0, help="Minimum prediction index for a variant to be called; "
                                                             "0 = chosen from the mean MT depth")
    p.add_argument("--refGenome", default=None, required=False, help="indexed FASTA of the reference genome")
    p.add_argument("--bedTandemRepeats", default=None, help="bed for UCSC tandem repeats")
    p.add_argument("--bedRepeatMaskerSubset", default=None, help="bed for RepeatMasker simple repeats, low complexity, "
                                                                  "microsatellite regions")
    p.add_argument("--bedtoolsPath", default=None, help="ignored: BED operations run in-process")
    p.add_argument("--runPath", default=None, help="path to working directory")
    p.add_argument("--logFile", default=None, help="log file")
    p.add_argument("--paramFile", default=None, help="optional parameter file; if given it replaces every other "
                                                     "parameter except --logFile")
    p.add_argument("--device", type=int, default=0, help="GPU index (single process only: under torch.distributed.run "
                                                          "every rank uses GPU LOCAL_RANK and this flag is ignored)")
    p.add_argument("--batchReads", type=int, default=4_000_000, help="pileup reads per device batch")
    p.add_argument("--sampler", choices=("reference", "philox"), default="reference",
                   help="how a locus with more UMIs than the cap (maxMT, or 2 x mtDepth) is down-sampled.  reference (default): as "
                        "smCounter.py:496-498 does - Python 2's random.sample over the barcode texts, seeded with the position string, "
                        "reproduced on the host (the same rows as smCounter).  philox: on the GPU, a counter-based generator "
                        "(Philox4x32-10) keyed by position and --samplerSeed - NOT the reference's sample: the rows of such loci differ "
                        "from smCounter's (another random subset of the same size); the same for every run, launch shape and GPU count")
    p.add_argument("--samplerSeed", type=int, default=0, help="seed of --sampler philox")
    return p


class _EarlyEngine(object):
    """Engine(device) created in a helper thread (binding the library, bringing up the GPU runtime, the context and its
    tables: ctypes calls, the interpreter lock is free meanwhile); get() joins and hands it over, or re-raises."""

    def __init__(self, device: int):
        import threading
        self._eng = self._err = None

        def work():
            try:
                from . import _lib
                from .engine import Engine
                _lib.load(with_torch=False)
                self._eng = _ENGINES.pop(device, None) or Engine(device)
            except BaseException as e:
                self._err = e
        self._t = threading.Thread(target=work, daemon=True)
        self._t.start()

    def get(self):
        self._t.join()
        if self._err is not None:
            raise self._err
        return self._eng


_ENGINES = {}          # device -> Engine kept for the process's next run (smc_create + the first allocations are ~ 0.1 s)


def _release_engine(eng):
    """The engine stays with the process: a second main() in the same process finds the context, its tables and its buffers
    again, and the command line does not spend 20 ms of its wall time freeing device memory the exiting process gives back
    anyway (SMC_CLOSE_ENGINE=1: close it, as round 2 did)."""
    if _lib.exp_env("SMC_CLOSE_ENGINE"):
        _ENGINES.pop(eng.device, None)
        eng.close()
    else:
        _ENGINES[eng.device] = eng


def call_shard(args, params: VcParams, loci, device: int, early=None):
    """The per-locus rows (strings, smCounter.py:599) of a run of loci: BAM decode -> device batches -> kernels."""
    from .engine import Engine
    ref = fasta.FastaFile(args.refGenome)
    eng = early.get() if early is not None else (_ENGINES.pop(device, None) or Engine(device))
    output = _Rows()
    decoder = os.environ.get("SMC_BAM_DECODER", "native")
    # (one process per GPU: the ranks of a node share its cores for decoding)
    # (LOCAL_WORLD_SIZE: WORLD_SIZE also counts the ranks of other nodes, which do not share these cores)
    per_node = int(os.environ.get("LOCAL_WORLD_SIZE") or os.environ.get("WORLD_SIZE", "1"))
    nthreads = bamio.host_threads(per_node)
    if decoder == "python":                                           # readable decoder, same batches
        batches = bamio.iter_pileup_batches(bamio.BamFile(args.bamFile), ref, loci, max_reads=args.batchReads)
    elif os.environ.get("SMC_PLANES", "device") == "host":             # planes built by the host threads, then uploaded
        batches = bamio.iter_device_batches_native(args.bamFile, ref, loci, params, max_reads=args.batchReads,
                                                   nthreads=nthreads)
    else:
        # default: the host decodes alignments, the GPU builds the planes from them (k_build_planes) and they stay in HBM
        from . import devplanes
        # (a batch only lives in HBM here - 16 B per read - so it can be eight times the host-built default)
        batches = devplanes.iter_resident_batches(args.bamFile, ref, loci, params, eng, max_reads=32 * args.batchReads,
                                                  nthreads=nthreads, all_planes=False, sampler=getattr(args, "sampler", "reference"),
                                                  sampler_seed=getattr(args, "samplerSeed", 0))
        # (a batch ahead in a helper thread: decoding and building batch i + 1 overlaps the kernels and the strings of batch i;
        # the two threads use different staging buffers of the engine, device work is ordered by the default stream)
        if not _lib.exp_env("SMC_NO_PREFETCH"):
            batches = _prefetch(batches, depth=1)
        for first, rb in batches:
            output.add(vc.vc_resident(rb, params, ref, eng))
            _report_boundary(eng.last_rows, rb.chrom, rb.pos)
        _release_engine(eng)
        return output.done()
    for first, pb in _prefetch(batches):
        output.add(vc.vc_batch(pb, params, ref, eng=eng))
    eng.close()
    return output.done()


def call_shard_rows(args, params: VcParams, loci, device: int):
    """A rank's share as NUMBERS: (rows abi.ROW_DTYPE[n], reference letters, allele tables) - the distributed command line
    sends these to rank 0 (packed: dist.pack_shard) which prints every row; no strings are made on the other ranks."""
    import numpy as np
    from . import abi, devplanes
    from .engine import Engine
    if not len(loci):
        return np.zeros(0, abi.ROW_DTYPE), [], []
    ref = fasta.FastaFile(args.refGenome)
    eng = Engine(device)
    per_node = int(os.environ.get("LOCAL_WORLD_SIZE") or os.environ.get("WORLD_SIZE", "1"))
    nthreads = bamio.host_threads(per_node)
    parts, refs, tables = [], [], []
    try:
        if os.environ.get("SMC_PLANES", "device") == "host":
            for _, db in bamio.iter_device_batches_native(args.bamFile, ref, loci, params, max_reads=args.batchReads, nthreads=nthreads):
                parts.append(eng.call_batch_host(db, params)); refs += list(db.ref); tables += list(db.alleles)
        else:
            batches = devplanes.iter_resident_batches(args.bamFile, ref, loci, params, eng, max_reads=32 * args.batchReads,
                                                      nthreads=nthreads, all_planes=False, sampler=getattr(args, "sampler", "reference"),
                                                      sampler_seed=getattr(args, "samplerSeed", 0))
            for _, rb in _prefetch(batches, depth=1):
                parts.append(vc.vc_resident_rows(rb, params, eng)); refs += list(rb.ref); tables += list(rb.alleles)
    finally:
        eng.close()
    return (np.concatenate(parts) if parts else np.zeros(0, abi.ROW_DTYPE)), refs, tables


def _report_boundary(out_rows, chrom, pos):
    """Log the loci whose PI lies within 1e-8 of a printing / gating boundary (rows.pi_boundary_loci): their text may differ
    from the reference's in the last printed digit or in the FILTER gate although the numbers agree to ~ 3e-9."""
    if out_rows is None:
        return
    from . import rows as _rows
    idx = _rows.pi_boundary_loci(out_rows)
    for l in idx.tolist():
        print("note: prediction index of %s:%d lies within 1e-8 of a printing boundary" % (chrom[l], int(pos[l])), file=sys.stderr)
    import numpy as np
    from . import abi
    for l in np.flatnonzero((out_rows["status"] & abi.ST_UNDERFLOW) != 0).tolist():
        print("note: a barcode at %s:%d has so many fragments that the posterior arithmetic left the double range; the "
              "reference's own numbers there depend on its multiplication order" % (chrom[l], int(pos[l])), file=sys.stderr)


class _Rows(list):
    """The shard's row strings; keeps the native printer's per-row int(PI) (rows.RowLines.pred) alongside when every batch
    came with one, for the post-filter and the writers."""
    pred = None

    def __init__(self):
        super().__init__()
        self._parts = []

    def add(self, part):
        self.extend(part)
        self._parts.append(getattr(part, "pred", None))

    def done(self):
        if self._parts and all(p is not None for p in self._parts):
            import numpy as np
            self.pred = np.concatenate(self._parts)
        return self


def _prefetch(it, depth: int = 2):
    """Run a batch generator in a helper thread, `depth` batches ahead: the native decoder releases the GIL, so
    decoding batch i + 1 overlaps the GPU call and the string formatting of batch i."""
    import queue
    import threading
    q = queue.Queue(maxsize=depth)
    done = object()

    def work():
        try:
            for item in it:
                q.put(item)
            q.put(done)
        except BaseException as e:          # re-raised in the consumer
            q.put(e)
    threading.Thread(target=work, daemon=True).start()
    while True:
        item = q.get()
        if item is done:
            return
        if isinstance(item, BaseException):
            raise item
        yield item


def main(args) -> int:
    """Same contract as the reference's main(args): accepts a Namespace or a dict of argument values,
    returns the PI threshold used (smCounter.py:909)."""
    # The run builds a few long lists of strings and no reference cycles: the cyclic collector would only rescan them, again
    # and again (several milliseconds per 20,000 loci).  Off for the duration of the call.
    import gc
    gc_was_on = gc.isenabled()
    gc.disable()
    try:
        return _main(args)
    finally:
        if gc_was_on:
            gc.enable()


def _main(args) -> int:
    t0 = datetime.datetime.now()
    print("smCounter started at " + str(t0))
    parser = build_parser()
    if not isinstance(args, argparse.Namespace):
        args = parser.parse_args(["--{0}={1}".format(k, v) for k, v in args.items()])
    elif args.paramFile is not None:
        args = parser.parse_args(("@" + args.paramFile,))
    for k, v in vars(args).items():
        print((k, v))
    if args.runPath is not None:
        os.chdir(args.runPath)
    if not args.refGenome:
        raise SystemExit("--refGenome is required (indexed FASTA)")

    params = VcParams(minBQ=args.minBQ, minMQ=args.minMQ, mtDepth=args.mtDepth, rpb=args.rpb, hpLen=args.hpLen,
                      mismatchThr=args.mismatchThr, mtDrop=args.mtDrop, maxMT=args.maxMT, primerDist=args.primerDist)
    early = None
    if int(os.environ.get("WORLD_SIZE", "1")) == 1 and os.environ.get("SMC_BAM_DECODER", "native") != "python":
        # a single process: the GPU runtime and the context come up (~ 0.1 s) in a helper thread while the target is expanded
        early = _EarlyEngine(args.device)
    loc_list = bedops.expand_loci(args.bedTarget)
    # One process per GPU when launched through torch.distributed.run: rank r calls a contiguous range of the
    # ordered locus list (loci share nothing, smCounter.py:683-685) on GPU LOCAL_RANK, rank 0 gathers the rows
    # in submission order and writes the files.
    rank, local_rank, world = smcdist.init_from_env()
    if world == 1:
        # a single process never touches torch.distributed: bind the C ABI without importing PyTorch first
        # (about a second of start-up; the host-buffer entry point needs none of it)
        from . import _lib
        _lib.load(with_torch=False)
    if world > 1:
        # contiguous ranges balanced by depth, not by locus count (amplicon depth varies several-fold): the BAI's
        # linear index gives compressed bytes per 16 kb window without decoding anything; every rank computes the
        # same cuts
        cuts = smcdist.shard_by_reads(bamio.locus_weights(args.bamFile, loc_list), world)
        lo, hi = cuts[rank], cuts[rank + 1]
    else:
        lo, hi = 0, len(loc_list)
    if world == 1:
        output = call_shard(args, params, loc_list[lo:hi], args.device, **({"early": early} if early is not None else {}))
        vc.raise_on_exception(output, loc_list[lo:hi])
    else:
        # A failing locus (or a decoder error) on one rank must not leave the others waiting in the collective
        # until the RCCL timeout: every rank first agrees on a status, then all raise together or all gather.
        import torch.distributed as tdist
        import numpy as np
        import torch
        from . import abi
        err, payload = None, None
        try:
            r_rows, r_ref, r_tab = call_shard_rows(args, params, loc_list[lo:hi], local_rank)
            if len(r_rows) != hi - lo:
                raise RuntimeError("%d rows for %d loci" % (len(r_rows), hi - lo))
            payload = smcdist.pack_shard(abi.pack_wire(r_rows), r_ref, r_tab)
        except Exception as e:                       # reported by every rank below
            err = "rank %d: %s: %s" % (rank, type(e).__name__, e)
        try:
            failed = [m for m in smcdist.all_gather_status(err) if m]
            if failed:
                raise RuntimeError("smCounter failed on %d of %d ranks: %s" % (len(failed), world, " | ".join(failed)))
            # ONE gather of byte blocks: 168-byte wire rows + the rank's allele-string table; a rank whose
            # share is empty sends an empty table
            t = torch.from_numpy(payload)
            if tdist.get_backend() == "nccl":
                t = t.to(torch.device("cuda", local_rank))
            blocks = smcdist.gatherv_bytes(t, dst=0)
            tdist.barrier()
        finally:
            if tdist.is_initialized():
                tdist.destroy_process_group()
        if rank != 0:
            return writers.pi_threshold(args.mtDepth, args.threshold)
        wires, refs, tabs = [], [], []
        for b in blocks:
            w, rf, tb = smcdist.unpack_shard(b.cpu().numpy())
            wires.append(w); refs += rf; tabs += tb
        all_rows = abi.unpack_wire(np.concatenate(wires)) if wires else np.zeros(0, abi.ROW_DTYPE)
        view = vc.LocusView([c for c, _ in loc_list], [int(p) for _, p in loc_list], refs, tabs)
        output = vc._strings(all_rows, view, params, fasta.FastaFile(args.refGenome))
        _report_boundary(all_rows, view.chrom, view.pos)
        vc.raise_on_exception(output, loc_list)

    print("begin variant filtering and output")
    have_rep = [b for b in (args.bedTandemRepeats, args.bedRepeatMaskerSubset) if b and os.path.exists(b)]
    if len(have_rep) < 2:
        print("note: repeat tracks not given or not found; RepT/RepS/LowC/SL flags are not applied", file=sys.stderr)
    trf, rm = postfilter.load_repeat_regions(
        args.bedTarget,
        args.bedTandemRepeats if args.bedTandemRepeats and os.path.exists(args.bedTandemRepeats) else None,
        args.bedRepeatMaskerSubset if args.bedRepeatMaskerSubset and os.path.exists(args.bedRepeatMaskerSubset) else None)
    pred = getattr(output, "pred", None)                  # (single process: the printer's int(PI) per row)
    output = postfilter.apply_repeat_filters(output, trf, rm, pred=pred)
    threshold = writers.pi_threshold(args.mtDepth, args.threshold)
    writers.write_outputs(args.outPrefix, output, threshold, pred=pred)
    t1 = datetime.datetime.now()
    print("smCounter completed running at " + str(t1))
    print("smCounter total time: " + str(t1 - t0))
    return threshold


if __name__ == "__main__":
    ns = build_parser().parse_args()
    if ns.logFile:
        runlog.init(ns.logFile)
    try:
        main(ns)
    finally:
        runlog.close()
