"""Device-resident batches: BAM -> alignments (host decode) -> planes built ON THE GPU (smc_build_planes) -> hot path.

The per-pileup-read half of the reference's feature extraction and barcode bookkeeping (smCounter.py:316-366, :371-452,
:462-471) runs in csrc/k_build_planes.inc; the host keeps the per-alignment half (csrc/smc_bam.cpp: smc_bam_alignments)
and whatever needs strings: the texts of indel alleles, the reference's down-sampling on barcode texts (:496-498, py2
semantics).  A run the device path does not take (a locus deeper than smc_build_max_depth(), an alignment with neither
READ1 nor READ2, an overflow the kernel reports) is built by the host builder (smc_bam_planes) and uploaded into the same
device arrays - the same batch either way up to the numbering of barcodes and fragments within a locus, which the layout
contract leaves to the builder (planecheck.py; tests/test_gpu_devplanes.py compares them).
"""
from __future__ import annotations

import ctypes
import dataclasses
import os
from typing import List, Sequence, Tuple

import numpy as np

from . import _lib, abi, bamio
from .features import LF_SAMPLED, LOCUS_DTYPE, USTART_DROPPED
from .pileup import BASE_ALLELES

_BASE_TABLE = list(BASE_ALLELES)           # the allele table of a locus that met only the six fixed keys (shared: read-only)


_TIMES = {"decode": 0.0, "upload+launch": 0.0, "kernel (sync)": 0.0}      # SMC_DEVPLANES_TIMING=1: seconds per stage


@dataclasses.dataclass
class ResidentBatch:
    """A batch whose planes live in HBM (engine.DevBuf allocations: no PyTorch in this path); the descriptors and what row
    formatting needs are on the host."""
    planes: list               # [meta, umi, frag, dist, umi_start] device buffers (`data_ptr()`), uint32 words; the four
                               # raw-field planes are None when the batch was built with all_planes=False
    loci: np.ndarray           # LOCUS_DTYPE[n_loci] (host copy; offsets are batch-relative)
    chrom: List[str]
    pos: np.ndarray
    ref: List[str]
    alleles: List[List[str]]
    n_device_runs: int = 0     # runs built by k_build_planes / by the host builder (fallback)
    n_host_runs: int = 0
    n_slots: int = 0
    n_ustart: int = 0
    words: object = None       # the read words (one per read: what the locus kernels read), device buffer; 32 bits each, or 16
                               # (include/smcounter_hip.h: smc_read_word16) when its `word_bits` says so

    @property
    def n_loci(self) -> int:
        return len(self.loci)

    def to_host(self):
        """-> features.DeviceBatch with the planes copied back (tests, the CPU restatement)."""
        from .features import DeviceBatch
        m, u, f, d = (t.download(np.uint32, self.n_slots) for t in self.planes[:4])
        us = self.planes[4].download(np.uint32, self.n_ustart)
        return DeviceBatch(loci=self.loci.copy(), meta=m, umi=u, frag=f, dist=d, umi_start=us, chrom=list(self.chrom),
                           pos=self.pos.copy(), ref=list(self.ref), alleles=[list(t) for t in self.alleles])


FIRST_RUN_WHOLE_BYTES = 48 << 20   # ... unless the whole stretch (up to 65536 loci) is no more than this in the file
FIRST_RUN_LOCI = 64            # loci of a file's first run (iter_resident_batches): enough to learn the depth from
BUILD_ST_NARROW = 32           # smc_build_planes_w16's status bit: the run has no room in 16-bit read words
NARROW = "narrow"              # what build_run answers then


def class16_table() -> np.ndarray:
    """smc_read_class's number -> smc_class16's code (included << 4 | index), 31 where there is no class (include/smcounter_hip.h)."""
    t = np.full(32, 31, np.uint32)
    for c in range(22):
        if c < 2:
            t[c] = (c & 1) << 4
        elif c < 6:
            t[c] = ((c & 1) << 4) | (1 + ((c - 2) >> 1))
        else:
            rev, sub = (c - 6) >> 3, (c - 6) & 7
            t[c] = 3 + 2 * rev + sub if sub < 2 else 16 | (3 + 6 * rev + (sub - 2))
    return t


def words16_from_32(w: np.ndarray):
    """32-bit read words -> 16-bit ones (smc_read_word16), or None when one of them has no room: an allele id beyond 15, a
    quality beyond 63, a class that is none (a word of zero - the pad behind a locus's last read - stays zero)."""
    w = np.asarray(w, np.uint32)
    cls = w >> np.uint32(27)
    c16 = class16_table()[cls]
    real = w != 0
    if bool(((((w & np.uint32(0xFF)) > 15) | (((w >> np.uint32(8)) & np.uint32(0xFF)) > 63) | (c16 == 31)) & real).any()):
        return None
    h = (w & np.uint32(15)) | ((w >> np.uint32(16)) & np.uint32(1)) << np.uint32(4) | ((c16 >> np.uint32(4)) & np.uint32(1)) << np.uint32(5) | \
        (c16 & np.uint32(3)) << np.uint32(6) | ((w >> np.uint32(8)) & np.uint32(63)) << np.uint32(8) | ((c16 >> np.uint32(2)) & np.uint32(3)) << np.uint32(14)
    return np.where(real, h, 0).astype(np.uint16)


def words32_from_16(h: np.ndarray) -> np.ndarray:
    """... and back (smc_read_word32): what tests compare with the host-built words."""
    h = np.asarray(h, np.uint16).astype(np.uint32)
    inv = np.full(32, 31, np.uint32)
    t = class16_table()
    for c in range(22):
        inv[t[c]] = c
    c16 = ((h >> np.uint32(5)) & np.uint32(1)) << np.uint32(4) | ((h >> np.uint32(6)) & np.uint32(3)) | ((h >> np.uint32(14)) & np.uint32(3)) << np.uint32(2)
    cls = inv[c16]
    inc = np.where(cls < 6, cls & 1, ((cls - 6) & 7) >= 2).astype(np.uint32)
    w = (h & np.uint32(15)) | ((h >> np.uint32(8)) & np.uint32(63)) << np.uint32(8) | ((h >> np.uint32(4)) & np.uint32(1)) << np.uint32(16) | \
        inc << np.uint32(17) | np.uint32(1 << 18) | cls << np.uint32(27)
    return np.where(h != 0, w, 0).astype(np.uint32)


def pack_words_host(meta: np.ndarray, frag: np.ndarray, loci: np.ndarray) -> np.ndarray:
    """The read words (include/smcounter_hip.h: smc_read_word) of a host-built batch, as smc_pack_words makes them on the device:
    allele and quality from the meta word, the class from the frag word, "first read of its fragment" from the slots."""
    from .features import FRAG_SLOT_MASK, FRAG_CLASS_SHIFT
    ns = len(meta)
    slot = frag & np.uint32(FRAG_SLOT_MASK)
    nf = np.ones(ns, bool)
    nf[1:] = slot[1:] != slot[:-1]
    start = 4 * loci["read_off4"].astype(np.int64)
    n = loci["n_reads"].astype(np.int64)
    nf[start[n > 0]] = True
    d = np.zeros(ns + 1, np.int64)
    np.add.at(d, start, 1)
    np.add.at(d, start + n, -1)
    valid = np.cumsum(d[:-1]) > 0
    cls = frag >> np.uint32(FRAG_CLASS_SHIFT)
    # the slot contract, checked as k_pack_words.inc checks it: within a locus the slots start at 0, step by 0 or 1 and end at
    # n_frag - 1; a read that breaks it gets class 31 (no read class), which the kernels answer with SMC_ST_BAD_INPUT
    prev = np.empty(ns, np.uint32)
    prev[1:] = slot[:-1]
    prev[:1] = 0
    prev[start[n > 0]] = np.uint32(0xFFFFFFFF)
    step = slot - prev
    bad = step > np.uint32(1)
    last = (start + n - 1)[n > 0]
    bad[last] |= slot[last] != (loci["n_frag"][n > 0].astype(np.int64) - 1).astype(np.uint32)
    nf = step != 0
    cls = np.where(bad, np.uint32(31), cls)
    # bits 17 / 18: the read is included / its class is a known one (include/smcounter_hip.h: smc_class_bits)
    c = np.arange(32)
    inc = np.where(c < 6, c & 1, ((c - 6) & 7) >= 2).astype(np.uint32)
    bits = np.where(c < 22, (np.uint32(4) | inc << np.uint32(1)), np.uint32(0)).astype(np.uint32) << np.uint32(16)
    w = (meta & np.uint32(0xFFFF)) | (nf.astype(np.uint32) << np.uint32(16)) | bits[cls] | (cls << np.uint32(27))
    return np.where(valid, w, np.uint32(0)).astype(np.uint32)


READS_PER_BYTE = 6        # pileup reads a batch is sized for per compressed byte of the file under its targets (typical: ~ 2.5)


def iter_resident_batches(path: str, fasta, loci: Sequence[Tuple[str, str]], params, eng, max_reads: int = 128_000_000,
                          nthreads: int = 0, force_host: bool = False, all_planes: bool = True, sampler: str = "reference",
                          sampler_seed: int = 0):
    """BAM -> `ResidentBatch` chunks: same loci per chunk as bamio.iter_device_batches_native, planes built on the GPU.
    `all_planes=False`: only what the locus kernels read - the read words and umi_start - is built and kept (a fifth of the
    device memory of a batch, a quarter of the builder's stores); the four raw-field planes (for checks) are then None.
    (Round 5 built and measured decoding run i + 1 on a helper thread - a second decoder handle, the streaming cursor handed
    over - beside the upload, build and call of run i: at the example run's depth a run is 14 ms of decoding against 4 ms of
    everything else, the helper's decodes were 30 % slower than the main thread's and its exit cost 40 ms per file: 13 k loci/s
    against 22-24 k without.  Taken out again; what stayed is the sizing of the runs below.)"""
    from .engine import DevBuf
    L = eng.L
    bam = bamio.NativeBam(path)
    nthreads = nthreads or bamio.host_threads()
    max_depth = L.smc_build_max_depth()
    cp = abi.c_params(params)
    i, n = 0, len(loci)
    # last locus of the stretch of consecutive positions (same chromosome) every locus lies in
    if n:
        pos_all = np.array([p for _, p in loci]).astype(np.int64)
        chrom_all = [c for c, _ in loci]
        brk = pos_all[1:] != pos_all[:-1] + 1
        if chrom_all.count(chrom_all[0]) != n:
            brk |= np.array([chrom_all[k] != chrom_all[k + 1] for k in range(n - 1)], bool)
        ends = np.append(np.flatnonzero(brk), n - 1)
        stretch_end = ends[np.searchsorted(ends, np.arange(n))]
    # The device arrays of a batch are sized for max_reads - but not for more than the FILE can hold (a 2 MB BAM made every
    # batch allocate three 576 MB barcode arrays and a read-word block large enough to send the library through its search among
    # twelve allocations).  The linear index says how many compressed bytes lie under the target stretches; a batch takes at most
    # READS_PER_BYTE pileup reads per byte of them (a 150-base read is 60-100 compressed bytes and at most 150 pileup reads; an estimate
    # that is too low only ends batches earlier - the decoder is handed `max_reads - total` and stops there).
    if n and not _lib.exp_env("SMC_NO_FILE_SIZED_BATCH"):
        try:
            nb, seen = 0, set()
            for e in np.unique(stretch_end).tolist():
                b = int(np.searchsorted(stretch_end, e, side="left"))
                k = bam.span_bytes(chrom_all[b], int(pos_all[b]) - 1, int(pos_all[e]))
                if k < 0:
                    nb = -1
                    break
                nb += int(k)
            if nb >= 0:
                max_reads = int(min(max_reads, max(4_000_000, READS_PER_BYTE * nb)))
        except Exception:
            pass
    cap = max_reads + (max_reads >> 3) + 65536
    per_locus = 0.0          # pileup reads per locus of the previous run: sizes the next run (a run is decoded as a whole)

    def span_of(i, total, per_locus, first):
        """The run that starts at locus i of a batch that holds `total` reads so far: (chrom, lo, hi, reads it may take)."""
        chrom = loci[i][0]
        # (the decoder inflates everything up to the run's end: a run much longer than what max_reads lets through
        # would be decoded again by the next one)
        # (the floor was 1024 loci and the factor 1.15 until round 5: at the example run's 58,000x a batch's first run took 550
        # loci, its second was asked for 1024 - half a 2000-locus file inflated and parsed - to take the 51 the batch still had room for)
        span_cap = 65536 if per_locus <= 0 else int(max(64, min(65536, 1.02 * (max_reads - total) / per_locus)))
        if per_locus <= 0 and i == first:
            # nothing decoded yet: a short first run tells the depth, the runs after it are sized by it.  (Until round 5 the first
            # run was sized from the compressed bytes the linear index gives for the stretch, at 2.5 pileup reads per
            # byte: on the 58,000x fixture that asked for all 2000 loci - 40 ms of decoding - to keep the 550 a batch has room for.)
            # A stretch whose compressed bytes are few (the linear index knows) is taken whole: decoding all of it costs less than a
            # run of its own for the first loci would.
            span_cap = FIRST_RUN_LOCI
            jj = min(int(stretch_end[i]), i + 65536 - 1)
            nb = bam.span_bytes(chrom, int(pos_all[i]) - 1, int(pos_all[jj]))
            if 0 <= nb < FIRST_RUN_WHOLE_BYTES:
                span_cap = 65536
            if _lib.exp_env("SMC_FIRST_RUN_8192"):                                             # (the switch: measurement)
                span_cap = 8192
        j = min(int(stretch_end[i]), i + span_cap - 1)
        return chrom, int(pos_all[i]) - 1, int(pos_all[j]), max_reads - total

    import time
    try:
        while i < n:
            first = i
            planes = [DevBuf(eng, 4 * cap) if all_planes else None for k in range(4)]
            # 16-bit read words where nothing but the words is kept, until a run has no room in them (an allele id beyond 15, a
            # quality beyond 63): that batch is then built again, and the rest of the file, with 32-bit words
            bits = 16 if (not all_planes and eng.word_bits == 16 and 0 <= params.minBQ <= 63) else 32
            words = DevBuf(eng, (bits // 8) * cap, walk_output=True)      # (a large one is chosen by the write-pattern probe: engine.DevBuf)
            words.word_bits = bits
            narrow = False
            uaux = [DevBuf(eng, 4 * (cap + 8192)) for _ in range(3)]      # umi_start, u_gid, u_finc
            LC, chroms, poss, refs, tables = [], [], [], [], []
            total = slots = n_loc = 0
            n_dev = n_host = 0
            while i < n and total < max_reads:
                key = span_of(i, total, per_locus, first)
                chrom, lo, hi = key[0], key[1], key[2]
                run_ref = fasta.fetch(chrom, lo, hi).upper()
                umi_base = slots + n_loc
                done = None
                if not force_host:
                    T = _TIMES if os.environ.get("SMC_DEVPLANES_TIMING") else None
                    t0 = time.perf_counter()
                    A = bam.alignments_run(chrom, lo, hi, key[3], params, nthreads, host_array=eng.pinned)
                    if T is not None:
                        T["decode"] += time.perf_counter() - t0
                    done = build_run(A, L, eng, cp, params, chrom, lo, fasta, run_ref, [words] + planes, uaux, slots, umi_base, cap, max_depth,
                                     bam.allele_key, bam.barcode_name, sampler=sampler, sampler_seed=sampler_seed,
                                     barcode_idents=bam.barcode_idents)
                    if done == NARROW:
                        narrow = True
                        break
                if done is None:
                    # host builder (the run is not one the device path takes): same planes, uploaded
                    nl, hp, ustart, lc, tb = bam.planes_run(chrom, lo, hi, max_reads - total, params, run_ref, nthreads, fasta)
                    if sampler == "philox" and params.ds > 0 and bool((lc["n_umi"] > params.ds).any()):
                        import warnings
                        warnings.warn("--sampler philox: the run %s:%d-%d was built on the host (not one the device builder takes): its loci "
                                      "over the UMI cap keep the reference's sample" % (chrom, lo + 1, lo + nl))
                    ns = len(hp[0])
                    if slots + ns > cap or umi_base + len(ustart) > cap + 8192:
                        raise bamio.BamError("run of %d read slots does not fit the device arena" % ns)
                    for k in range(4):
                        if planes[k] is not None:
                            planes[k].upload(hp[k], 4 * slots)
                    hw = pack_words_host(hp[0], hp[2], lc)
                    if bits == 16:
                        hw = words16_from_32(hw)
                        if hw is None:
                            narrow = True
                            break
                    words.upload(hw, (bits // 8) * slots)
                    uaux[0].upload(ustart, 4 * umi_base)
                    lc = lc.copy()
                    lc["read_off4"] += slots // 4
                    lc["umi_off"] += umi_base
                    n_host += 1
                else:
                    nl, ns, lc, tb = done
                    n_dev += 1
                LC.append(lc)
                if nl:
                    per_locus = max(1.0, float(lc["n_reads"].sum()) / nl)
                slots += ns
                n_loc += nl
                total += int(lc["n_reads"].sum())
                chroms += [chrom] * nl
                poss.append(np.arange(lo + 1, lo + 1 + nl, dtype=np.int64))
                refs += list(run_ref[:nl]) + [""] * max(0, nl - len(run_ref))
                tables += tb
                i += nl
            if narrow:
                for b in [words] + uaux + [p for p in planes if p is not None]:
                    b.free()
                eng.word_bits = 32
                eng.drop_spare_tuned()           # (the block chosen for the 16-bit words is of a size nothing asks for again: back to the runtime)
                i = first
                continue
            lc_all = LC[0] if len(LC) == 1 else np.concatenate(LC)
            uaux[1].free(); uaux[2].free()
            yield first, ResidentBatch(planes=planes + [uaux[0]], words=words, n_slots=slots, n_ustart=slots + n_loc + 1,
                                       loci=lc_all, chrom=chroms, pos=poss[0] if len(poss) == 1 else np.concatenate(poss) if poss else np.zeros(0, np.int64),
                                       ref=refs, alleles=tables,
                                       n_device_runs=n_dev, n_host_runs=n_host)
    finally:
        bam.close()


def build_run(A, L, eng, cp, params, chrom, lo, fasta, run_ref, planes, uaux, slot_base, umi_base, cap, max_depth,
              allele_key, barcode_name, stream_sync=True, sampler: str = "reference", sampler_seed: int = 0, barcode_idents=None):
    """smc_build_planes over one run's alignments `A` (bamio.NativeBam.alignments_run or synth.generate_alignments) into the
    batch's device arrays (`planes`: [words, meta, umi, frag, dist], any of them None).  `allele_key(ai, qpos, indel)` / `barcode_name(gid)` give the texts the host needs (indel allele
    keys, barcode names for the reference's down-sampling)."""
    import time
    from .engine import DevBuf
    T = _TIMES if os.environ.get("SMC_DEVPLANES_TIMING") else None
    t1 = time.perf_counter()
    nl, ns = A["nl"], A["n_slots"]
    if nl == 0:
        return 0, 0, np.zeros(0, LOCUS_DTYPE), []
    deepest = int(A["loc"]["n"].max()) if len(A["loc"]) else 0
    # (status bit 1 - an alignment flagged neither READ1 nor READ2 - no longer sends the run to the host builder: the walk's
    # exact path takes the previous pileup read's pairOrder, smCounter.py:359-362)
    if (A["status"] & ~1) != 0 or deepest > max_depth or slot_base + ns > cap or \
            umi_base + ns + nl + 1 > cap + 8192:
        return None
    if (A["status"] & 1) and len(A["aln"]):
        # unflagged alignments: the walk looks back for the previous flagged pileup read of every such (read, locus) - fine for a
        # sprinkle of them, quadratic for single-end data, and a run that BEGINS with one ends in the host builder's error anyway
        unfl = (A["aln"]["oflag"] & 3) == 0
        if bool(unfl[0]) or float(unfl.mean()) > 0.25:
            return None
    up = lambda a: DevBuf(eng, a.nbytes + 256).upload(a.view(np.uint8).reshape(-1) if a.nbytes else np.zeros(4, np.uint8))
    d_aln, d_cig, d_bq, d_loc = up(A["aln"]), up(A["cig"]), up(A["bq"]), up(A["loc"])
    d_ref = up(np.frombuffer(run_ref[:nl].encode().ljust(nl, b"\0"), np.uint8).copy())
    d_loci = DevBuf(eng, nl * LOCUS_DTYPE.itemsize)
    xcap = 4 * nl + 4096
    d_x = DevBuf(eng, 20 * xcap)
    d_cnt = DevBuf(eng, 8)
    loc_host = np.ascontiguousarray(A["loc"])          # (the windows size the sort and the launch grids: read on the host)
    bi = abi.SmcBuildIn(d_aln.data_ptr(), d_cig.data_ptr(), d_bq.data_ptr(), d_loc.data_ptr(), d_ref.data_ptr(),
                        lo, nl, A["n_bc"], A["n_pair"], deepest, len(A["aln"]), loc_host.ctypes.data)
    pp = [t.data_ptr() if t is not None else None for t in planes]     # [words, meta, umi, frag, dist]
    w16 = getattr(planes[0], "word_bits", 32) == 16
    if w16:                                                            # (16-bit read words: nothing but the words is written)
        assert all(p is None for p in pp[1:])
        _lib.check(L.smc_build_planes_w16(eng.ctx, ctypes.byref(cp), ctypes.byref(bi), slot_base, umi_base, pp[0], uaux[0].data_ptr(),
                                          uaux[1].data_ptr(), uaux[2].data_ptr(), d_loci.data_ptr(), d_x.data_ptr(), xcap,
                                          d_cnt.data_ptr(), ctypes.c_void_p(0)), "smc_build_planes_w16")
    else:
        _lib.check(L.smc_build_planes(eng.ctx, ctypes.byref(cp), ctypes.byref(bi), slot_base, umi_base, pp[0], pp[1], pp[2], pp[3], pp[4],
                                      uaux[0].data_ptr(),
                                      uaux[1].data_ptr(), uaux[2].data_ptr(), d_loci.data_ptr(), d_x.data_ptr(), xcap,
                                      d_cnt.data_ptr(), ctypes.c_void_p(0)), "smc_build_planes")
    t2 = time.perf_counter()
    cnt = d_cnt.download(np.uint32, 2)        # (a copy on the default stream: behind the kernel)
    t3 = time.perf_counter()
    if T is not None:
        T["upload+launch"] += t2 - t1; T["kernel (sync)"] += t3 - t2
    if int(cnt[1]) != 0 or int(cnt[0]) > xcap:
        if w16 and (int(cnt[1]) & ~BUILD_ST_NARROW) == 0 and int(cnt[0]) <= xcap:
            return NARROW                     # (an allele id beyond 15 or a quality beyond 63: the caller builds with 32-bit words)
        if int(cnt[1]) & 4:
            from .features import PileupError
            raise PileupError("base quality > 126 at %s:%d-%d" % (chrom, lo + 1, lo + nl))
        return None
    lc = d_loci.download(LOCUS_DTYPE, nl)
    if sampler == "philox" and params.ds > 0 and bool((lc["n_umi"] > params.ds).any()):
        # the NON-parity down-sampling of the loci over the barcode cap, on the device (smc_philox_marks: Philox4x32-10 keyed by
        # position): marks in umi_start + SMC_LF_SAMPLED in the descriptors, where the reference-exact sampling below leaves its own
        d_pos = DevBuf(eng, 8 * nl + 256).upload(np.arange(lo + 1, lo + 1 + nl, dtype=np.int64))
        d_st = DevBuf(eng, 256).upload(np.zeros(4, np.uint32))
        # (what identifies a barcode is a hash of its TEXT, by the run-wide id the builder left in u_gid at these loci: the sample does
        # not depend on where the runs and batches of the file were cut)
        idents = barcode_idents(A["n_bc"]) if barcode_idents is not None else \
            np.array([_fnv64(barcode_name(g)) for g in range(int(A["n_bc"]))], np.uint64)
        d_id = DevBuf(eng, 8 * max(1, len(idents)) + 256).upload(idents)
        _lib.check(L.smc_philox_marks(eng.ctx, ctypes.byref(cp), d_loci.data_ptr(), nl, d_pos.data_ptr(), planes[0].data_ptr(),
                                      16 if w16 else 32, uaux[0].data_ptr(), d_id.data_ptr(), uaux[1].data_ptr(),
                                      ctypes.c_uint64(int(sampler_seed) & 0xFFFFFFFFFFFFFFFF), d_st.data_ptr(), ctypes.c_void_p(0)), "smc_philox_marks")
        lc = d_loci.download(LOCUS_DTYPE, nl)
        d_pos.free(); d_st.free(); d_id.free()
    for b in (d_aln, d_cig, d_bq, d_loc, d_ref, d_loci, d_cnt):
        b.free()                              # (back to the engine's spare list right away, not whenever the collector gets to them)
    # allele tables: the six fixed keys + what the kernel met, in the order it numbered them
    tables = [_BASE_TABLE] * nl                 # (shared, never written: a locus that met more alleles gets its own list)
    nx = int(cnt[0])
    if nx:
        xl = d_x.download(np.uint32, 5 * nx).reshape(nx, 5)
        xl = xl[np.lexsort((xl[:, 1], xl[:, 0]))]
        for l, aid, ai, qpos, indel in xl.tolist():
            key = allele_key(ai, qpos, np.int32(np.uint32(indel)))
            if key[0] == "D" and len(key) > 1:        # "D<len>|<site>" -> DEL|site+deleted|site (smCounter.py:392-396)
                ln, site = key[1:].split("|")
                pos1 = lo + l + 1
                key = "DEL|" + site + fasta.fetch(chrom, pos1, pos1 + int(ln)).upper() + "|" + site
            if tables[l] is _BASE_TABLE:
                tables[l] = list(BASE_ALLELES)
            assert aid == len(tables[l]), (l, aid, len(tables[l]))
            tables[l].append(key)
    d_x.free()
    # the reference's down-sampling (smCounter.py:496-498) on loci over the barcode cap: barcode texts by first included read
    over = np.nonzero((lc["n_umi"] > params.ds) & ((lc["flags"] & LF_SAMPLED) == 0))[0] if params.ds > 0 else []
    if len(over):
        from .py2compat import py2_downsample_barcodes
        for l in over.tolist():
            o, nu = int(lc["umi_off"][l]), int(lc["n_umi"][l])
            us = uaux[0].download(np.uint32, nu, 4 * o)
            gid = uaux[1].download(np.uint32, nu, 4 * o)
            finc = uaux[2].download(np.uint32, nu, 4 * o)
            keys = np.nonzero(finc != 0xFFFFFFFF)[0]
            keys = keys[np.argsort(finc[keys], kind="stable")]
            if len(keys) <= params.ds:
                continue
            names = [barcode_name(int(gid[u])) for u in keys]
            kept = set(py2_downsample_barcodes(str(lo + l + 1), names, params.ds))
            for u, name in zip(keys.tolist(), names):
                if name not in kept:
                    us[u] |= USTART_DROPPED
            uaux[0].upload(us, 4 * o)
            lc["flags"][l] |= LF_SAMPLED
    return nl, ns, lc, tables


def _fnv64(text: str) -> int:
    """FNV-1a, 64 bits, of a barcode's text (as smc_bam_barcode_idents computes it natively)."""
    x = 1469598103934665603
    for c in text.encode():
        x = ((x ^ c) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return x


def synth_allele_key(A):
    """`allele_key` for a run of synth.generate_alignments (what smc_bam_allele_key answers for a decoded run)."""
    def key(ai, qpos, indel):
        a = A["aln"][int(ai)]
        so, ls = int(a["seq_off"]), int(a["l_seq"])
        site = chr(A["seq"][so + int(qpos)])
        indel = int(indel)
        if indel > 0:
            return "INS|%s|%s%s" % (site, site, A["seq"][so + int(qpos) + 1:so + min(ls, int(qpos) + 1 + indel)].tobytes().decode())
        if indel < 0:
            return "D%d|%s" % (-indel, site)
        return site
    return key


def resident_from_alignments(A, eng, params, all_planes: bool = True, chrom: str = None, sampler: str = "reference",
                             sampler_seed: int = 0) -> ResidentBatch:
    """One run of synthetic alignments (synth.generate_alignments) -> a ResidentBatch built by smc_build_planes; raises when
    the device path does not take the run."""
    from . import synth
    from .engine import DevBuf
    L = eng.L
    chrom = chrom or synth.ALN_CHROM
    nl, ns, lo = A["nl"], A["n_slots"], int(A["start0"])
    cap = ns + 64
    planes = [DevBuf(eng, 4 * cap) if all_planes else None for k in range(4)]
    uaux = [DevBuf(eng, 4 * (cap + nl + 8192)) for _ in range(3)]
    run_ref = synth.aln_ref_fetch(lo, lo + nl)
    ref = synth.CyclicRef()
    for bits in ((16, 32) if (not all_planes and eng.word_bits == 16 and 0 <= params.minBQ <= 63) else (32,)):
        words = DevBuf(eng, (bits // 8) * cap, walk_output=True)
        words.word_bits = bits
        done = build_run(A, L, eng, abi.c_params(params), params, chrom, lo, ref, run_ref, [words] + planes, uaux, 0, 0, cap + nl,
                         L.smc_build_max_depth(), synth_allele_key(A), lambda gid: "B%d" % gid, sampler=sampler, sampler_seed=sampler_seed)
        if done != NARROW:
            break
        words.free()
    if done is None:
        raise RuntimeError("smc_build_planes did not take the run (status / size)")
    nl, ns, lc, tb = done
    uaux[1].free(); uaux[2].free()
    return ResidentBatch(planes=planes + [uaux[0]], words=words, n_slots=ns, n_ustart=ns + nl + 1, loci=lc, chrom=[chrom] * nl,
                         pos=np.arange(lo + 1, lo + 1 + nl, dtype=np.int64), ref=list(run_ref), alleles=tb, n_device_runs=1)


def DevLoci(eng, loci: np.ndarray):
    """The descriptors of a batch uploaded as a device array (for engine.make_plan_dev)."""
    from .engine import DevBuf
    loci = np.ascontiguousarray(loci)
    return DevBuf(eng, max(32, loci.nbytes)).upload(loci.view(np.uint8).reshape(-1))
