"""Host driver of the HIP path: one `Engine` per process / GPU.

Mirrors the reference's dispatch (smCounter.py:683-685: one task per locus, results in input
order) as one batched launch over an SoA batch.  PyTorch is used only to own device memory and
streams; every compute call goes through the C ABI (include/smcounter_hip.h).
"""
from __future__ import annotations

import ctypes

import numpy as np

from . import _lib, abi
from .features import DeviceBatch
from .params import VcParams


class DevBuf(object):
    """A device allocation owned through the C ABI (smc_mem_*): what the command line uses instead of a torch tensor, so
    that a single-process run never imports PyTorch.  Quacks like one where the bindings need it (`data_ptr()`)."""

    def __init__(self, eng, nbytes: int, walk_output: bool = False):
        """`walk_output`: the array is one the plane builder's walk writes (the read words of a batch).  Which physical pages hold
        it moves that kernel by up to 10 %, so a block of 256 MB or more is chosen among `eng.alloc_tries`
        allocations by the library's write-pattern probe (smc_mem_alloc_best, a few milliseconds per candidate, once: the block
        then serves every later batch of the process through the engine's spare list)."""
        # Sizes are rounded up to {1, 1.25, 1.5, 1.75} x 2^k and a freed allocation waits in the engine for the next request
        # of its size class (a batch allocates and frees some twenty buffers; the runtime's free synchronises the device)
        self.eng, self.nbytes = eng, int(nbytes)
        self.word_bits = 32          # (an array of read words: 32, or 16 once smc_build_planes_w16 has filled it - Plan.run_words looks)
        self.cls = _size_class(self.nbytes)
        self.tuned = bool(walk_output) and eng.alloc_tries > 1 and self.cls >= TUNED_MIN_BYTES
        spare = (eng._spare_tuned if self.tuned else eng._spare).get(self.cls)
        if spare:
            self.ptr = spare.pop()
            return
        p = ctypes.c_void_p()
        if self.tuned:
            info = (ctypes.c_float * 3)()
            _lib.check(eng.L.smc_mem_alloc_best(eng.ctx, self.cls, eng.alloc_tries, ctypes.byref(p), info), "smc_mem_alloc_best")
            eng.alloc_log.append({"bytes": self.cls, "probe_ms_kept": round(float(info[0]), 3), "probe_ms_slowest": round(float(info[1]), 3),
                                  "candidates": int(info[2])})
        else:
            _lib.check(eng.L.smc_mem_alloc(eng.ctx, self.cls, ctypes.byref(p)), "smc_mem_alloc")
        self.ptr = p.value or 0

    def data_ptr(self) -> int:
        return self.ptr

    def upload(self, arr: np.ndarray, offset_bytes: int = 0):
        arr = np.ascontiguousarray(arr)
        assert offset_bytes + arr.nbytes <= self.nbytes
        _lib.check(self.eng.L.smc_mem_h2d(self.eng.ctx, self.ptr + offset_bytes, arr.ctypes.data, arr.nbytes), "smc_mem_h2d")
        return self

    def download(self, dtype, count: int, offset_bytes: int = 0, out: np.ndarray = None) -> np.ndarray:
        if out is None:
            out = np.empty(count, dtype)
        assert out.dtype == np.dtype(dtype) and len(out) == count and out.flags.c_contiguous
        assert offset_bytes + out.nbytes <= self.nbytes
        _lib.check(self.eng.L.smc_mem_d2h(self.eng.ctx, out.ctypes.data, self.ptr + offset_bytes, out.nbytes), "smc_mem_d2h")
        return out

    def view(self, offset_bytes: int):
        """A pointer into the allocation (no ownership); an array of read words keeps its width."""
        return _DevView(self.ptr + offset_bytes, self.word_bits)

    def free(self):
        if self.ptr and self.eng.ctx:
            (self.eng._spare_tuned if self.tuned else self.eng._spare).setdefault(self.cls, []).append(self.ptr)
        self.ptr = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


TUNED_MIN_BYTES = 256 << 20      # (smc_mem_alloc_best's own threshold)


def _size_class(n: int) -> int:
    n = max(256, int(n))
    k = n.bit_length() - 1                    # 2^k <= n
    q = 1 << max(0, k - 2)                    # quarter steps
    return (n + q - 1) // q * q


class _DevView(object):
    def __init__(self, ptr, word_bits: int = 32):
        self.ptr = ptr
        self.word_bits = word_bits       # (Plan.run_words picks the kernels by it: a view must not lose it - ADVICE r5)

    def data_ptr(self) -> int:
        return self.ptr


class Engine(object):
    def __init__(self, device: int = 0):
        self.L = _lib.load()
        if self.L.smc_device_count() <= 0:
            raise _lib.SmcError("no HIP device visible; the smCounter HIP path needs an MI355X (gfx950)")
        self.device = device
        h = ctypes.c_void_p()
        _lib.check(self.L.smc_create(device, ctypes.byref(h)), "smc_create")
        self.ctx = h
        self._spare = {}                      # size class -> freed DevBuf pointers, reused before allocating anew
        self._spare_tuned = {}                # ... of the blocks chosen by the write-pattern probe (DevBuf(walk_output=True))
        import os
        # candidates per such block (1: take what comes).  About one allocation in eight is of the fast kind; the search stops at
        # the first it finds, a candidate costs ~ 5 ms of probing + its allocation (1-80 ms for 2.4 GB)
        self.alloc_tries = int(os.environ.get("SMC_ALLOC_TRIES", "12"))
        # the read words between the plane builder and the locus kernels: 16 bits each (smc_read_word16) until a run has an allele
        # id beyond 15 or a base quality beyond 63 - from then on 32 (devplanes.py; SMC_WORD_BITS=32: from the start)
        self.word_bits = 32 if os.environ.get("SMC_WORD_BITS", "16") == "32" else 16
        self.alloc_log = []                   # what the probe saw, per chosen block
        self._pinned = {}                     # name -> (pointer, bytes) of page-locked staging memory
        self.last_rows = None                 # rows of the last vc.vc_resident call (pinned staging: valid until the next)

    def pinned(self, name: str, dtype, count: int) -> np.ndarray:
        """A numpy array over page-locked host memory owned by the engine, one per `name`, grown when needed: staging for the
        copies of a run (valid until the next request under the same name)."""
        dtype = np.dtype(dtype)
        need = max(1, int(count)) * dtype.itemsize
        slot = self._pinned.get(name)
        if slot is None or slot[1] < need:
            if slot is not None:
                self.L.smc_mem_free_host(self.ctx, slot[0])
            size = _size_class(need + need // 4)
            p = ctypes.c_void_p()
            _lib.check(self.L.smc_mem_alloc_host(self.ctx, size, ctypes.byref(p)), "smc_mem_alloc_host")
            slot = self._pinned[name] = (p.value, size)
        raw = (ctypes.c_ubyte * need).from_address(slot[0])
        return np.frombuffer(raw, dtype, int(count))

    def trim(self):
        """Give the cached device buffers (DevBuf) and the context's pool of plan blocks back to the runtime."""
        if self.ctx:
            self.L.smc_pool_trim(self.ctx)
            for ptrs in list(self._spare.values()) + list(self._spare_tuned.values()):
                for p in ptrs:
                    self.L.smc_mem_free(self.ctx, p)
        self._spare = {}
        self._spare_tuned = {}

    def drop_spare_tuned(self):
        """Give the spare blocks that the write-pattern probe chose (DevBuf(walk_output=True)) back to the runtime: a file that has
        fallen back to 32-bit read words never asks for the 16-bit block's size again."""
        if self.ctx:
            for ptrs in self._spare_tuned.values():
                for p in ptrs:
                    self.L.smc_mem_free(self.ctx, p)
        self._spare_tuned = {}

    def close(self):
        if self.ctx:
            self.trim()
            for p, _ in self._pinned.values():
                self.L.smc_mem_free_host(self.ctx, p)
            self._pinned = {}
            self.L.smc_destroy(self.ctx)
            self.ctx = None

    # ---- pure C-ABI path: host buffers in, host rows out
    def call_batch_host(self, db: DeviceBatch, params: VcParams) -> np.ndarray:
        rows = np.zeros(db.n_loci, abi.ROW_DTYPE)
        cp = abi.c_params(params)
        loci = np.ascontiguousarray(db.loci)
        planes = [np.ascontiguousarray(x, np.uint32) for x in (db.meta, db.umi, db.frag, db.dist)]
        us = np.ascontiguousarray(db.umi_start, np.uint32)
        _lib.check(self.L.smc_call_batch_host(
            self.ctx, ctypes.byref(cp), loci.ctypes.data, db.n_loci,
            planes[0].ctypes.data, planes[1].ctypes.data, planes[2].ctypes.data, planes[3].ctypes.data,
            db.n_slots, us.ctypes.data, len(us), rows.ctypes.data), "smc_call_batch_host")
        return rows

    # ---- resident path: the four planes + umi_start live in HBM (torch tensors), a plan is reused
    def upload(self, db: DeviceBatch):
        import torch
        dev = torch.device("cuda", self.device)
        return [torch.from_numpy(np.ascontiguousarray(x).view(np.int32)).to(dev)
                for x in (db.meta, db.umi, db.frag, db.dist, db.umi_start)]

    def make_plan(self, loci: np.ndarray):
        loci = np.ascontiguousarray(loci)
        h = ctypes.c_void_p()
        _lib.check(self.L.smc_plan_create(self.ctx, loci.ctypes.data, len(loci), ctypes.byref(h)),
                   "smc_plan_create")
        return Plan(self, h, len(loci))


    def make_plan_dev(self, d_loci, n_loci: int, stream=None, spec_params: VcParams = None):
        """The plan of a batch whose descriptors are in HBM (`d_loci`: DevBuf / tensor of smc_locus, e.g. what smc_build_planes
        wrote): binned on the device (smc_plan_create_dev), default stream.  `d_loci` must outlive the plan.
        `spec_params` (the parameters the planes were built with): smc_plan_create_dev_spec - nothing waits for the device, the
        launches are sized from the context's last plan; the caller asks `plan.ok()` once it has the rows and, on False, makes the
        plan again and re-runs it."""
        h = ctypes.c_void_p()
        sp = ctypes.c_void_p(stream.cuda_stream if stream is not None else 0)     # (a torch stream, or the default one)
        if spec_params is not None:
            cp = abi.c_params(spec_params)
            _lib.check(self.L.smc_plan_create_dev_spec(self.ctx, ctypes.byref(cp), d_loci.data_ptr(), int(n_loci), sp, ctypes.byref(h)),
                       "smc_plan_create_dev_spec")
        else:
            _lib.check(self.L.smc_plan_create_dev(self.ctx, d_loci.data_ptr(), int(n_loci), sp, ctypes.byref(h)),
                       "smc_plan_create_dev")
        return Plan(self, h, int(n_loci))

    def reset_plan_hint(self):
        """The next plan made with `spec_params` goes the exact way (smc_plan_hint_reset): before batches of another kind."""
        _lib.check(self.L.smc_plan_hint_reset(self.ctx), "smc_plan_hint_reset")

    def spec_counts(self):
        """(plans made without the host, of those made the exact way, found not to fit on the device) - smc_plan_spec_counts."""
        a, b, c = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
        _lib.check(self.L.smc_plan_spec_counts(self.ctx, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)), "smc_plan_spec_counts")
        return a.value, b.value, c.value


class Plan(object):
    def __init__(self, eng: Engine, handle, n_loci: int):
        self.eng, self.h, self.n_loci = eng, handle, n_loci

    def ok(self) -> bool:
        """After the rows have been waited for: are they this batch's (always, for a plan made the exact way)?  smc_plan_spec_ok."""
        v = ctypes.c_int(1)
        _lib.check(self.eng.L.smc_plan_spec_ok(self.h, ctypes.byref(v)), "smc_plan_spec_ok")
        return bool(v.value)

    def info(self):
        nl, sb = ctypes.c_int32(), ctypes.c_int64()
        _lib.check(self.eng.L.smc_plan_info(self.h, ctypes.byref(nl), ctypes.byref(sb)), "smc_plan_info")
        return nl.value, sb.value

    def alloc_rows(self):
        import torch
        return torch.empty(self.n_loci * abi.ROW_DTYPE.itemsize, dtype=torch.uint8,
                           device=torch.device("cuda", self.eng.device))

    def _stream_ptr(self, stream):
        if stream == 0:
            return ctypes.c_void_p(0)
        import torch
        st = stream if stream is not None else torch.cuda.current_stream(self.eng.device)
        return ctypes.c_void_p(st.cuda_stream)

    def pack_words(self, meta, frag, words, stream=None):
        """meta + frag planes of the plan's batch -> read words (smc_pack_words; one uint32 per read, what the kernels read)."""
        _lib.check(self.eng.L.smc_pack_words(self.h, meta.data_ptr(), frag.data_ptr(), words.data_ptr(), self._stream_ptr(stream)),
                   "smc_pack_words")
        return words

    def run_words(self, words, umi_start, params: VcParams, rows=None, stream=None):
        """The hot path on read words (smc_plan_run_words): what smc_build_planes writes, or pack_words has made."""
        if rows is None:
            rows = self.alloc_rows()
        cp = abi.c_params(params)
        # (the width travels WITH the array - DevBuf / its views / bench wrappers carry `word_bits`; a torch tensor of 32-bit words has
        # none.  A wrapper of 16-bit words that dropped it would run the 32-bit kernels on them: anything that is not a tensor must say)
        if not hasattr(words, "word_bits") and not hasattr(words, "dtype"):
            raise _lib.SmcError("Plan.run_words: %r does not say how wide its read words are (word_bits)" % type(words).__name__)
        if getattr(words, "word_bits", 32) == 16:                     # (smc_read_word16: what smc_build_planes_w16 has written)
            _lib.check(self.eng.L.smc_plan_run_words16(self.h, ctypes.byref(cp), words.data_ptr(), umi_start.data_ptr(), rows.data_ptr(),
                                                       self._stream_ptr(stream)), "smc_plan_run_words16")
            return rows
        _lib.check(self.eng.L.smc_plan_run_words(self.h, ctypes.byref(cp), words.data_ptr(), umi_start.data_ptr(), rows.data_ptr(),
                                                 self._stream_ptr(stream)), "smc_plan_run_words")
        return rows

    def run(self, planes, params: VcParams, rows=None, stream=None):
        """Enqueue the hot path on `stream` (a torch.cuda.Stream; default: torch's current one; 0: the default HIP stream,
        without touching torch - the DevBuf path).  `planes`: [meta, umi, frag, dist, umi_start] (the raw-field planes: packed
        into read words first, smc_plan_run) or [words, umi_start]."""
        if len(planes) == 2:
            return self.run_words(planes[0], planes[1], params, rows, stream)
        st_ptr = self._stream_ptr(stream)
        if rows is None:
            rows = self.alloc_rows()
        cp = abi.c_params(params)
        ptr = lambda t: t.data_ptr() if t is not None else None     # (umi / dist may be absent: the kernels do not read them)
        _lib.check(self.eng.L.smc_plan_run(self.h, ctypes.byref(cp), ptr(planes[0]), ptr(planes[1]),
                                           ptr(planes[2]), ptr(planes[3]), ptr(planes[4]),
                                           rows.data_ptr(), st_ptr), "smc_plan_run")
        return rows

    def run_devbuf(self, planes, params: VcParams) -> np.ndarray:
        """The torch-free path: planes are DevBuf / views, the rows come back as a numpy array (synchronous) - over the engine's
        page-locked staging memory: valid until the next run_devbuf on this engine."""
        rows = DevBuf(self.eng, self.n_loci * abi.ROW_DTYPE.itemsize)
        self.run(planes, params, rows, stream=0)
        # (hipMemcpy on the default stream: ordered behind the kernels)
        out = rows.download(abi.ROW_DTYPE, self.n_loci, out=self.eng.pinned("rows", abi.ROW_DTYPE, self.n_loci))
        rows.free()
        return out

    def alloc_wire(self):
        """Device buffer for the packed wire rows of this plan's loci (abi.WIRE_DTYPE, 168 B per locus)."""
        import torch
        return torch.empty(self.n_loci * abi.WIRE_DTYPE.itemsize, dtype=torch.uint8,
                           device=torch.device("cuda", self.eng.device))

    def pack(self, rows, wire=None, stream=None):
        """Enqueue k_pack_rows: rows -> wire rows (what the gather to the writing rank moves)."""
        import torch
        if wire is None:
            wire = self.alloc_wire()
        st = stream if stream is not None else torch.cuda.current_stream(self.eng.device)
        _lib.check(self.eng.L.smc_pack_rows(self.eng.ctx, rows.data_ptr(), self.n_loci, wire.data_ptr(),
                                            ctypes.c_void_p(st.cuda_stream)), "smc_pack_rows")
        return wire

    @staticmethod
    def download_wire(wire) -> np.ndarray:
        return wire.cpu().numpy().view(abi.WIRE_DTYPE)

    def set_timing(self, slots: int):
        """Keep HIP-event pairs around the dominant kernel of the next `slots` runs (0 = off)."""
        _lib.check(self.eng.L.smc_plan_set_timing(self.h, int(slots)), "smc_plan_set_timing")

    def kernel_ms(self):
        """(mean ms, samples, loci, reads) of the dominant k_call_v2 launch over the timed runs."""
        ms, ns, nl, nr = ctypes.c_float(), ctypes.c_int32(), ctypes.c_int64(), ctypes.c_int64()
        _lib.check(self.eng.L.smc_plan_kernel_ms(self.h, ctypes.byref(ms), ctypes.byref(ns), ctypes.byref(nl),
                                                 ctypes.byref(nr)), "smc_plan_kernel_ms")
        return ms.value, ns.value, nl.value, nr.value

    @staticmethod
    def download(rows) -> np.ndarray:
        return rows.cpu().numpy().view(abi.ROW_DTYPE)

    def close(self):
        if self.h:
            self.eng.L.smc_plan_destroy(self.h)
            self.h = None
