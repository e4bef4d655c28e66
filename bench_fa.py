"""bench.py's step: the hot path timed from where the reference's hot loop starts (smCounter.py:316) - a run's ALIGNMENTS
resident in HBM (what the BAM decoder hands over) -> smc_build_planes (sort, count, scan, the walk that writes the read words:
csrc/k_build_planes.inc, k_bp_emit2.inc) -> smc_plan_create_dev (launch plan made where the descriptors are) ->
smc_plan_run_words -> rows in HBM.

A measurement harness beside bench.py (not part of the package): the resident run, its step, and the parity pass - EVERY locus of
the run against oracle/aln_planes.c (the reference's per-pileup-read logic from the same alignments, on the host cores) +
oracle/smc_oracle.c.  `python3 -m bench_fa` runs the leg alone (what the rocprofv3 scripts profile).
"""
from __future__ import annotations

import ctypes
import dataclasses
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from smcounter_amd import _lib, abi, synth          # noqa: E402
from smcounter_amd.engine import DevBuf             # noqa: E402
from smcounter_amd.features import LOCUS_DTYPE      # noqa: E402

HBM_PEAK_GBS = 8000.0


class AlignmentRun(object):
    """A run of synthetic alignments resident in HBM + the output arrays of smc_build_planes (only what the locus kernels read:
    the read words).  `shard`: which stretch of the seeded config this rank takes (its loci start n_loci * shard further on,
    its molecules are drawn from another seed) - the weak-scaling input of rank `shard`."""

    def __init__(self, eng, cfg, params, n_loci, nthreads, shard: int = 0, slots: int = 1, place: int = 0):
        """(The read words of every slot are a block the LIBRARY chose: engine.DevBuf(walk_output=True) -> smc_mem_alloc_best.)
        `slots` > 1: consecutive steps alternate between that many sets of output arrays, each with a stream of its own - the
        builder of step i + 1 is then enqueued while the locus kernels of step i still run (the host waits, inside
        smc_plan_create_dev, only for the builder of the step it is issuing), as the runs of a BAM follow each other.
        `place` > 0: WHICH ALLOCATION holds the read words moves the walk's time by up to 10 % (1.33 / 1.38 / 1.46 ms
        for three allocations of the same 2.4 GB in one process, the same from launch to launch; offsets inside an allocation do not
        matter).  The arrays of a resident run are allocated once and used step after step, so the run tries `place` more
        allocations (behind spacer allocations of different sizes: allocations made back to back tend to be alike), times the walk
        into each - set-up, outside every timed region - and keeps the fastest."""
        self.eng, self.params = eng, params
        eng.reset_plan_hint()                    # (a run of its own kind: its first plan is made the exact way, the others are sized from it)
        if shard:
            cfg = dataclasses.replace(cfg, seed=cfg.seed + 7919 * shard, start_pos=cfg.start_pos + n_loci * shard)
        self.cfg = cfg
        t0 = time.time()
        kw = {}
        if os.environ.get("SMC_FA_INDEL_RATE"):                              # (experiments: the share of alignments with an insertion / a deletion)
            kw = dict(p_ins_aln=float(os.environ["SMC_FA_INDEL_RATE"]), p_del_aln=float(os.environ["SMC_FA_INDEL_RATE"]))
        A = synth.generate_alignments(cfg, n_loci, params, nthreads=nthreads, **kw)
        self.t_gen = time.time() - t0
        self.A = A
        self.nl, self.ns, self.lo = A["nl"], A["n_slots"], int(A["start0"])
        self.reads = int(A["reads"])
        up = lambda a: DevBuf(eng, a.nbytes + 256).upload(a.view(np.uint8).reshape(-1))
        self.d_in = [up(A[k]) for k in ("aln", "cig", "bq", "loc")]
        run_ref = synth.aln_ref_fetch(self.lo, self.lo + self.nl)
        self.d_ref = up(np.frombuffer(run_ref.encode(), np.uint8).copy())
        self.xcap = 4 * self.nl + 4096
        self.slots = []
        # the read words between the walk and the locus kernels: 16 bits each (smc_read_word16) unless the engine has been told
        # otherwise (SMC_WORD_BITS=32) or the run has no room in them (checked below: one build, then 32-bit words)
        self.word_bits = 16 if (eng.word_bits == 16 and 0 <= params.minBQ <= 63) else 32
        for k in range(max(1, slots)):
            S = {"words": self._words_buf(), "uaux": [DevBuf(eng, 4 * (self.ns + self.nl + 64)) for _ in range(3)],
                 "d_loci": DevBuf(eng, self.nl * LOCUS_DTYPE.itemsize), "d_x": DevBuf(eng, 20 * self.xcap), "d_cnt": DevBuf(eng, 8),
                 "rows": DevBuf(eng, self.nl * abi.ROW_DTYPE.itemsize), "stream": None}
            if slots > 1:
                import torch
                S["stream"] = torch.cuda.Stream(device=eng.device)
            self.slots.append(S)
        self.k = 0
        S0 = self.slots[0]       # (what the callers that time the locus kernels alone, or check the rows, look at)
        self.words, self.uaux, self.d_loci, self.d_x, self.d_cnt = S0["words"], S0["uaux"], S0["d_loci"], S0["d_x"], S0["d_cnt"]
        self.loc_host = np.ascontiguousarray(A["loc"])
        self.bi = abi.SmcBuildIn(self.d_in[0].data_ptr(), self.d_in[1].data_ptr(), self.d_in[2].data_ptr(), self.d_in[3].data_ptr(),
                                 self.d_ref.data_ptr(), self.lo, self.nl, A["n_bc"], A["n_pair"],
                                 int(A["loc"]["n"].max()), len(A["aln"]), self.loc_host.ctypes.data)
        self.cp = abi.c_params(params)
        self.rows = S0["rows"]
        self.lc = np.empty(self.nl, LOCUS_DTYPE)
        self.host_plan = bool(os.environ.get("SMC_FA_HOST_PLAN"))      # (measurement: descriptors back to the host, smc_plan_create)
        self.pace = not os.environ.get("SMC_FA_NO_PACE")
        self.exact_plans = bool(os.environ.get("SMC_FA_EXACT_PLANS"))   # (measurement: every plan waits for its batch's own record, round 5's way)
        self.t = {"build_issue": 0.0, "descriptors_d2h": 0.0, "plan_create": 0.0, "run_issue": 0.0, "n": 0}
        self.placement = None
        self.active_slots = len(self.slots)            # (slots the steps alternate between: _place may leave it at one)
        if self.word_bits == 16:
            self.step(slot=0)
            eng.L.smc_device_sync(eng.ctx)
            if self.status()[1] & 32:                   # (an allele id beyond 15 or a quality beyond 63 somewhere in the run)
                self.word_bits = 32
                for S in self.slots:
                    S["words"].free()
                    S["words"] = self._words_buf()
                self.words = self.slots[0]["words"]
        self.slot_choice = None
        if len(self.slots) > 1 and not os.environ.get("SMC_FA_NO_SLOT_CHOICE"):
            self._choose_slots()
        # (tests: one rank of a multi-rank bench steps through ONE slot while the others use two - its row buffers are then all filled
        # by one stream, which the gather has to order against: ADVICE r4)
        if os.environ.get("SMC_FA_ONE_SLOT_ON_RANK") is not None and os.environ.get("SMC_FA_ONE_SLOT_ON_RANK") == os.environ.get("RANK", "0"):
            self.active_slots = 1
        elif os.environ.get("SMC_FA_ONE_SLOT_ON_RANK") is not None and len(self.slots) > 1:
            self.active_slots = len(self.slots)
        if place > 0:
            try:
                self._place(place)
            except Exception as e:                       # (a failed trial allocation must not cost the run: take the arrays as they are)
                self.placement = {"note": "placement trials abandoned: %s" % e}
                eng.trim()
            self.words = self.slots[0]["words"]

    def _choose_slots(self, steps: int = 10):
        """Two steps in flight on two streams, or one after the other on one?  The second stream lets the builder of step i + 1 run
        beside the locus kernels of step i: 2-3 % on most boxes, minus 3-5 % on some (measured round 6 on four boxes: 2.56 vs 2.62,
        2.62 vs 2.76, 2.65 vs 2.73 - and 2.80 vs 2.66 ms per C3 step).  A pipeline picks its depth where it runs: a few steps of each
        at set-up, outside every timed region; the faster stays."""
        L, eng = self.eng.L, self.eng
        ms = {}
        for n in (1, len(self.slots)):                  # (one slot first: the pools of plan blocks are warm by the time two are timed)
            self.active_slots = n
            for _ in range(4):
                self.step()
            L.smc_device_sync(eng.ctx)
            t0 = time.perf_counter()
            for _ in range(steps):
                self.step()
            L.smc_device_sync(eng.ctx)
            ms[n] = (time.perf_counter() - t0) / steps * 1e3
        self.active_slots = min(ms, key=ms.get)
        self.slot_choice = {"ms_per_step_by_slots_in_flight": {str(k): round(v, 4) for k, v in ms.items()}, "kept": self.active_slots}
        for k in self.t:
            self.t[k] = 0

    def _words_buf(self):
        from smcounter_amd.engine import DevBuf
        w = DevBuf(self.eng, (self.word_bits // 8) * (self.ns + 64), walk_output=True)
        w.word_bits = self.word_bits
        return w

    def _walk_ms(self, reps: int = 3) -> float:
        eng, L = self.eng, self.eng.L
        self.step(slot=0)
        L.smc_device_sync(eng.ctx)
        _lib.check(L.smc_build_set_timing(eng.ctx, reps), "smc_build_set_timing")
        for _ in range(reps):
            self.step(slot=0)
        L.smc_device_sync(eng.ctx)
        k_ms, k_n = ctypes.c_float(), ctypes.c_int32()
        _lib.check(L.smc_build_kernel_ms(eng.ctx, ctypes.byref(k_ms), ctypes.byref(k_n)), "smc_build_kernel_ms")
        L.smc_build_set_timing(eng.ctx, 0)
        return float(k_ms.value)

    def _place(self, extra: int):
        eng = self.eng
        cands, spacers = [S["words"] for S in self.slots], []
        budget = 110e9                                # bytes the candidates and their spacers may take together (of 288 GB)
        try:
            import torch
            budget = min(budget, 0.6 * torch.cuda.mem_get_info(eng.device)[0])      # ... and of what is free now
        except Exception:
            pass
        for i in range(extra):
            mb = (37, 301, 1024, 2500, 150, 4097, 611, 1777)[i % 8]
            budget -= (mb << 20) + (self.word_bits // 8) * (self.ns + 64)
            if budget < 0:
                break
            try:
                spacers.append(DevBuf(eng, (mb << 20) + 4096))
                cands.append(self._words_buf())
            except Exception:                             # (memory is short: what has been allocated so far are the candidates)
                break
        ms = []
        for w in cands:
            self.slots[0]["words"] = w
            ms.append(self._walk_ms())
        order = sorted(range(len(cands)), key=lambda i: ms[i])
        for k, S in enumerate(self.slots):
            S["words"] = cands[order[k]]
        for i in order[len(self.slots):]:
            cands[i].free()
        for sp in spacers:
            sp.free()
        eng.trim()                                  # (the candidates that lost go back to the runtime, not to the engine's spare list)
        kept = [ms[i] for i in order[:len(self.slots)]]
        # two slots buy ~ 1 % (the next step's builder beside this step's locus kernels) - less than a second-class allocation for
        # the second slot costs: then every step goes through the first
        if len(kept) > 1 and kept[1] > 1.015 * kept[0]:
            self.active_slots = 1
        part_ms = {}
        self.placement = {"walk_ms_by_allocation": [round(x, 3) for x in ms], "kept": [round(x, 3) for x in kept], "slots_used": self.active_slots,
                          "step_ms_by_rows_per_part": {str(k): round(v, 3) for k, v in part_ms.items()},
                          "note": "which allocation holds the read words moves the walk's time (DESIGN.md section 8); candidates are timed "
                                  "at set-up, outside every timed region, and the fastest kept"}
        for k in self.t:
            self.t[k] = 0

    def input_bytes(self):
        A = self.A
        return int(A["aln"].nbytes + A["cig"].nbytes + A["bq"].nbytes + A["loc"].nbytes)

    def needed_bytes(self):
        """What the walk that writes the read words has to move per launch: per pileup read one base + one quality in and
        the word out; the alignment records, their row records (written once, read once per tile they touch - counted once)
        and CIGARs; umi_start and the descriptor per locus."""
        A = self.A
        return 2.0 * self.reads + (self.word_bits // 8) * float(self.ns) + (36.0 + 32.0) * len(A["aln"]) + float(A["cig"].nbytes) + 36.0 * self.nl

    def step(self, keep_plan=False, rows=None, slot=None):
        """build -> descriptors -> plan -> run; everything the product path does between the decoder and the rows."""
        eng, L = self.eng, self.eng.L
        if slot is None:
            slot = self.k
            self.k += 1
        slot %= self.active_slots
        S = self.slots[slot]
        self.last_slot = slot
        st = S["stream"]
        # (nothing in a step waits for the device any more - smc_plan_create_dev_spec -, so the host paces itself: a slot is issued
        # again only when its previous step is through; without that the host runs hundreds of steps ahead, the two streams' queues
        # interleave as they like and the pool of plan blocks is exhausted - measured: X3 0.73 -> 2.3 ms per step)
        if st is not None and self.pace:
            # (steps in flight: two - one per slot with two slots, the one running and the one queued behind it with one.  Measured
            # with two per slot on two slots: C3 3.5 ms, X3 2.4 ms per step)
            while len(S.setdefault("done", [])) >= (1 if self.active_slots > 1 else 2):
                S["done"].pop(0).synchronize()
        sp = ctypes.c_void_p(st.cuda_stream if st is not None else 0)
        words, uaux, d_loci = S["words"], S["uaux"], S["d_loci"]
        t0 = time.perf_counter()
        if self.word_bits == 16:
            _lib.check(L.smc_build_planes_w16(eng.ctx, ctypes.byref(self.cp), ctypes.byref(self.bi), 0, 0, words.data_ptr(), uaux[0].data_ptr(),
                                              uaux[1].data_ptr(), uaux[2].data_ptr(), d_loci.data_ptr(), S["d_x"].data_ptr(), self.xcap,
                                              S["d_cnt"].data_ptr(), sp), "smc_build_planes_w16")
        else:
            _lib.check(L.smc_build_planes(eng.ctx, ctypes.byref(self.cp), ctypes.byref(self.bi), 0, 0, words.data_ptr(), None, None,
                                          None, None, uaux[0].data_ptr(), uaux[1].data_ptr(),
                                          uaux[2].data_ptr(), d_loci.data_ptr(), S["d_x"].data_ptr(), self.xcap,
                                          S["d_cnt"].data_ptr(), sp), "smc_build_planes")
        t1 = time.perf_counter()
        if self.host_plan:
            d_loci.download(LOCUS_DTYPE, self.nl, out=self.lc)        # (behind the kernels on the default stream)
            t2 = time.perf_counter()
            plan = eng.make_plan(self.lc)
        else:
            t2 = t1
            # binned where the descriptors are; from the run's second step on without waiting for the builder (smc_plan_create_dev_spec:
            # the launches sized from the step before - the same run: they fit; run_leg / bench.py check the device's count of misfits)
            plan = eng.make_plan_dev(d_loci, self.nl, stream=st, spec_params=None if self.exact_plans else self.params)
        t3 = time.perf_counter()
        plan.run([words, uaux[0]], self.params, S["rows"] if rows is None else rows, stream=st if st is not None else 0)
        t4 = time.perf_counter()
        if st is not None and self.pace:
            import torch
            ev = torch.cuda.Event()
            ev.record(st)
            S["done"].append(ev)
        T = self.t
        T["build_issue"] += t1 - t0; T["descriptors_d2h"] += t2 - t1; T["plan_create"] += t3 - t2; T["run_issue"] += t4 - t3; T["n"] += 1
        if keep_plan:
            return plan
        plan.close()

    def serial_ms(self, steps: int, rows=None) -> float:
        """ms per step with every step on slot 0's stream, one after the other (the host still a step ahead)."""
        L, eng = self.eng.L, self.eng
        keep = self.active_slots
        self.active_slots = 1
        try:
            L.smc_device_sync(eng.ctx)
            t0 = time.perf_counter()
            for _ in range(steps):
                self.step(rows=rows, slot=0)
            L.smc_device_sync(eng.ctx)
            return (time.perf_counter() - t0) / steps * 1e3
        finally:
            self.active_slots = keep

    def measure_other_word_width(self, steps: int, rows=None):
        """The same run, one step at a time on slot 0, with the read words in the OTHER width (32 bits where the run uses 16): the
        walk's mean HIP-event time, the step's wall time and the walk's fraction of the roofline on ITS needed bytes - measured in
        the same process right after the headline, so that the two formats can be compared on one box and one set of inputs."""
        eng, L = self.eng, self.eng.L
        keep_bits, keep_words = self.word_bits, self.slots[0]["words"]
        self.word_bits = 32 if keep_bits == 16 else 16
        try:
            self.slots[0]["words"] = self._words_buf()
            for _ in range(2):
                self.step(rows=rows, slot=0)
            L.smc_device_sync(eng.ctx)
            _lib.check(L.smc_build_set_timing(eng.ctx, steps), "smc_build_set_timing")
            t0 = time.perf_counter()
            for _ in range(steps):
                self.step(rows=rows, slot=0)
            L.smc_device_sync(eng.ctx)
            ms = (time.perf_counter() - t0) / steps * 1e3
            k_ms, k_n = ctypes.c_float(), ctypes.c_int32()
            _lib.check(L.smc_build_kernel_ms(eng.ctx, ctypes.byref(k_ms), ctypes.byref(k_n)), "smc_build_kernel_ms")
            L.smc_build_set_timing(eng.ctx, 0)
            need = self.needed_bytes()
            out = {"read_word_bits": self.word_bits, "ms_per_step_one_at_a_time": ms, "k_bp_emit2_ms": float(k_ms.value),
                   "needed_bytes_per_launch": need, "frac": need / (float(k_ms.value) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                   "builder_status": self.status()}
            self.slots[0]["words"].free()
            return out
        finally:
            self.word_bits, self.slots[0]["words"] = keep_bits, keep_words

    def status(self):
        return self.d_cnt.download(np.uint32, 2).tolist()

    def stream_of(self, slot):
        return self.slots[slot]["stream"]

    def close(self):
        for b in self.d_in + [self.d_ref]:
            b.free()
        for S in self.slots:
            for b in [S["words"], S["d_loci"], S["d_x"], S["d_cnt"], S["rows"]] + S["uaux"]:
                if b is not None:
                    b.free()
        self.A = None


def parity_full(run: AlignmentRun, nthreads: int, n_loci: int = 0, chunk: int = 20000):
    """Rows of the run's loci (all of them, or `n_loci` spread as a first, a middle and a last stretch) against the CPU: the same
    alignments through oracle/aln_planes.c (the reference's pileup logic, smCounter.py:316-471) and oracle/smc_oracle.c
    (:26-98, :482-600), chunk by chunk on the host cores."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_lib
    got_all = run.rows.download(abi.ROW_DTYPE, run.nl)
    if n_loci and n_loci < run.nl:
        third = max(1, n_loci // 3)
        mid = (run.nl - third) // 2
        spans = [(0, third), (mid, mid + third), (run.nl - third, run.nl)]
    else:
        spans = [(0, run.nl)]
    tot = {"loci": 0, "mismatches": 0, "fragile_skipped": 0, "near_tie_skipped": 0, "underflow_skipped": 0, "pi_max_abs_diff": 0.0,
           "loci_filtered": 0, "fisher_tests_run": 0, "p_max_abs_diff": 0.0, "detail": []}
    t0 = time.time()
    for a, b in spans:
        for c0 in range(a, b, chunk):
            c1 = min(b, c0 + chunk)
            db = oracle_lib.aln_planes(run.A, run.params, c0, c1, n_threads=nthreads)
            want, fragile, pi_all = oracle_lib.call_batch_mt(db, abi.c_params(run.params), abi.ROW_DTYPE, nthreads, return_fragile=True,
                                                             return_pi_all=True)
            rep = abi.parity_report(got_all[c0:c1], want, fragile, pi_all)
            for k in ("loci", "mismatches", "fragile_skipped", "near_tie_skipped", "underflow_skipped", "loci_filtered", "fisher_tests_run"):
                tot[k] += rep[k]
            for k in ("pi_max_abs_diff", "p_max_abs_diff"):
                tot[k] = max(tot[k], rep[k])
            tot["detail"] += [("chunk at locus %d" % c0, d) for d in rep["detail"]]
    from smcounter_amd import rows as _rows
    tot["pi_boundary_loci"] = int(len(_rows.pi_boundary_loci(got_all)))
    tot["detail"] = tot["detail"][:3]
    tot["spans"] = spans
    tot["seconds"] = round(time.time() - t0, 1)
    tot["checked_against"] = ("oracle/aln_planes.c (pileup, allele, barcode / fragment bookkeeping from the same alignments) + "
                              "oracle/smc_oracle.c, on %d host threads" % nthreads)
    return tot


def lib_sha16() -> str:
    """First 16 hex digits of sha256(libsmcounter_hip.so): what a traffic record was measured on."""
    import hashlib
    from smcounter_amd import build
    with open(os.environ.get("SMC_HIP_LIB") or build.LIB, "rb") as fh:
        return hashlib.sha256(fh.read()).hexdigest()[:16]


def traffic_record(key: str):
    """-> (record of profiles/traffic.json or None, why not).  A record carries the hash of the library its counters were taken
    on (`lib_sha16`): when the library that runs now is another one, the PMC bytes say nothing about it and are not printed."""
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(tpath):
        return None, "no profiles/traffic.json"
    rec = json.load(open(tpath)).get(key)
    if rec is None:
        return None, "no PMC record for %s in profiles/traffic.json" % key
    have = lib_sha16()
    if rec.get("lib_sha16") != have:
        return None, ("profiles/traffic.json's record for %s was measured on library %s, this run's is %s: not printed"
                      % (key, rec.get("lib_sha16", "(unrecorded)"), have))
    return rec, None


def roofline_block(run: AlignmentRun, k_ms: float, k_n: int, cfg_name: str):
    need = run.needed_bytes()
    rec, why = traffic_record("fa:%s:%d" % (cfg_name, run.nl))
    traffic = rec["hbm_bytes_per_launch"] / (k_ms * 1e-3) / 1e9 if rec else None
    return {"bound": "hbm", "kernel": "k_bp_emit2 (the walk that writes the read words: the step's dominant kernel)",
            "kernel_ms": k_ms, "kernel_samples": k_n, "needed_bytes_per_launch": need,
            "achieved": need / (k_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": need / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "frac_basis": "needed bytes: 2 B in (base + quality) + %d B out (the read word) per pileup read, alignment + row records "
                          "and CIGARs once, umi_start + descriptor per locus; over the kernel's mean HIP-event duration%s"
                          % (run.word_bits // 8, " (16-bit read words since round 5's second half: a third fewer bytes to move than the 32-bit word's "
                                                 "2 + 4 per read - the kernel's time fell by less than its bytes, so this fraction is LOWER than with "
                                                 "32-bit words while the step is faster; SMC_WORD_BITS=32 runs the old format)" if run.word_bits == 16 else ""),
            # (for comparison across rounds: the same launch priced on the bytes the 32-bit word needed - what rounds 4 and 5a quoted)
            **({"frac_on_the_32_bit_words_bytes": (need + 2.0 * run.ns) / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS} if run.word_bits == 16 else {}),
            "reads_per_s_kernel": run.reads / (k_ms * 1e-3),
            "traffic": traffic, "traffic_measured_in_run": False, **({"traffic_note": why} if why else {}),
            "traffic_source": ("profiles/traffic.json <- " + rec.get("source", "rocprofv3 --pmc")) if rec else None,
            "hbm_bytes_per_launch_pmc": rec["hbm_bytes_per_launch"] if rec else None}


def run_leg(eng, cfg_name: str, n_loci: int, steps: int, warmup: int, blocks: int, nthreads: int, parity_loci: int = -1, slots: int = 2,
            place: int = 0):
    """The leg on one GPU, alone (scripts, `python3 -m bench_fa`; bench.py's `from_alignments` entries of the shapes beside the
    headline: C5, X3, EX, C2); bench.py drives the same pieces itself for the headline."""
    cfg = synth.CONFIGS[cfg_name]
    params = synth.params_for(cfg)
    L = eng.L
    run = AlignmentRun(eng, cfg, params, n_loci, nthreads, slots=slots, place=place)
    for _ in range(max(1, warmup)):
        run.step()
    L.smc_device_sync(eng.ctx)
    st = run.status()
    for k in run.t:
        run.t[k] = 0
    _lib.check(L.smc_build_set_timing(eng.ctx, min(256, steps * blocks)), "smc_build_set_timing")
    times = []
    for _ in range(max(1, blocks)):
        L.smc_device_sync(eng.ctx)
        t0 = time.perf_counter()
        for _ in range(steps):
            run.step()
        L.smc_device_sync(eng.ctx)
        times.append(time.perf_counter() - t0)
    k_ms, k_n = ctypes.c_float(), ctypes.c_int32()
    _lib.check(L.smc_build_kernel_ms(eng.ctx, ctypes.byref(k_ms), ctypes.byref(k_n)), "smc_build_kernel_ms")
    L.smc_build_set_timing(eng.ctx, 0)
    # one step at a time (every step on slot 0's stream): what a step takes when nothing of the next one runs beside it
    serial = run.serial_ms(steps)
    # the locus kernels of the same planes, timed alone with the last plan
    plan = run.step(keep_plan=True, slot=0)
    L.smc_device_sync(eng.ctx)
    plan.set_timing(8)
    for _ in range(8):
        plan.run([run.words, run.uaux[0]], params, run.rows, stream=0)
    c_ms = plan.kernel_ms()[0]
    plan.close()
    el = sorted(times)[len(times) // 2]
    n = max(1, run.t["n"])
    out = {
        "workload": describe(run, cfg_name),
        "step": "smc_build_planes%s (read words) -> smc_plan_create_dev_spec (binning on the device, nothing waits for it) -> smc_plan_run_words%s -> rows in HBM" % (
            ("_w16", "16") if run.word_bits == 16 else ("", "")), "read_word_bits": run.word_bits,
        "value": run.nl * steps / el, "unit": "loci/s", "ms_per_step": el / steps * 1e3,
        "blocks_ms_per_step": [round(t / steps * 1e3, 3) for t in times],
        "slots": slots, "slot_choice": run.slot_choice, "ms_per_step_one_at_a_time": serial, "placement": run.placement,
        "pileup_reads_per_s": run.reads * steps / el,
        "host_ms_per_step": {k: round(v / n * 1e3, 3) for k, v in run.t.items() if k != "n"},
        "k_bp_emit2_ms": k_ms.value, "k_call_v2_ms": c_ms,
        "roofline": roofline_block(run, k_ms.value, k_n.value, cfg_name),
        # SURVEY.md 8d's per-locus figure (16 B per pileup read + 360 B per locus) charged to the WHOLE step
        "whole_step_on_survey_8d": {"bytes_per_step": 16.0 * run.reads + 360.0 * run.nl,
                                    "frac": (16.0 * run.reads + 360.0 * run.nl) / (el / steps) / 1e9 / HBM_PEAK_GBS},
        "builder_status": st,
        "generate_s": round(run.t_gen, 1),
    }
    out["plans"] = plans_record(eng)
    if parity_loci:
        run.step(slot=0)
        L.smc_device_sync(eng.ctx)
        # (the oracle's batch of a chunk is 16 B per pileup read on the host: chunks of ~ 60 M reads)
        out["parity"] = parity_full(run, nthreads, 0 if parity_loci < 0 else parity_loci,
                                    chunk=int(max(200, min(20000, 60e6 // max(1.0, run.reads / run.nl)))))
    run.close()
    return out


def plans_record(eng):
    """Plans this engine has made without waiting for the device (smc_plan_create_dev_spec), how many of them went the exact way,
    how many the DEVICE found not to fit their launches - such a step computed nothing: the measurement would be void."""
    made, exact, misfit = eng.spec_counts()
    if misfit:
        raise SystemExit("bench: %d of %d plans made without the host did not fit their launches - those steps computed nothing" % (misfit, made))
    return {"made_without_waiting_for_the_device": made, "made_the_exact_way": exact, "not_fitting": misfit}


def describe(run: AlignmentRun, cfg_name: str) -> str:
    return ("%s-shaped alignments: %d loci, %d alignments (%d barcodes, %d fragments), %d pileup reads, depth %.0f; resident in HBM "
            "(%.2f GB)" % (cfg_name, run.nl, len(run.A["aln"]), run.A["n_bc"], run.A["n_pair"], run.reads, run.reads / run.nl,
                           run.input_bytes() / 1e9))


if __name__ == "__main__":
    import argparse
    from smcounter_amd.engine import Engine
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C3")
    ap.add_argument("--loci", type=int, default=0)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--blocks", type=int, default=3)
    ap.add_argument("--parity-loci", type=int, default=-1, help="-1: every locus; 0: none; n: n loci as a first, a middle and a last stretch")
    ap.add_argument("--slots", type=int, default=2, help="sets of output arrays + streams consecutive steps alternate between (1: one step at a time)")
    ap.add_argument("--place", type=int, default=0, help="bench-side trials: extra allocations of the read words timed with the real walk at set-up, the fastest kept (0: the library's own placement only)")
    a = ap.parse_args()
    eng = Engine(0)
    cfg = synth.CONFIGS[a.config]
    print(json.dumps(run_leg(eng, a.config, a.loci or cfg.n_loci, a.steps, a.warmup, a.blocks, len(os.sched_getaffinity(0)), a.parity_loci,
                             slots=a.slots, place=a.place)))
