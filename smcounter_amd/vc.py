"""Batch equivalent of the reference's per-locus worker call.

`vc_batch` plays the role of `[pool.apply_async(vc_wrapper, ...) for x in locList]` +
`[p.get() for p in results]` (smCounter.py:683-685): one string per locus, in submission order,
each either the 45-field row of `vc()` (:599), the Zero_Coverage row (:493), or - for a locus the
device flagged - the reference's failure convention `"Exception thrown!\\n" + traceback`
(:605-611), which the caller detects by prefix and re-raises (:689-694, `raise_on_exception`)."""
from __future__ import annotations

import traceback
from typing import List, Optional

from . import abi, engine as _engine, features, rows
from .params import VcParams
from .pileup import PileupBatch

EXC_PREFIX = "Exception thrown!"


def vc_batch(pb, params: VcParams, refprov, eng: Optional[_engine.Engine] = None,
             device: int = 0) -> List[str]:
    """pb: a `PileupBatch` (features are extracted here) or an already built `features.DeviceBatch`
    (bamio.iter_device_batches_native)."""
    own = eng is None
    if own:
        eng = _engine.Engine(device)           # raises loudly without a GPU: there is no CPU path
    try:
        db = pb if isinstance(pb, features.DeviceBatch) else features.extract_features(pb, params)
        out_rows = eng.call_batch_host(db, params)
    finally:
        if own:
            eng.close()
    return _strings(out_rows, db, params, refprov)


def _strings(out_rows, db, params: VcParams, refprov) -> List[str]:
    """The batch's rows as the strings vc() returns (smCounter.py:599): the batch formatter, or - when a locus fails in it -
    locus by locus with the reference's failure convention (:605-611) for the one that does."""
    try:
        return rows.format_rows(out_rows, db, params, refprov)
    except Exception:
        pass
    text = []
    for l in range(db.n_loci):
        try:
            text.append(rows.format_row(out_rows[l], db.chrom[l], db.pos[l], db.ref[l], db.alleles[l],
                                        params, refprov))
        except Exception:
            print("Exception thrown in vc() function at genome location:", db.chrom[l], int(db.pos[l]))
            text.append(EXC_PREFIX + "\n" + traceback.format_exc())
    return text


def _kernel_planes(rb):
    """What the kernels read of a resident batch: its read words + umi_start (else the raw-field planes, packed by the run)."""
    return [rb.words, rb.planes[4]] if getattr(rb, "words", None) is not None else rb.planes


def vc_resident(rb, params: VcParams, refprov, eng: _engine.Engine) -> List[str]:
    """`vc_batch` for a batch whose planes are already in HBM (devplanes.ResidentBatch: built there by smc_build_planes):
    plan, kernels, rows back, strings - no plane ever crosses PCIe."""
    plan = eng.make_plan(rb.loci)
    try:
        out_rows = plan.run_devbuf(_kernel_planes(rb), params)
    finally:
        plan.close()
    eng.last_rows = out_rows                  # (the command line looks at them once more: rows.pi_boundary_loci)
    return _strings(out_rows, rb, params, refprov)


def vc_resident_rows(rb, params: VcParams, eng: _engine.Engine):
    """The numeric rows (abi.ROW_DTYPE, a copy) of a resident batch - what a rank of the distributed command line hands to the
    writing rank (packed) instead of strings."""
    plan = eng.make_plan(rb.loci)
    try:
        return plan.run_devbuf(_kernel_planes(rb), params).copy()
    finally:
        plan.close()


class LocusView(object):
    """What rows.format_rows needs of a batch besides the rows: chrom / pos / ref / allele tables per locus."""

    def __init__(self, chrom, pos, ref, alleles):
        import numpy as np
        self.chrom, self.pos, self.ref, self.alleles = list(chrom), np.asarray(pos, np.int64), list(ref), list(alleles)

    @property
    def n_loci(self):
        return len(self.chrom)


def raise_on_exception(output: List[str], loc_list) -> None:
    """main()'s scan for worker failures (smCounter.py:689-694)."""
    pred = getattr(output, "pred", None)
    if pred is not None and len(pred) == len(output):
        # (a row the native printer produced is no failure message: only the others are looked at)
        import numpy as np
        from .rows import PRED_NONE
        for i in np.flatnonzero(np.asarray(pred) == PRED_NONE).tolist():
            if output[i].startswith(EXC_PREFIX):
                print(output[i])
                raise Exception("Exception thrown in vc() at location: " + str(loc_list[i]))
        return
    for line, loc in zip(output, loc_list):
        if line.startswith(EXC_PREFIX):
            print(line)
            raise Exception("Exception thrown in vc() at location: " + str(loc))
