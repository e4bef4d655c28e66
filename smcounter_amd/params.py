"""Numeric parameters of the per-locus caller.

Mirrors the positional arguments `vc()` receives from `main()` in the reference
(smCounter.py:274, :684) plus the two values `vc()` derives from them before the pileup loop:
`smt` (smCounter.py:302-308) and `ds` (smCounter.py:486).
"""
from __future__ import annotations

import dataclasses

from .py2compat import py2_round


@dataclasses.dataclass(frozen=True)
class VcParams:
    minBQ: int = 20
    minMQ: int = 30
    mtDepth: int = 0
    rpb: float = 0.0
    hpLen: int = 10
    mismatchThr: float = 6.0
    mtDrop: int = 0
    maxMT: int = 0
    primerDist: int = 2

    @property
    def smt(self) -> float:
        # strong-MT threshold by mean reads per barcode (smCounter.py:302-308)
        if self.rpb < 1.5:
            return 2.0
        if self.rpb < 3.0:
            return 3.0
        return 4.0

    @property
    def ds(self) -> int:
        # number of UMIs kept at most (smCounter.py:486)
        return self.maxMT if self.maxMT > 0 else int(py2_round(2.0 * self.mtDepth))
