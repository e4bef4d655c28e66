"""Minimal indexed FASTA reader with the two calls the caller needs (`fetch`,
`get_reference_length`; the reference uses pysam.FastaFile at smCounter.py:124-129, :311-312, :394).
Uses `<fasta>.fai` when present, otherwise indexes the file in one pass."""
from __future__ import annotations

import os


class FastaFile(object):
    def __init__(self, path: str):
        self._path = path
        self._fh = open(path, "rb")
        self._fd = self._fh.fileno()
        self._idx = {}
        fai = path + ".fai"
        if os.path.exists(fai):
            for line in open(fai):
                name, length, off, lb, lw = line.rstrip("\n").split("\t")[:5]
                self._idx[name] = (int(length), int(off), int(lb), int(lw))
        else:
            self._build()

    def _build(self):
        fh = self._fh
        fh.seek(0)
        name, length, off, lb, lw = None, 0, 0, 0, 0
        pos = 0
        for raw in fh:
            if raw.startswith(b">"):
                if name is not None:
                    self._idx[name] = (length, off, lb, lw)
                name = raw[1:].split()[0].decode()
                length, off, lb, lw = 0, pos + len(raw), 0, 0
            else:
                s = raw.rstrip(b"\r\n")
                if lb == 0:
                    lb, lw = len(s), len(raw)
                length += len(s)
            pos += len(raw)
        if name is not None:
            self._idx[name] = (length, off, lb, lw)

    def get_reference_length(self, chrom: str) -> int:
        return self._idx[chrom][0]

    def fetch(self, chrom: str = None, start: int = 0, end: int = None, reference: str = None) -> str:
        chrom = reference if chrom is None else chrom
        length, off, lb, lw = self._idx[chrom]
        start = max(0, start)
        end = length if end is None else min(end, length)
        if end <= start:
            return ""
        b0 = off + (start // lb) * lw + start % lb
        b1 = off + ((end - 1) // lb) * lw + (end - 1) % lb + 1
        # positional read: atomic per call, so the decoder's helper thread (reference run / DEL allele strings)
        # and the main thread (HP / LowC flanks) can share one handle; seek() + read() would interleave
        raw, want = b"", b1 - b0
        while len(raw) < want:
            part = os.pread(self._fd, want - len(raw), b0 + len(raw))
            if not part:
                break
            raw += part
        return raw.replace(b"\n", b"").replace(b"\r", b"").decode()

    def close(self):
        self._fh.close()
