"""Stand-in for the `pysam` module, used ONLY by oracle/ref_harness.py in the build container.

TEST INFRASTRUCTURE - not part of the product path.

The reference (`/root/reference/smCounter.py`) reads BAM/FASTA through pysam; pysam is not
installed here and the example BAM was never shipped.  This module exposes exactly the
attributes the reference touches (smCounter.py:124-129, 275, 311-312, 316-448, 394) and serves
them from in-memory pileup records (smcounter_amd.pileup.PileupBatch) registered under a fake
file name, so the reference's own vc()/filterVariants() code runs unmodified on synthetic loci.
"""
from __future__ import annotations

_BAMS = {}     # path -> {(chrom, pos:int): [read dict, ...]}
_FASTAS = {}   # path -> {chrom: str}


def register_bam(path, loci):
    _BAMS[path] = loci


def register_fasta(path, chroms):
    _FASTAS[path] = chroms


class _Seq(object):
    """Lazy query_sequence: only the positions the reference indexes are materialised."""
    __slots__ = ("qpos", "site", "ins")

    def __init__(self, qpos, site, ins):
        self.qpos, self.site, self.ins = qpos, site, ins

    def __getitem__(self, k):
        if isinstance(k, slice):
            lo = k.start - (self.qpos + 1)
            hi = k.stop - (self.qpos + 1)
            assert lo == 0 and 0 <= hi <= len(self.ins), (k, self.qpos, self.ins)
            return self.ins[lo:hi]
        assert k == self.qpos, (k, self.qpos)
        return self.site


class _Quals(object):
    __slots__ = ("qpos", "bq")

    def __init__(self, qpos, bq):
        self.qpos, self.bq = qpos, bq

    def __getitem__(self, k):
        assert k == self.qpos
        return self.bq


class _Alignment(object):
    __slots__ = ("query_name", "mapping_quality", "tags", "cigar", "query_length", "is_read1",
                 "is_read2", "is_reverse", "query_sequence", "query_qualities",
                 "query_alignment_length")


class _PileupRead(object):
    __slots__ = ("alignment", "indel", "is_del", "query_position")


class _Column(object):
    __slots__ = ("pileups",)


def _make_read(r):
    a = _Alignment()
    a.query_name = r["qname"]
    a.mapping_quality = r["mq"]
    a.tags = [("XX", 1), ("NM", r["nm"])] if r["has_nm"] else [("XX", 1)]
    a.cigar = r["cigar"]
    a.query_length = r["qlen"]
    a.is_read1 = r["is_read1"]
    a.is_read2 = r["is_read2"]
    a.is_reverse = r["is_reverse"]
    a.query_alignment_length = r["qalen"]
    if "seq" in r:                     # a real record (tests/golden/make_bam_golden.py): the whole sequence and qualities
        a.query_sequence = r["seq"]
        a.query_qualities = r["quals"]
    else:
        a.query_sequence = _Seq(r["qpos"], r["site"], r["ins"])
        a.query_qualities = _Quals(r["qpos"], r["bq"])
    p = _PileupRead()
    p.alignment = a
    p.indel = r["indel"]
    p.is_del = r["is_del"]
    p.query_position = r["qpos"]
    return p


class AlignmentFile(object):
    def __init__(self, path, mode="rb"):
        self._loci = _BAMS[path]

    def pileup(self, region=None, truncate=False, max_depth=8000, stepper="all"):
        # the reference always asks for "chrom:pos:pos" (smCounter.py:316)
        chrom, p0, p1 = region.rsplit(":", 2)
        assert p0 == p1 and truncate and stepper == "nofilter" and max_depth >= 1000000
        reads = self._loci.get((chrom, int(p0)))
        if not reads:
            return
        col = _Column()
        col.pileups = [_make_read(r) for r in reads]
        yield col


class FastaFile(object):
    def __init__(self, path):
        self._chroms = _FASTAS[path]

    def fetch(self, reference=None, start=None, end=None):
        s = self._chroms[reference]
        start = max(0, start)
        return s[start:end]

    def get_reference_length(self, chrom):
        return len(self._chroms[chrom])
