"""BAM decode -> per-locus pileup records.

pysam is not available where this runs, so this module reads BGZF/BAM itself (zlib) and produces,
for every requested locus, what the reference takes from `samfile.pileup(region, truncate=True,
max_depth=1000000, stepper='nofilter')` at smCounter.py:316-448:

* every mapped alignment whose reference span covers the position, in file (coordinate) order -
  no flag, MAPQ or base-quality filtering, duplicates / secondary / supplementary included,
  no mate-overlap handling;
* per alignment: `query_position` (index into the read incl. soft clips), `is_del` (position inside
  a D or N operation), `indel` (length of the insertion (+) / deletion (-) that starts right after
  this position, i.e. the position is the last base of an M block followed by I / D);
* the facts of smCounter.py:319-366: UMI and read id from the qname, MAPQ, NM, CIGAR summary,
  lengths, flags.

The result is a `pileup.PileupBatch` (dense barcode / fragment ids, per-locus allele tables).
pysam/samtools pileup semantics are not pinned by the reference (no version, BAM not shipped):
this follows samtools 0.1.19's resolve_cigar, the era the README names.

A small BAM writer (`write_bam`) is included for tests and fixtures.
"""
from __future__ import annotations

import os
import struct
import zlib
from typing import Dict, Iterable, Iterator, List, Optional, Sequence, Tuple

import numpy as np

from .pileup import (BASE_ALLELES, F_HAS_NM, F_READ1, F_READ2, F_REVERSE, PileupBatch)

_SEQ_CODE = "=ACMGRSVTWYHKDBN"
_CIG_M, _CIG_I, _CIG_D, _CIG_N, _CIG_S, _CIG_H, _CIG_P, _CIG_EQ, _CIG_X = range(9)
_REF_OPS = (_CIG_M, _CIG_D, _CIG_N, _CIG_EQ, _CIG_X)
_QRY_OPS = (_CIG_M, _CIG_I, _CIG_S, _CIG_EQ, _CIG_X)


class BamError(Exception):
    pass


# ------------------------------------------------------------------------------------------------
# BGZF
# ------------------------------------------------------------------------------------------------
class BgzfReader(object):
    """Sequential / seekable reader over BGZF blocks; `tell`/`seek` use BAM virtual offsets."""

    def __init__(self, path: str):
        self._fh = open(path, "rb")
        self._block_start = 0
        self._buf = b""
        self._off = 0
        self._next_block = 0

    def close(self):
        self._fh.close()

    def _load_block(self, coffset: int) -> bool:
        self._fh.seek(coffset)
        hdr = self._fh.read(18)
        if len(hdr) < 18:
            self._buf, self._off = b"", 0
            return False
        if hdr[0] != 31 or hdr[1] != 139 or hdr[12:14] != b"BC":
            raise BamError("not a BGZF block at offset %d" % coffset)
        xlen = struct.unpack_from("<H", hdr, 10)[0]
        bsize = struct.unpack_from("<H", hdr, 16)[0] + 1
        rest = self._fh.read(bsize - 18)
        cdata = rest[xlen - 6:-8]
        self._buf = zlib.decompress(cdata, -15) if cdata else b""
        self._off = 0
        self._block_start = coffset
        self._next_block = coffset + bsize
        return True

    def seek(self, voffset: int):
        self._load_block(voffset >> 16)
        self._off = voffset & 0xFFFF

    def tell(self) -> int:
        return (self._block_start << 16) | self._off

    def read(self, n: int) -> bytes:
        out = []
        while n > 0:
            if self._off >= len(self._buf):
                if not self._load_block(self._next_block):
                    break
                continue
            take = self._buf[self._off:self._off + n]
            out.append(take)
            self._off += len(take)
            n -= len(take)
        return b"".join(out)


def _bgzf_block(data: bytes) -> bytes:
    comp = zlib.compressobj(6, zlib.DEFLATED, -15)
    c = comp.compress(data) + comp.flush()
    bsize = len(c) + 25
    return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", bsize) + c
            + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data)))


# ------------------------------------------------------------------------------------------------
# BAM records
# ------------------------------------------------------------------------------------------------
class Alignment(object):
    __slots__ = ("tid", "pos", "end", "qname", "flag", "mapq", "cigar", "seq", "qual", "nm", "has_nm",
                 "l_seq")


def _parse_aux_nm(aux: bytes):
    i, n = 0, len(aux)
    fixed = {ord("A"): 1, ord("c"): 1, ord("C"): 1, ord("s"): 2, ord("S"): 2, ord("i"): 4, ord("I"): 4,
             ord("f"): 4}
    fmt = {ord("c"): "<b", ord("C"): "<B", ord("s"): "<h", ord("S"): "<H", ord("i"): "<i", ord("I"): "<I"}
    while i + 3 <= n:
        tag, typ = aux[i:i + 2], aux[i + 2]
        i += 3
        if typ in fixed:
            if tag == b"NM" and typ in fmt:
                return struct.unpack_from(fmt[typ], aux, i)[0], True
            i += fixed[typ]
        elif typ in (ord("Z"), ord("H")):
            i = aux.index(b"\x00", i) + 1
        elif typ == ord("B"):
            sub = aux[i]
            cnt = struct.unpack_from("<I", aux, i + 1)[0]
            i += 5 + cnt * fixed[sub]
        else:
            raise BamError("unknown aux type %r" % chr(typ))
    return 0, False            # the reference's default when no NM tag is present (smCounter.py:329)


def _parse_record(b: bytes) -> Alignment:
    tid, pos, l_name, mapq, _bin, n_cig, flag, l_seq, _ntid, _npos, _tlen = struct.unpack_from("<iiBBHHHiiii", b, 0)
    a = Alignment()
    a.tid, a.pos, a.mapq, a.flag, a.l_seq = tid, pos, mapq, flag, l_seq
    o = 32
    a.qname = b[o:o + l_name - 1].decode()
    o += l_name
    cig = struct.unpack_from("<%dI" % n_cig, b, o) if n_cig else ()
    a.cigar = [(c & 15, c >> 4) for c in cig]
    o += 4 * n_cig
    nb = (l_seq + 1) // 2
    packed = b[o:o + nb]
    o += nb
    s = []
    for byte in packed:
        s.append(_SEQ_CODE[byte >> 4])
        s.append(_SEQ_CODE[byte & 15])
    a.seq = "".join(s[:l_seq])
    a.qual = b[o:o + l_seq]
    o += l_seq
    a.nm, a.has_nm = _parse_aux_nm(b[o:])
    a.end = pos + sum(l for op, l in a.cigar if op in _REF_OPS)
    return a


class BamFile(object):
    def __init__(self, path: str):
        self.path = path
        self._bg = BgzfReader(path)
        if self._bg.read(4) != b"BAM\x01":
            raise BamError("%s is not a BAM file" % path)
        l_text = struct.unpack("<i", self._bg.read(4))[0]
        self.header_text = self._bg.read(l_text).decode(errors="replace")
        n_ref = struct.unpack("<i", self._bg.read(4))[0]
        self.refs: List[Tuple[str, int]] = []
        for _ in range(n_ref):
            l_name = struct.unpack("<i", self._bg.read(4))[0]
            name = self._bg.read(l_name)[:-1].decode()
            self.refs.append((name, struct.unpack("<i", self._bg.read(4))[0]))
        self.tid_of = {n: i for i, (n, _) in enumerate(self.refs)}
        self._first_record = self._bg.tell()
        self._lin_index = _load_bai(path + ".bai", n_ref) or _load_bai(path[:-4] + ".bai", n_ref)

    def close(self):
        self._bg.close()

    def _records(self) -> Iterator[Alignment]:
        while True:
            h = self._bg.read(4)
            if len(h) < 4:
                return
            yield _parse_record(self._bg.read(struct.unpack("<i", h)[0]))

    def fetch(self, chrom: str, start: int, end: int) -> List[Alignment]:
        """Mapped alignments overlapping [start, end) (0-based), in file order.  Uses the BAI linear
        index when present, else scans from the first record (file must be coordinate-sorted)."""
        tid = self.tid_of.get(chrom)
        if tid is None:
            return []
        voff = self._first_record
        if self._lin_index and tid < len(self._lin_index) and self._lin_index[tid]:
            iv = self._lin_index[tid]
            w = min(start >> 14, len(iv) - 1)
            while w >= 0 and iv[w] == 0:
                w -= 1
            if w >= 0:
                voff = iv[w]
        self._bg.seek(voff)
        out = []
        for a in self._records():
            if a.tid < 0 or a.tid > tid or (a.tid == tid and a.pos >= end):
                break
            if a.tid < tid or (a.flag & 0x4) or not a.cigar:
                continue
            if a.end > start:
                out.append(a)
        return out


def _load_bai(path: str, n_ref: int):
    try:
        b = open(path, "rb").read()
    except OSError:
        return None
    if b[:4] != b"BAI\x01":
        return None
    o = 8
    lin = []
    for _ in range(struct.unpack_from("<i", b, 4)[0]):
        n_bin = struct.unpack_from("<i", b, o)[0]
        o += 4
        for _ in range(n_bin):
            n_chunk = struct.unpack_from("<i", b, o + 4)[0]
            o += 8 + 16 * n_chunk
        n_intv = struct.unpack_from("<i", b, o)[0]
        o += 4
        lin.append(list(struct.unpack_from("<%dQ" % n_intv, b, o)))
        o += 8 * n_intv
    return lin


def locus_weights(path: str, loci: Sequence[Tuple[str, str]]) -> np.ndarray:
    """A depth proxy per locus for balancing shards without decoding anything: compressed bytes of the locus's 16 kb
    window according to the BAI linear index (file offsets of the first alignment of every window).  Panels put one
    amplicon in a window, so this is proportional to the amplicon's reads.  All ones when there is no index."""
    w = np.ones(len(loci), np.float64)
    try:
        bam = BamFile(path)
    except (OSError, BamError):
        return w
    lin, tid_of = bam._lin_index, bam.tid_of
    bam.close()
    if not lin:
        return w
    size = os.path.getsize(path)
    starts = [(tid, k, v >> 16) for tid, iv in enumerate(lin) for k, v in enumerate(iv) if v]
    dens = {}
    for j, (tid, k, c) in enumerate(starts):
        nxt = starts[j + 1][2] if j + 1 < len(starts) else size
        dens[(tid, k)] = float(max(nxt - c, 0))
    for i, (chrom, pos) in enumerate(loci):
        tid = tid_of.get(chrom)
        if tid is not None:
            w[i] = 1.0 + dens.get((tid, (int(pos) - 1) >> 14), 0.0)
    return w


def write_bam(path: str, refs: Sequence[Tuple[str, int]], records: Iterable[dict], block: int = 60000) -> None:
    """Tiny BAM writer (coordinate-sorted input expected).  record keys: tid, pos, qname, flag, mapq,
    cigar [(op, len)], seq, qual (bytes or list of ints), nm (int or None)."""
    text = "@HD\tVN:1.4\tSO:coordinate\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % r for r in refs)
    out = [b"BAM\x01", struct.pack("<i", len(text)), text.encode(), struct.pack("<i", len(refs))]
    for name, ln in refs:
        out.append(struct.pack("<i", len(name) + 1) + name.encode() + b"\x00" + struct.pack("<i", ln))
    for r in records:
        seq = r["seq"]
        qual = bytes(r["qual"])
        packed = bytearray((len(seq) + 1) // 2)
        for i, ch in enumerate(seq):
            packed[i >> 1] |= _SEQ_CODE.index(ch) << (4 if i % 2 == 0 else 0)
        cig = b"".join(struct.pack("<I", (l << 4) | op) for op, l in r["cigar"])
        aux = b"" if r.get("nm") is None else b"NMC" + struct.pack("<B", r["nm"])
        name = r["qname"].encode() + b"\x00"
        body = struct.pack("<iiBBHHHiiii", r["tid"], r["pos"], len(name), r["mapq"], 4680, len(r["cigar"]),
                           r["flag"], len(seq), -1, -1, 0) + name + cig + bytes(packed) + qual + aux
        out.append(struct.pack("<i", len(body)) + body)
    data = b"".join(out)
    with open(path, "wb") as fh:
        for i in range(0, len(data), block):
            fh.write(_bgzf_block(data[i:i + block]))
        fh.write(_bgzf_block(b""))


def write_bai(bam_path: str) -> str:
    """Write <bam>.bai holding the 16 kb linear index only (no bins): enough for fetch() here and in
    csrc/smc_bam.cpp, which use nothing else."""
    bam = BamFile(bam_path)
    bg = bam._bg
    bg.seek(bam._first_record)
    lin = [[] for _ in bam.refs]
    while True:
        v = bg.tell()
        if bg._off >= len(bg._buf):           # record starts in the next block: normalise the offset
            if not bg._load_block(bg._next_block):
                break
            if not bg._buf:
                break
            v = bg.tell()
        h = bg.read(4)
        if len(h) < 4:
            break
        a = _parse_record(bg.read(struct.unpack("<i", h)[0]))
        if a.tid < 0:
            continue
        iv = lin[a.tid]
        for w in range(a.pos >> 14, (max(a.end, a.pos + 1) - 1 >> 14) + 1):
            while len(iv) <= w:
                iv.append(0)
            if iv[w] == 0:
                iv[w] = v
    out = [b"BAI\x01", struct.pack("<i", len(lin))]
    for iv in lin:
        out.append(struct.pack("<ii", 0, len(iv)) + struct.pack("<%dQ" % len(iv), *iv))
    out.append(struct.pack("<Q", 0))
    bam.close()
    with open(bam_path + ".bai", "wb") as fh:
        fh.write(b"".join(out))
    return bam_path + ".bai"


def iter_raw_records(path: str):
    """(header bytes up to the first record, iterator of (tid, qname, raw record incl. its block_size))."""
    bam = BamFile(path)
    bg = bam._bg
    bg.seek(0)
    n_head = None
    # header length in uncompressed bytes: re-read it sequentially
    hdr = bytearray()
    magic = bg.read(4)
    l_text = bg.read(4)
    text = bg.read(struct.unpack("<i", l_text)[0])
    n_ref_b = bg.read(4)
    hdr += magic + l_text + text + n_ref_b
    for _ in range(struct.unpack("<i", n_ref_b)[0]):
        ln = bg.read(4)
        rest = bg.read(struct.unpack("<i", ln)[0] + 4)
        hdr += ln + rest

    def records():
        while True:
            h = bg.read(4)
            if len(h) < 4:
                break
            body = bg.read(struct.unpack("<i", h)[0])
            tid = struct.unpack_from("<i", body, 0)[0]
            l_name = body[8]
            yield tid, body[32:32 + l_name - 1].decode(), h + body
        bam.close()
    return bytes(hdr), records()


def write_raw(path: str, header: bytes, raw_records: Iterable[bytes], block: int = 60000) -> None:
    """BGZF-compress a header and raw BAM records (as `iter_raw_records` yields them) into a BAM file."""
    with open(path, "wb") as fh:
        buf = bytearray(header)
        for r in raw_records:
            buf += r
            while len(buf) >= block:
                fh.write(_bgzf_block(bytes(buf[:block])))
                del buf[:block]
        if buf:
            fh.write(_bgzf_block(bytes(buf)))
        fh.write(_bgzf_block(b""))


# ------------------------------------------------------------------------------------------------
# pileup
# ------------------------------------------------------------------------------------------------
def _column(a: Alignment, pos0: int):
    """(qpos, is_del, indel) of alignment `a` at 0-based reference position pos0, or None."""
    x, y = a.pos, 0
    cig = a.cigar
    for k, (op, l) in enumerate(cig):
        if op in (_CIG_M, _CIG_EQ, _CIG_X):
            if x <= pos0 < x + l:
                indel = 0
                if pos0 == x + l - 1 and k + 1 < len(cig):
                    nop, nl = cig[k + 1]
                    if nop == _CIG_I:
                        indel = nl
                    elif nop == _CIG_D:
                        indel = -nl
                return y + (pos0 - x), False, indel
            x += l
            y += l
        elif op in (_CIG_I, _CIG_S):
            y += l
        elif op in (_CIG_D, _CIG_N):
            if x <= pos0 < x + l:
                # the peek at the next operation applies to every current operation (samtools resolve_cigar2): the
                # last column of a D/N block followed by I or D carries that indel (the reference tests `indel`
                # before `is_del`, smCounter.py:371,392,416)
                indel = 0
                if pos0 == x + l - 1 and k + 1 < len(cig):
                    nop, nl = cig[k + 1]
                    if nop == _CIG_I:
                        indel = nl
                    elif nop == _CIG_D:
                        indel = -nl
                if indel and y >= a.l_seq:
                    indel = 0           # no base to name the allele with: plain 'DEL'
                return y, True, indel
            x += l
    return None


class _LocusBuilder(object):
    def __init__(self):
        self.cols = {k: [] for k in ("umi", "frag", "flag", "mq", "nm", "n_indel", "left_sp", "qlen", "qalen",
                                     "qpos", "indel", "is_del", "allele", "bq")}
        self.chrom, self.pos, self.ref, self.alleles, self.off = [], [], [], [], [0]
        self.umi_names = []

    def add_locus(self, chrom: str, pos1: int, reads: List[Alignment], fasta):
        pos0 = pos1 - 1
        table = list(BASE_ALLELES)
        index = {s: i for i, s in enumerate(table)}
        umis: Dict[str, int] = {}
        frags: List[Dict[str, int]] = []
        c = self.cols
        n = 0
        for a in reads:
            col = _column(a, pos0)
            if col is None:
                continue
            qpos, is_del, indel = col
            parts = a.qname.split(":")
            if len(parts) < 3:
                raise BamError("read name %r has fewer than 3 ':' fields; the reference needs "
                               "<readid>:<tag>:<UMI>:<x> (smCounter.py:320-325)" % a.qname)
            bc, readid = parts[-2], ":".join(parts[:-2])
            u = umis.setdefault(bc, len(umis))
            if u == len(frags):
                frags.append({})
            f = frags[u].setdefault(readid, len(frags[u]))
            if a.l_seq == 0:
                raise BamError("alignment %s has no sequence; the reference indexes query_sequence" % a.qname)
            if is_del and indel == 0:
                key, bq = "DEL", 0
            else:
                site = a.seq[qpos]
                bq = a.qual[qpos]
                if indel > 0:
                    key = "INS|" + site + "|" + site + a.seq[qpos + 1:qpos + 1 + indel]
                elif indel < 0:
                    key = "DEL|" + site + fasta.fetch(chrom, pos1, pos1 - indel).upper() + "|" + site
                else:
                    key = site
            ai = index.get(key)
            if ai is None:
                ai = index[key] = len(table)
                table.append(key)
            cig = a.cigar
            c["umi"].append(u)
            c["frag"].append(f)
            c["flag"].append((F_READ1 if a.flag & 0x40 else 0) | (F_READ2 if a.flag & 0x80 else 0)
                             | (F_REVERSE if a.flag & 0x10 else 0) | (F_HAS_NM if a.has_nm else 0))
            c["mq"].append(a.mapq)
            c["nm"].append(a.nm)
            c["n_indel"].append(sum(l for op, l in cig if op in (_CIG_I, _CIG_D)))
            c["left_sp"].append(cig[0][1] if cig[0][0] == _CIG_S else 0)
            c["qlen"].append(a.l_seq)
            c["qalen"].append(sum(l for op, l in cig if op in (_CIG_M, _CIG_I, _CIG_EQ, _CIG_X)))
            c["qpos"].append(qpos)
            c["indel"].append(indel)
            c["is_del"].append(is_del)
            c["allele"].append(ai)
            c["bq"].append(bq)
            n += 1
        self.chrom.append(chrom)
        self.pos.append(pos1)
        self.ref.append(fasta.fetch(chrom, pos0, pos1).upper())
        self.alleles.append(table)
        self.umi_names.append(list(umis))          # barcode texts by dense id (for the py2 down-sampling)
        self.off.append(self.off[-1] + n)

    def build(self) -> PileupBatch:
        dt = dict(umi=np.uint32, frag=np.uint32, flag=np.uint8, mq=np.uint8, nm=np.uint32, n_indel=np.uint32,
                  left_sp=np.uint32, qlen=np.uint32, qalen=np.uint32, qpos=np.int32, indel=np.int32,
                  is_del=bool, allele=np.uint8, bq=np.uint8)
        return PileupBatch(chrom=self.chrom, pos=np.array(self.pos, np.int64), ref=self.ref,
                           alleles=self.alleles, read_off=np.array(self.off, np.int64), umi_names=self.umi_names,
                           **{k: np.array(v, dt[k]) for k, v in self.cols.items()})


def iter_pileup_batches(bam: BamFile, fasta, loci: Sequence[Tuple[str, str]], max_reads: int = 2_000_000):
    """Yield (first locus index, PileupBatch) chunks covering `loci` [(chrom, '1-based pos')] in order;
    a chunk closes once it holds max_reads pileup reads.  Consecutive positions of one chromosome
    share one fetch."""
    i, n = 0, len(loci)
    while i < n:
        builder = _LocusBuilder()
        first = i
        while i < n and builder.off[-1] < max_reads:
            chrom = loci[i][0]
            j = i
            while j + 1 < n and loci[j + 1][0] == chrom and int(loci[j + 1][1]) == int(loci[j][1]) + 1 \
                    and j + 1 - i < 4096:
                j += 1
            lo, hi = int(loci[i][1]) - 1, int(loci[j][1])          # 0-based [lo, hi)
            reads = bam.fetch(chrom, lo, hi)
            w0 = 0
            for k in range(i, j + 1):
                p0 = int(loci[k][1]) - 1
                while w0 < len(reads) and reads[w0].end <= p0:
                    w0 += 1            # sorted by start: a head read that ended is never needed again
                cover = []
                for r in reads[w0:]:
                    if r.pos > p0:
                        break
                    if p0 < r.end:
                        cover.append(r)
                builder.add_locus(chrom, p0 + 1, cover, fasta)
                if builder.off[-1] >= max_reads:
                    j = k
                    break
            i = j + 1
        yield first, builder.build()


# ------------------------------------------------------------------------------------------------
# native decoder (csrc/smc_bam.cpp): same batches, two orders of magnitude faster
# ------------------------------------------------------------------------------------------------
_NATIVE = None
import ctypes as _C
# allocation callback of smc_bam_planes: (ctx, n_slots, n_loci, void* out[5])
_PLANES_ALLOC = _C.CFUNCTYPE(None, _C.c_void_p, _C.c_int64, _C.c_int64, _C.POINTER(_C.c_void_p))
# allocation callback of smc_bam_alignments: (ctx, n_aln, n_cig, n_seq, n_loci, void* out[5])
_ALN_ALLOC = _C.CFUNCTYPE(None, _C.c_void_p, _C.c_int64, _C.c_int64, _C.c_int64, _C.c_int64, _C.POINTER(_C.c_void_p))


def _native_lib():
    global _NATIVE
    if _NATIVE is None:
        import ctypes as C
        from . import build
        lib = C.CDLL(build.build_bam())
        lib.smc_bam_open.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
        lib.smc_bam_close.argtypes = [C.c_void_p]
        lib.smc_bam_error.argtypes = [C.c_void_p]
        lib.smc_bam_error.restype = C.c_char_p
        lib.smc_bam_pileup.argtypes = [C.c_void_p, C.c_char_p, C.c_int64, C.c_int64, C.c_int64, C.POINTER(C.c_int64)]
        lib.smc_bam_pileup.restype = C.c_int64
        lib.smc_bam_keys_len.argtypes = [C.c_void_p]
        lib.smc_bam_keys_len.restype = C.c_int64
        lib.smc_bam_copy.argtypes = [C.c_void_p] * 18
        lib.smc_bam_planes.argtypes = [C.c_void_p, C.c_char_p, C.c_int64, C.c_int64, C.c_int64, C.c_double,
                                       C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _PLANES_ALLOC, C.c_void_p] \
            + [C.POINTER(C.c_int64)] * 3
        lib.smc_bam_ds_info.argtypes = [C.c_void_p]
        lib.smc_bam_ds_info.restype = C.c_char_p
        lib.smc_bam_planes.restype = C.c_int64
        lib.smc_bam_planes_copy.argtypes = [C.c_void_p] * 4
        lib.smc_bam_alignments.argtypes = [C.c_void_p, C.c_char_p, C.c_int64, C.c_int64, C.c_int64, C.c_double, C.c_int,
                                           _ALN_ALLOC, C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                           C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        lib.smc_bam_alignments.restype = C.c_int64
        lib.smc_bam_allele_key.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_char_p, C.c_int]
        lib.smc_bam_barcode_name.argtypes = [C.c_void_p, C.c_int32]
        lib.smc_bam_barcode_name.restype = C.c_char_p
        lib.smc_bam_barcode_idents.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        lib.smc_bam_barcode_idents.restype = C.c_int64
        lib.smc_bam_span_bytes.argtypes = [C.c_void_p, C.c_char_p, C.c_int64, C.c_int64]
        lib.smc_bam_span_bytes.restype = C.c_int64
        _NATIVE = lib
    return _NATIVE


_COLS = (("umi", np.uint32), ("frag", np.uint32), ("flag", np.uint8), ("mq", np.uint8), ("nm", np.uint32),
         ("n_indel", np.uint32), ("left_sp", np.uint32), ("qlen", np.uint32), ("qalen", np.uint32),
         ("qpos", np.int32), ("indel", np.int32), ("is_del", np.uint8), ("allele", np.uint8), ("bq", np.uint8))


class NativeBam(object):
    """ctypes handle on csrc/smc_bam.cpp.  `pileup_run` returns the columns of a run of consecutive
    positions; deletion-start allele keys are completed here from the FASTA."""

    def __init__(self, path: str):
        import ctypes as C
        self._lib = _native_lib()
        self._h = C.c_void_p()
        rc = self._lib.smc_bam_open(path.encode(), C.byref(self._h))
        if rc:
            raise BamError("%s: cannot open as BAM (rc %d)" % (path, rc))

    def close(self):
        if self._h:
            self._lib.smc_bam_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def pileup_run(self, chrom: str, lo: int, hi: int, max_reads: int, fasta):
        """-> (n loci done, column dict, read_off [n+1], allele tables).  [lo, hi) 0-based."""
        import ctypes as C
        done = C.c_int64(0)
        n = self._lib.smc_bam_pileup(self._h, chrom.encode(), lo, hi, max_reads, C.byref(done))
        if n < 0:
            raise BamError(self._lib.smc_bam_error(self._h).decode())
        nl = done.value
        cols = {k: np.empty(n, dt) for k, dt in _COLS}
        off = np.empty(nl + 1, np.int64)
        n_keys = np.empty(nl, np.int32)
        keys = C.create_string_buffer(max(1, self._lib.smc_bam_keys_len(self._h)))
        ptr = lambda a: a.ctypes.data_as(C.c_void_p)
        self._lib.smc_bam_copy(self._h, *[ptr(cols[k]) for k, _ in _COLS], ptr(off), ptr(n_keys),
                               C.cast(keys, C.c_void_p))
        cols["is_del"] = cols["is_del"].astype(bool)
        return nl, cols, off, self._tables(nl, n_keys, keys, chrom, lo, fasta)

    def _tables(self, nl, n_keys, keys, chrom, lo, fasta):
        tables = [list(BASE_ALLELES) for _ in range(nl)]
        if n_keys.any():
            flat = keys.raw[:self._lib.smc_bam_keys_len(self._h)].decode().split("\n")
            k = 0
            for l in np.nonzero(n_keys)[0]:
                pos1 = lo + int(l) + 1
                for key in flat[k:k + int(n_keys[l])]:
                    if key[0] == "D" and len(key) > 1:   # "D<len>|<site>" -> DEL|site+deleted|site (smCounter.py:392-396)
                        ln, site = key[1:].split("|")
                        key = "DEL|" + site + fasta.fetch(chrom, pos1, pos1 + int(ln)).upper() + "|" + site
                    tables[l].append(key)
                k += int(n_keys[l])
        return tables

    def alignments_run(self, chrom: str, lo: int, hi: int, max_reads: int, params, nthreads: int, host_array=None):
        """Host half of the device plane builder (smc_bam_alignments): the run's alignments as a structure of arrays -
        dict(aln, cig, bq (+ its views seq, qual), loc, nl, n_slots, n_bc, n_pair, status, reads).  The handle keeps the run's records:
        `allele_key` / `barcode_name` refer to them until the next call.  `host_array(name, dtype, count)` (optional)
        provides the arrays (engine.Engine.pinned: page-locked staging memory, reused from run to run)."""
        import ctypes as C
        from .abi import DEV_ALN_DTYPE, DEV_LOCUS_DTYPE
        got = {}
        mk = host_array or (lambda name, dtype, count: np.empty(count, dtype))

        def alloc(ctx, n_aln, n_cig, n_seq, n_loci, out):
            got["aln"] = mk("aln", DEV_ALN_DTYPE, n_aln)
            got["cig"] = mk("cig", np.uint32, max(1, n_cig))
            got["bq"] = mk("bq", np.uint8, 2 * max(1, n_seq))             # (letter, quality) byte pairs: smc_build_in.bq
            got["seq"], got["qual"] = got["bq"][0::2], got["bq"][1::2]      # (views: the letters, the qualities)
            got["loc"] = mk("loc", DEV_LOCUS_DTYPE, n_loci)
            for k, name in enumerate(("aln", "cig", "bq", "loc")):
                out[k] = got[name].ctypes.data
        cb = _ALN_ALLOC(alloc)
        done, n_slots = C.c_int64(0), C.c_int64(0)
        n_bc, n_pair, status = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        n = self._lib.smc_bam_alignments(self._h, chrom.encode(), lo, hi, max_reads, float(params.mismatchThr), int(nthreads),
                                         cb, None, C.byref(done), C.byref(n_slots), C.byref(n_bc), C.byref(n_pair),
                                         C.byref(status))
        if n < 0:
            raise BamError(self._lib.smc_bam_error(self._h).decode())
        got.update(nl=done.value, n_slots=n_slots.value, n_bc=n_bc.value, n_pair=n_pair.value, status=status.value, reads=n)
        return got

    def span_bytes(self, chrom: str, lo: int, hi: int) -> int:
        """Compressed bytes holding [lo, hi) of `chrom` per the linear index (16 kb granules); -1 when not known."""
        return int(self._lib.smc_bam_span_bytes(self._h, chrom.encode(), lo, hi))

    def allele_key(self, ai: int, qpos: int, indel: int) -> str:
        import ctypes as C
        buf = C.create_string_buffer(70000)
        n = self._lib.smc_bam_allele_key(self._h, int(ai), int(qpos), int(indel), buf, len(buf))
        if n < 0:
            raise BamError("allele key of alignment %d at %d: rc %d" % (ai, qpos, n))
        return buf.value.decode()

    def barcode_name(self, gid: int) -> str:
        return self._lib.smc_bam_barcode_name(self._h, int(gid)).decode()

    def barcode_idents(self, n_bc: int) -> np.ndarray:
        """FNV-1a (64 bits) of every barcode text of the last run, by run-wide id: what the non-parity sampler keys on."""
        out = np.zeros(max(1, int(n_bc)), np.uint64)
        n = self._lib.smc_bam_barcode_idents(self._h, out.ctypes.data, len(out))
        return out[:min(int(n), len(out))]

    def planes_run(self, chrom: str, lo: int, hi: int, max_reads: int, params, refseq: str, nthreads: int, fasta,
                   arena=None, arena_off: int = 0):
        """Fused decode + feature extraction of a run -> (n loci, 4 planes, umi_start, LOCUS_DTYPE array,
        allele tables); offsets in the descriptors are relative to this run.  With `arena` (four uint32 arrays) the
        planes are written at `arena_off` of those when they fit (views are returned: a batch of several runs then
        needs no concatenation), into fresh arrays otherwise."""
        import ctypes as C
        from .features import LOCUS_DTYPE, PileupError
        done, n_slots, n_us = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        refb = refseq.encode().ljust(hi - lo, b"\0")
        got = {}

        def alloc(ctx, ns, nl_, out):                    # the library writes straight into these arrays
            if arena is not None and arena_off + ns <= len(arena[0]):
                got["planes"] = [a[arena_off:arena_off + ns] for a in arena]
            else:
                got["planes"] = [np.empty(ns, np.uint32) for _ in range(4)]
            got["loci"] = np.empty(nl_, LOCUS_DTYPE)
            for k in range(4):
                out[k] = got["planes"][k].ctypes.data
            out[4] = got["loci"].ctypes.data
        cb = _PLANES_ALLOC(alloc)
        n = self._lib.smc_bam_planes(self._h, chrom.encode(), lo, hi, max_reads, float(params.mismatchThr), refb,
                                     int(nthreads), int(params.ds), int(params.minBQ), int(params.minMQ),
                                     int(params.primerDist), cb, None,
                                     C.byref(done), C.byref(n_slots), C.byref(n_us))
        if n < 0:
            msg = self._lib.smc_bam_error(self._h).decode()
            raise (PileupError if n <= -6 else BamError)(msg)
        nl = done.value
        planes, loci = got["planes"], got["loci"]
        ustart = np.empty(n_us.value, np.uint32)
        n_keys = np.empty(nl, np.int32)
        keys = C.create_string_buffer(max(1, self._lib.smc_bam_keys_len(self._h)))
        ptr = lambda a: a.ctypes.data_as(C.c_void_p)
        self._lib.smc_bam_planes_copy(self._h, ptr(ustart), ptr(n_keys), C.cast(keys, C.c_void_p))
        # the reference's down-sampling (smCounter.py:496-498) on loci over the barcode cap
        info = self._lib.smc_bam_ds_info(self._h).decode()
        if info:
            from .features import LF_SAMPLED, USTART_DROPPED
            from .py2compat import py2_downsample_barcodes
            for line in info.splitlines():
                parts = line.split("\t")
                l = int(parts[0])
                ids, names = zip(*[(int(x.split(":", 1)[0]), x.split(":", 1)[1]) for x in parts[1:]])
                kept = set(py2_downsample_barcodes(str(lo + l + 1), list(names), params.ds))
                o = int(loci["umi_off"][l])
                for u, name in zip(ids, names):
                    if name not in kept:
                        ustart[o + u] |= USTART_DROPPED
                loci["flags"][l] |= LF_SAMPLED
        return nl, planes, ustart, loci, self._tables(nl, n_keys, keys, chrom, lo, fasta)


def iter_pileup_batches_native(path: str, fasta, loci: Sequence[Tuple[str, str]], max_reads: int = 2_000_000):
    """Same chunks as iter_pileup_batches, decoded by the native library."""
    bam = NativeBam(path)
    i, n = 0, len(loci)
    while i < n:
        first = i
        parts, offs, chroms, poss, refs, tables = [], [np.zeros(1, np.int64)], [], [], [], []
        total = 0
        while i < n and total < max_reads:
            chrom = loci[i][0]
            j = i
            while j + 1 < n and loci[j + 1][0] == chrom and int(loci[j + 1][1]) == int(loci[j][1]) + 1 \
                    and j + 1 - i < 4096:
                j += 1
            lo, hi = int(loci[i][1]) - 1, int(loci[j][1])
            nl, cols, off, tb = bam.pileup_run(chrom, lo, hi, max_reads - total, fasta)
            parts.append(cols)
            offs.append(off[1:] + total)
            total += int(off[-1])
            chroms += [chrom] * nl
            poss += list(range(lo + 1, lo + 1 + nl))
            run_ref = fasta.fetch(chrom, lo, lo + nl).upper()
            refs += [run_ref[k:k + 1] for k in range(nl)]
            tables += tb
            i += nl
        yield first, PileupBatch(chrom=chroms, pos=np.array(poss, np.int64), ref=refs, alleles=tables,
                                 read_off=np.concatenate(offs),
                                 **{k: np.concatenate([p[k] for p in parts]) for k, _ in _COLS})
    bam.close()


_ARENA_SLACK = 65536          # room for the overshoot of a batch's last locus and the 4-read padding (tests shrink it)


def host_threads(per_node: int = 1) -> int:
    """Threads the decoder's stages run on: SMC_HOST_THREADS, else the process's cores shared among the node's ranks, at most
    HOST_THREADS_MAX.  A stage of a decode is a millisecond or two of work and ends when its LAST thread does: on every logical CPU
    of a 256-thread host, one worker that the scheduler holds back (another process, a runtime thread) stalls the stage for a time
    slice - measured: stages of 1.5 ms taking 25-57 ms now and then; with 48 threads the stages are as fast and the stalls gone."""
    env = int(os.environ.get("SMC_HOST_THREADS", "0") or 0)
    if env > 0:
        return env
    return max(1, min(HOST_THREADS_MAX, len(os.sched_getaffinity(0)) // max(1, per_node)))


HOST_THREADS_MAX = 48


def iter_device_batches_native(path: str, fasta, loci: Sequence[Tuple[str, str]], params, max_reads: int = 2_000_000,
                               nthreads: int = 0):
    """BAM -> `features.DeviceBatch` chunks in one native pass (decode, per-read features, barcode-major
    order, padding): equals extract_features(iter_pileup_batches(...)) chunk for chunk."""
    from .features import DeviceBatch
    bam = NativeBam(path)
    nthreads = nthreads or host_threads()
    i, n = 0, len(loci)
    while i < n:
        first = i
        P, US, LC = [[] for _ in range(4)], [], []
        chroms, poss, refs, tables = [], [], [], []
        total = slots = n_us = 0
        # the runs of a batch are written one after the other into one set of arrays (untouched pages cost nothing);
        # a run that does not fit - the last locus of a batch may overshoot max_reads - gets its own arrays and the
        # batch is concatenated as before
        arena = [np.empty(max_reads + (max_reads >> 3) + _ARENA_SLACK, np.uint32) for _ in range(4)]
        in_arena = True
        while i < n and total < max_reads:
            chrom = loci[i][0]
            j = i
            while j + 1 < n and loci[j + 1][0] == chrom and int(loci[j + 1][1]) == int(loci[j][1]) + 1 \
                    and j + 1 - i < 4096:
                j += 1
            lo, hi = int(loci[i][1]) - 1, int(loci[j][1])
            run_ref = fasta.fetch(chrom, lo, hi).upper()
            nl, planes, ustart, lc, tb = bam.planes_run(chrom, lo, hi, max_reads - total, params, run_ref, nthreads, fasta,
                                                        arena if in_arena else None, slots)
            in_arena = in_arena and (len(planes[0]) == 0 or planes[0].base is arena[0])
            lc["read_off4"] += slots // 4
            lc["umi_off"] += n_us
            for k in range(4):
                P[k].append(planes[k])
            US.append(ustart)
            LC.append(lc)
            slots += len(planes[0])
            n_us += len(ustart)
            total += int(lc["n_reads"].sum())
            chroms += [chrom] * nl
            poss += list(range(lo + 1, lo + 1 + nl))
            refs += [run_ref[k:k + 1] for k in range(nl)]
            tables += tb
            i += nl
        cat = lambda parts: parts[0] if len(parts) == 1 else np.concatenate(parts)
        if in_arena:
            P = [[a[:slots]] for a in arena]
        yield first, DeviceBatch(loci=cat(LC), meta=cat(P[0]), umi=cat(P[1]), frag=cat(P[2]), dist=cat(P[3]),
                                 umi_start=cat(US), chrom=chroms, pos=np.array(poss, np.int64), ref=refs, alleles=tables)
    bam.close()
