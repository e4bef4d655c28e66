"""BAM golden fixtures (tests/golden/bam_*.npz, made by tests/golden/make_bam_golden.py in the build container): the bytes of
a BAM and what the REFERENCE ITSELF returned for every target locus when its pysam calls were served from that BAM's
records.  Everything between the BAM and the row string is checked against those strings:
  CPU:  native decoder + host plane builder -> CPU restatement (oracle/smc_oracle.c) -> rows.format_rows;
        the readable Python decoder -> features.extract_features -> the same;
  GPU:  smc_bam_alignments + k_build_planes (device-built planes) -> k_call_v2 / k_filter_loci -> strings;
        host-built planes through the same kernels.
Loci the generator flagged tie_ambiguous (the reference's own choice there depends on py2 dict order of an indel key) are
compared on everything but the ALT-dependent columns."""
import json
import os
import sys

import numpy as np
import pytest

from conftest import ROOT
from smcounter_amd import abi, bamio, fasta, features, pileup, rows
from smcounter_amd.params import VcParams

GOLDEN = os.path.join(ROOT, "tests", "golden")
CASES = ("bam_cigars", "bam_deep", "bam_overcap", "bam_deep25k", "bam_unflagged")


def load_case(name, tmp_path):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    meta = json.loads(bytes(z["meta"]).decode())
    bam, fa_path = str(tmp_path / (name + ".bam")), str(tmp_path / (name + ".fa"))
    open(bam, "wb").write(bytes(z["bam"]))
    open(bam + ".bai", "wb").write(bytes(z["bai"]))
    open(fa_path, "wb").write(bytes(z["fasta"]))
    loci = [tuple(x) for x in meta["loci"]]
    return bam, fasta.FastaFile(fa_path), loci, VcParams(**meta["params"]), meta["expected"]


def assert_rows(text, expected, what):
    assert len(text) == len(expected)
    bad = []
    for i, (t, e) in enumerate(zip(text, expected)):
        if t == e["row"]:
            continue
        if e["tie_ambiguous"]:
            # CHROM POS and the allele-independent columns (DP FR MT UFR UMT ... the per-base columns) still have to agree
            a, b = t.split("\t"), e["row"].split("\t")
            keep = [0, 1, 5, 6, 7, 8, 9] + list(range(16, 44))          # (REF too follows the chosen allele: a deletion's bases)
            if len(a) == len(b) and all(a[k] == b[k] for k in keep):
                continue
        bad.append((i, t, e["row"]))
    assert not bad, "%s: %d rows differ from the reference's; first: %r" % (what, len(bad), bad[0])


@pytest.mark.parametrize("name", CASES)
def test_native_decoder_host_builder_and_cpu_restatement_against_the_reference(name, tmp_path):
    import oracle_lib
    bam, fa, loci, P, expected = load_case(name, tmp_path)
    text = []
    for _, db in bamio.iter_device_batches_native(bam, fa, loci, P, max_reads=40_000, nthreads=2):
        got = oracle_lib.call_batch(db, abi.c_params(P), abi.ROW_DTYPE)
        text += rows.format_rows(got, db, P, fa)
    assert_rows(text, expected, "native decoder -> smc_bam_planes -> smc_oracle.c")
    if name == "bam_overcap":
        assert all(e["sampled"] for e in expected)
    if name == "bam_deep":
        assert max(e["depth"] for e in expected) > 8192


@pytest.mark.parametrize("name", ("bam_cigars", "bam_overcap"))
def test_python_decoder_and_feature_extraction_against_the_reference(name, tmp_path):
    import oracle_lib
    bam, fa, loci, P, expected = load_case(name, tmp_path)
    pb = pileup.concat([b for _, b in bamio.iter_pileup_batches(bamio.BamFile(bam), fa, loci)])
    db = features.extract_features(pb, P)
    got = oracle_lib.call_batch(db, abi.c_params(P), abi.ROW_DTYPE)
    assert_rows(rows.format_rows(got, db, P, fa), expected, "bamio (Python) -> extract_features -> smc_oracle.c")


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_device_built_planes_and_kernels_against_the_reference(name, tmp_path, engine0):
    """BAM -> smc_bam_alignments -> k_build_planes -> k_call_v2 / k_filter_loci -> row strings == the reference's, on CIGARs
    with S/M/I/D/N/H, D next to I, two chromosomes, a locus deeper than 8192 reads, and loci over the barcode cap."""
    from smcounter_amd import devplanes, vc
    bam, fa, loci, P, expected = load_case(name, tmp_path)
    # with the four raw-field planes beside the words (32-bit words), and the words alone - the command line's way: 16-bit words
    # (smc_build_planes_w16 -> smc_plan_run_words16) until a run has no room in them, 32-bit from there on
    bits_seen = set()
    for all_planes in (True, False):
        engine0.word_bits = 16
        text, n_dev = [], 0
        for _, rb in devplanes.iter_resident_batches(bam, fa, loci, P, engine0, max_reads=40_000, nthreads=2, all_planes=all_planes):
            text += list(vc.vc_resident(rb, P, fa, engine0))
            n_dev += rb.n_device_runs
            bits_seen.add((all_planes, rb.words.word_bits))
        assert n_dev > 0
        assert_rows(text, expected, "smc_bam_alignments -> k_build_planes -> k_call_v2 (all_planes=%s)" % all_planes)
    engine0.word_bits = 16
    print(name, "word bits used:", sorted(bits_seen))
    assert (True, 32) in bits_seen and ((False, 16) in bits_seen or (False, 32) in bits_seen)


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_host_built_planes_and_kernels_against_the_reference(name, tmp_path, engine0):
    bam, fa, loci, P, expected = load_case(name, tmp_path)
    text = []
    for _, db in bamio.iter_device_batches_native(bam, fa, loci, P, max_reads=40_000, nthreads=2):
        text += rows.format_rows(engine0.call_batch_host(db, P), db, P, fa)
    assert_rows(text, expected, "smc_bam_planes -> k_call_v2")
