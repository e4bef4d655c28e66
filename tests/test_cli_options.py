"""Command-line plumbing the reference has (smCounter.py:616-640, :645-672, run_log.py): --paramFile, --runPath,
--logFile, dict arguments.  The GPU call is swapped for the CPU restatement: these tests are about the host path."""
import glob
import os
import sys

import pytest

import bam_fixture
from smcounter_amd import abi, bamio, cli, fasta, rows, runlog

import oracle_lib


@pytest.fixture
def cpu_cli(monkeypatch):
    def cpu_call_shard(args, params, loci, device, early=None):
        ref = fasta.FastaFile(args.refGenome)
        out = []
        for _, db in bamio.iter_device_batches_native(args.bamFile, ref, loci, params, max_reads=args.batchReads):
            R = oracle_lib.call_batch(db, abi.c_params(params), abi.ROW_DTYPE)
            out.extend(rows.format_rows(R, db, params, ref))
        return out
    monkeypatch.setattr(cli, "call_shard", cpu_call_shard)


def test_param_file_run_path_and_log_file(tmp_path, cpu_cli, monkeypatch):
    case = bam_fixture.make_case(str(tmp_path))
    work = tmp_path / "work"
    work.mkdir()
    pf = tmp_path / "params.txt"
    pf.write_text("\n".join(["--outPrefix=pf", "--bamFile=" + case["bam"], "--bedTarget=" + case["bed"], "--mtDepth=12",
                             "--rpb=3.0", "--hpLen=8", "--refGenome=" + case["fasta"], "--threshold=10",
                             "--runPath=" + str(work)]) + "\n")
    cwd = os.getcwd()
    try:
        ns = cli.build_parser().parse_args(["--outPrefix", "x", "--bamFile", "x", "--bedTarget", "x", "--mtDepth", "1",
                                            "--rpb", "1", "--paramFile", str(pf), "--logFile", str(tmp_path / "lg")])
        name = runlog.init(ns.logFile)                     # what the __main__ block does
        try:
            thr = cli.main(ns)
        finally:
            runlog.close()
    finally:
        os.chdir(cwd)
    assert thr == 10
    assert os.path.exists(work / "pf.smCounter.all.txt") and os.path.exists(work / "pf.smCounter.cut.vcf")
    log = open(name).read()
    assert "smCounter started at" in log and "begin variant filtering and output" in log and "('mtDepth', 12)" in log
    assert glob.glob(str(tmp_path / "lg.run-log_*.txt")) == [name]
    # dict arguments (the reference's main() is also called that way by its wrappers, smCounter.py:650-652)
    os.chdir(str(tmp_path))
    try:
        thr2 = cli.main(dict(outPrefix="d", bamFile=case["bam"], bedTarget=case["bed"], mtDepth=12, rpb=3.0, hpLen=8,
                             refGenome=case["fasta"]))
    finally:
        os.chdir(cwd)
    assert thr2 == int(round(14.0 + 0.012 * 12)) or thr2 > 0
    a = open(work / "pf.smCounter.all.txt").read().split("\n")
    b = open(tmp_path / "d.smCounter.all.txt").read().split("\n")
    assert a == b


def test_missing_reference_genome_is_refused(tmp_path, cpu_cli):
    case = bam_fixture.make_case(str(tmp_path))
    with pytest.raises(SystemExit):
        cli.main(dict(outPrefix=str(tmp_path / "o"), bamFile=case["bam"], bedTarget=case["bed"], mtDepth=12, rpb=3.0))
