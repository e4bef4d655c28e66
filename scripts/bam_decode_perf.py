import sys, time, tempfile
import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, R)
import bam_fixture, numpy as np
from smcounter_amd import bamio, bedops, fasta, features
from smcounter_amd.params import VcParams
d = tempfile.mkdtemp()
case = bam_fixture.make_case(d, n_umi=3000, frags_per_umi=6)
fa = fasta.FastaFile(case["fasta"]); loci = bedops.expand_loci(case["bed"])
P = VcParams(mtDepth=3000, rpb=6.0)
for nt in (1, 2, 4, 8, 16, 32):
    t=time.time(); got = list(bamio.iter_device_batches_native(case["bam"], fa, loci, P, nthreads=nt)); dt=time.time()-t
    n = sum(b.n_reads for _, b in got)
    print("fused native nthreads=%d: %d reads %.3fs -> %.1f M reads/s" % (nt, n, dt, n/dt/1e6))
