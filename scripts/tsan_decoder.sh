#!/bin/bash
# The decoder's parallel stages (concurrent interning table, record walk in pieces, kept-record / offset / window passes) under
# ThreadSanitizer: a C++ harness linked against csrc/smc_bam.cpp calls smc_bam_alignments three times with 8 threads on a 24,000-read
# BAM written by bamio.write_bam (CPU only).  Prints the result lines; any "WARNING: ThreadSanitizer" line is a finding.
R=$(cd $(dirname $0)/.. && pwd); T=${TMPDIR:-/tmp}/smc_tsan; mkdir -p $T
python3 - <<PY
import sys, numpy as np
sys.path.insert(0, "$R")
from smcounter_amd import bamio
rng = np.random.Generator(np.random.PCG64(5))
L = 30000
ref = "".join(rng.choice(list("ACGT"), size=L))
recs = []
for i in range(24000):
    pos = int(rng.integers(100, L - 400)); n = int(rng.integers(20, 250))
    recs.append(dict(tid=0, pos=pos, qname="r%d:n%d:UMI%d:x" % (i, i // 2, i % 97), flag=(0x40 if i % 2 == 0 else 0x80) | 1,
                     mapq=60, cigar=[(0, n)], seq=ref[pos:pos + n], qual=rng.integers(2, 41, size=n).astype(np.uint8).tolist(), nm=0))
recs.sort(key=lambda r: r["pos"])
bamio.write_bam("$T/t.bam", [("chrW", L)], recs); bamio.write_bai("$T/t.bam")
PY
sed "s#/tmp/tsan/t.bam#$T/t.bam#" $R/scripts/tsan_decoder.cpp > $T/h.cpp
g++ -O1 -g -fsanitize=thread -std=c++17 -pthread -I$R/include -o $T/h $T/h.cpp $R/smcounter_amd/csrc/smc_bam.cpp -lz -ldl && TSAN_OPTIONS="halt_on_error=0" $T/h 2>&1 | grep -E "WARNING|SUMMARY|reads " | sort | uniq -c
