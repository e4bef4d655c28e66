#!/bin/bash
# the 58,000x end-to-end fixture at several host thread counts: decode stage times of the three device-planes passes
for t in 16 32 48 64 128 256; do
  echo "== SMC_HOST_THREADS=$t"
  SMC_HOST_THREADS=$t SMC_BAM_TIMING=1 timeout 300 python3 scripts/e2e_perf.py 500 58000 9 2>&1 | grep -E "smc_bam_alignments|stages" | tail -6
done
