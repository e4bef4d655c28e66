#!/bin/bash
# same-box A/B of builds of the HIP library (boxes differ by ~15 % on the plane builder): from_alignments C3 with each, interleaved.
# usage: bash scripts/r03_ab_libs.sh TAG lib1.so lib2.so ...   (paths under smcounter_amd/)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; shift; mkdir -p $O; cd $R
for round in 1 2; do
  for lib in "$@"; do
    SMC_HIP_LIB=$R/smcounter_amd/$lib timeout 300 python -m bench_fa --config C3 --loci 200000 --steps 10 --warmup 3 --parity-loci 0 > $O/${lib%.so}_$round.txt 2>&1
    echo "$lib round $round: $(grep -oE '"ms_per_step": [0-9.]+|"kernel_ms": [0-9.]+|"k_call_v2_ms": [0-9.]+' $O/${lib%.so}_$round.txt | tr '\n' ' ')"
  done
done
