#!/bin/bash
# how much of the from_alignments step the alignments with indels cost: the same leg at indel rates 0, 2 % (default), 8 %
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-indel}; mkdir -p $O; cd $R
for r in 0.02 0 0.08 0.02; do
  SMC_FA_INDEL_RATE=$r timeout 300 python -m bench_fa --config C3 --loci 200000 --steps 10 --warmup 3 --parity-loci 0 > $O/fa_$r.txt 2>&1
  echo "rate $r: $(grep -oE '"ms_per_step": [0-9.]+|"kernel_ms": [0-9.]+|pileup reads' $O/fa_$r.txt | tr '\n' ' ')"
done
timeout 300 python -m bench_fa --config C3 --loci 200000 --steps 10 --warmup 3 > $O/fa_par.txt 2>&1
echo "default with parity: $(grep -oE '"ms_per_step": [0-9.]+|"kernel_ms": [0-9.]+|pileup reads' $O/fa_par.txt | tr '\n' ' ')"
rocm-smi --showclocks 2>/dev/null | grep -E "sclk|mclk" | head -4
