#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05m; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; export PYTHONPATH=$R
for g in 2048 8192 32768; do
export SMC_BP_LIN_GRID=$g
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$g -- python3 -m bench_fa --config C3 --steps 6 --warmup 2 --blocks 1 --parity-loci 0 --slots 1 > /dev/null 2>&1
echo "grid $g: $(python3 $R/scripts/kt_summary.py $O/kt_$g | grep k_bp_lin)"
done
find $O -name "*.csv" -size +300k -delete
