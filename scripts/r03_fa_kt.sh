#!/bin/bash
# kernel trace of the from_alignments leg (C3, 200k loci)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-fa_kt}; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; export PYTHONPATH=$R
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 -m bench_fa --config C3 --loci 200000 --steps 5 --warmup 2 --blocks 1 --parity-loci 0 > $O/fa.txt 2>&1
python3 $R/scripts/kt_summary.py $O/kt > $O/kernels.txt
find $O -name "*.csv" -size +300k -delete
head -30 $O/kernels.txt
