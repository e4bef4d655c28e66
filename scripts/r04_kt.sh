#!/bin/bash
# kernel trace of the from-alignments leg: per-kernel summary + the timeline of the last step.  usage: bash scripts/r04_kt.sh TAG LOCI
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-kt}; N=${2:-200000}; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; export PYTHONPATH=$R
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 -m bench_fa --config C3 --loci $N --steps 6 --warmup 2 --blocks 1 --parity-loci 0 > $O/leg.json 2>$O/leg.err
python3 $R/scripts/kt_summary.py $O/kt > $O/kernels.txt
python3 $R/scripts/kt_gaps.py $O/kt > $O/timeline.txt
find $O -name "*.csv" -size +300k -delete
cat $O/timeline.txt; head -30 $O/kernels.txt
