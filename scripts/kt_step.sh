#!/bin/bash
# per-kernel durations of ONE shape's steps, one step at a time (rocprofv3 --kernel-trace of `python3 -m bench_fa --slots 1`):
# usage: scripts/kt_step.sh C2 gpurun_out/kt_C2   -> <dir>/by_grid.txt (kt_summary.py) and <dir>/one_step.txt (kt_gaps.py)
cfg=$1; out=$2; R=$GRAFT_REPO_ROOT
mkdir -p $R/$out
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/$out/kt -- python3 $R/bench_fa.py --config $cfg --steps 6 --warmup 2 --blocks 1 --slots 1 --parity-loci 0 > $R/$out/leg.json 2>/dev/null
python3 $R/scripts/kt_summary.py $R/$out/kt > $R/$out/by_grid.txt
python3 $R/scripts/kt_gaps.py $R/$out/kt > $R/$out/one_step.txt 2>&1
find $R/$out -name "*.csv" -size +300k -delete
cat $R/$out/one_step.txt | head -60
