#!/bin/bash
# kernel timelines of one from-alignments step on some shapes.  usage: r05_trace.sh TAG "C3 EX ..."
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; export PYTHONPATH=$R
for c in $2; do
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${c}_kt -- python3 -m bench_fa --config $c --steps 6 --warmup 2 --blocks 1 --parity-loci 0 --slots 1 > /dev/null 2>&1
python3 $R/scripts/kt_gaps.py $O/${c}_kt > $O/${c}_timeline.txt 2>&1
python3 $R/scripts/kt_summary.py $O/${c}_kt > $O/${c}_kernels.txt 2>&1
echo "== $c"; tail -24 $O/${c}_timeline.txt
done
find $O -name "*.csv" -size +300k -delete
