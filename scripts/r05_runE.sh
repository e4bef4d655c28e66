#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05e; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_devplanes.py tests/test_gpu_from_alignments.py tests/test_bam_golden.py -m gpu -x -q > $O/pytest.txt 2>&1
tail -4 $O/pytest.txt
timeout 300 python3 scripts/r05_xcd_group.py 200000 > $O/xcd_group.txt 2>&1
cat $O/xcd_group.txt
for c in C5 EX X3; do
  timeout 300 python3 -m bench_fa --config $c --steps 10 --warmup 3 --blocks 3 --parity-loci 0 --place 0 --slots 1 > $O/fa_$c.json 2>/dev/null
  python3 -c "
import json,sys
d=json.load(open('$O/fa_$c.json'))
print('$c', round(d['value']/1e6,3), 'M loci/s', round(d['ms_per_step'],3), 'ms; emit2', round(d['k_bp_emit2_ms'],3), 'call', round(d['k_call_v2_ms'],3))"
done
cd /tmp; export TMPDIR=/tmp; export PYTHONPATH=$R
for c in EX C5; do
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${c}_kt -- python3 -m bench_fa --config $c --steps 6 --warmup 2 --blocks 1 --parity-loci 0 --place 0 --slots 1 > /dev/null 2>&1
python3 $R/scripts/kt_gaps.py $O/${c}_kt > $O/${c}_timeline.txt
tail -22 $O/${c}_timeline.txt
done
find $O -name "*.csv" -size +300k -delete
