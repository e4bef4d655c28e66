#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05j; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_devplanes.py tests/test_gpu_from_alignments.py tests/test_bam_golden.py -m gpu -x -q > $O/pytest.txt 2>&1
tail -6 $O/pytest.txt
for c in C3 C5 EX X3 C2; do
  timeout 300 python3 -m bench_fa --config $c --steps 10 --warmup 3 --blocks 3 --parity-loci 0 --slots 1 > $O/fa_$c.json 2>/dev/null
  python3 -c "
import json,sys
d=json.load(open('$O/fa_$c.json'))
print('$c', round(d['value']/1e6,3), 'M loci/s', round(d['ms_per_step'],3), 'ms; emit2', round(d['k_bp_emit2_ms'],3), 'call', round(d['k_call_v2_ms'],3))"
done
SMC_ALLOC_TRIES=1 timeout 300 python3 scripts/r05_part_step.py > $O/part_step.txt 2>&1; cat $O/part_step.txt
