#!/bin/bash
# the from-alignments leg on the bench's shapes, one line each (python3 -m bench_fa; every row checked unless PARITY=0):
# usage: [PARITY=0] scripts/legs.sh "C3 C5 X3 EX C2" [tag] [ENV=VALUE ...]
cfgs=${1:-C3 C5 X3 EX C2}; tag=${2:-legs}; shift 2
out=gpurun_out/${tag}.txt
mkdir -p gpurun_out; : > $out
for cfg in $cfgs; do
  env "$@" python3 -m bench_fa --config $cfg --steps ${STEPS:-10} --warmup 3 --blocks 3 --parity-loci ${PARITY:--1} 2>> gpurun_out/${tag}.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
p=d.get('parity',{})
print('$cfg ms_per_step %.4f one_at_a_time %.4f walk_ms %.4f call_ms %.4f Mloci/s %.2f 8d_frac %.3f status %s mism %s loci %s fisher %s' % (d['ms_per_step'], d['ms_per_step_one_at_a_time'], d['k_bp_emit2_ms'], d['k_call_v2_ms'], d['value']/1e6, d['whole_step_on_survey_8d']['frac'], d['builder_status'], p.get('mismatches'), p.get('loci'), p.get('fisher_tests_run')))
" >> $out 2>&1
done
cat $out
