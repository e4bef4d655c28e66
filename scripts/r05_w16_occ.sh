#!/bin/bash
# the walk at 5 and at 4 wavefronts per SIMD (SMC_BP_LDS_PAD caps them), 32-bit and 16-bit read words (dev tool)
for spec in "32 0" "32 8192" "16 0" "16 8192" "16 12288"; do
  set -- $spec
  SMC_WORD_BITS=$1 SMC_BP_LDS_PAD=$2 python3 -m bench_fa --config C3 --slots 1 --steps 10 --blocks 3 --parity-loci 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('bits $1 lds_pad $2: step %.3f ms, emit2 %.3f, call %.3f' % (d['ms_per_step'], d['k_bp_emit2_ms'], d['k_call_v2_ms']))"
done
