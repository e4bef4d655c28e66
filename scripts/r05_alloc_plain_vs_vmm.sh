#!/bin/bash
# smc_mem_alloc_best with plain hipMalloc candidates (the default since the end of round 5) against virtual-memory ones (SMC_ALLOC_VMM=1):
# the walk's time in fresh processes (dev tool)
for i in 1 2 3; do
  for mode in plain vmm; do
    if [ $mode = vmm ]; then export SMC_ALLOC_VMM=1; else unset SMC_ALLOC_VMM; fi
    python3 -m bench_fa --config C3 --slots 1 --steps 10 --blocks 3 --parity-loci 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$mode $i: step %.3f ms, emit2 %.3f, call %.3f' % (d['ms_per_step'], d['k_bp_emit2_ms'], d['k_call_v2_ms']))"
  done
done
