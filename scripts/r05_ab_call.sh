#!/bin/bash
# two builds of the library on the locus kernels of C3 / X3 / C2 / C5 from alignments, fresh processes in turn (dev tool)
for i in 1 2; do
  for lib in libv_cur.so libsmcounter_hip.so; do
    for c in ${SHAPES:-C3 X3 C5}; do
      SMC_HIP_LIB=$PWD/smcounter_amd/$lib python3 -m bench_fa --config $c --slots 1 --steps 10 --blocks 3 --parity-loci 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$lib $c: step %.3f ms, emit2 %.3f, call %.3f' % (d['ms_per_step'], d['k_bp_emit2_ms'], d['k_call_v2_ms']))"
    done
  done
done
