#!/bin/bash
# (the builds: for n in 1 2 3 4 5 6: hipcc <build.HIPCC_FLAGS> -DSMC_ABLATE=$n -o smcounter_amd/libv_abl$n.so smcounter_amd/csrc/smcounter_hip.hip)
# k_call_v2 by phase on C3 / X3 from alignments (16-bit words): builds that return after phase N (SMC_ABLATE; rows are garbage, timing only)
for c in ${SHAPES:-C3 X3}; do
for lib in libv_abl1.so libv_abl2.so libv_abl3.so libv_abl5.so libv_abl4.so libv_abl6.so libv_abl7.so libsmcounter_hip.so; do
  SMC_HIP_LIB=$PWD/smcounter_amd/$lib python3 -m bench_fa --config $c --slots 1 --steps 10 --blocks 2 --parity-loci 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$c $lib: call %.3f ms (step %.3f)' % (d['k_call_v2_ms'], d['ms_per_step']))"
done
done
