#!/bin/bash
# the deep class's geometry on the read-word kernels: EX (2000 loci x 58 k reads) under a few part / chunk sizes (rows do not depend on them)
R=$GRAFT_REPO_ROOT; cd $R
B="--config EX --steps 20 --warmup 3 --blocks 3 --no-cpu-baseline --no-other-configs --no-from-alignments --no-parity"
run() { echo -n "$1: "; env $1 timeout 300 python bench.py $B 2>/dev/null | python3 -c "import sys,json; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f ms  %.2f M loci/s' % (b['ms_per_step'], b['value']/1e6))"; }
run "SMC_X=0"
for p in 10240 15360 30720 61440; do run "SMC_DEEP_PART_READS=$p"; done
for f in 2688 8192; do run "SMC_DEEP_FCAP=$f"; done
run "SMC_DEEP_UCAP=1280"
run "SMC_DEEP_PART_READS=30720 SMC_DEEP_FCAP=8192"
