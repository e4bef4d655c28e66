#!/bin/bash
# per-kernel times of the hot path on some shapes.  usage: bash scripts/kt_shapes.sh "X3:40000 EX:2000" [lib.so]
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
[ -n "$2" ] && export SMC_HIP_LIB=$R/smcounter_amd/$2
for spec in $1; do
  c=${spec%%:*}; n=${spec##*:}; O=$R/gpurun_out/kts_$c; mkdir -p $O
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/scripts/quick_perf.py --cfg $c --loci $n --iters 5 > $O/out.txt 2>&1
  echo "== $c $n"; python3 $R/scripts/kt_summary.py $O/kt | head -8
  find $O -name "*.csv" -size +200k -delete
done
