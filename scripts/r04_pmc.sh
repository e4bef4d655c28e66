#!/bin/bash
# PMC passes over the from_alignments leg with a given build of the library.
# usage: bash scripts/r04_pmc.sh TAG LIBNAME LOCI "COUNTERS OF PASS 1" "COUNTERS OF PASS 2" ...
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; LIB=$2; N=$3; shift 3; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; export PYTHONPATH=$R; export SMC_HIP_LIB=$R/smcounter_amd/$LIB
ARGS="-m bench_fa --config C3 --loci $N --steps 2 --warmup 1 --blocks 1 --parity-loci 0"
i=0; dirs=""
for P in "$@"; do
  i=$((i+1))
  timeout 240 rocprofv3 --pmc $P --output-format csv -d $O/p$i -- python3 $ARGS > /dev/null 2>&1 || echo "pass $i failed/timeout"
  dirs="$dirs $O/p$i"
done
python3 $R/scripts/pmc_summary.py $dirs > $O/pmc_summary.txt
find $O -name "*.csv" -size +300k -delete
awk '/k_bp_emit2/{f=1} f&&/^k_|^void|^__/{if(!/k_bp_emit2/)exit} f' $O/pmc_summary.txt
