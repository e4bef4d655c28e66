#!/bin/bash
# Round profile of bench.py on the GPU box: kernel-trace stats, then HBM traffic counters in their own
# passes (FETCH_SIZE and WRITE_SIZE do not fit one pass; never mixed with trace domains).
# usage: bash scripts/profile_bench.sh gpurun_out/prof_rNN
out=$1
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
CMD="python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/kt -- $CMD > $R/$out.bench_under_trace.json 2>/dev/null
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$out/fetch -- $CMD > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/$out/write -- $CMD > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $R/$out/sq -- $CMD > /dev/null 2>&1
python3 $R/scripts/pmc_summary.py $R/$out > $R/$out/pmc_summary.txt
find $R/$out/kt -name "*kernel_stats.csv" -exec cp {} $R/$out/kernel_stats.csv \;
find $R/$out -name "*.csv" -size +300k -delete
cat $R/$out/kernel_stats.csv; grep -A12 "k_call_loci" $R/$out/pmc_summary.txt | head -14
