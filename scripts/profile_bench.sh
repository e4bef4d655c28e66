#!/bin/bash
# Round profile of bench.py on the GPU box.
#   1. rocprofv3 --kernel-trace --stats of the DRIVER'S command (bench.py --gpus 1 --steps 20 --warmup 5): per-kernel
#      durations that must agree with roofline.kernel_ms of the same run (the bench line under the trace is kept too);
#   2. HBM traffic counters in their own passes (FETCH_SIZE and WRITE_SIZE do not fit one pass; never mixed with trace
#      domains), for C3 (the metric's config), C5 and C2, on a short run of the same batch (one block of 5 steps);
#   3. SQ instruction / wait counters of the C3 run.
# usage: bash scripts/profile_bench.sh gpurun_out/prof_rNN
out=$1
R=$GRAFT_REPO_ROOT
mkdir -p $R/$out
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/kt -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $R/$out/bench_under_trace.json 2>/dev/null
find $R/$out/kt -name "*kernel_stats.csv" -exec cp {} $R/$out/kernel_stats.csv \;
python3 $R/scripts/kt_summary.py $R/$out/kt > $R/$out/kernel_trace_by_grid.txt
SHORT="--steps 5 --warmup 2 --blocks 1 --no-cpu-baseline --no-parity --no-other-configs"
for cfg in C3 C5 C2; do
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$out/fetch_$cfg -- python3 $R/bench.py $SHORT --config $cfg > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/$out/write_$cfg -- python3 $R/bench.py $SHORT --config $cfg > /dev/null 2>&1
  echo "== $cfg" >> $R/$out/pmc_summary.txt
  python3 $R/scripts/pmc_summary.py $R/$out/fetch_$cfg $R/$out/write_$cfg >> $R/$out/pmc_summary.txt
done
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $R/$out/sq -- python3 $R/bench.py $SHORT > /dev/null 2>&1
echo "== C3 SQ" >> $R/$out/pmc_summary.txt
python3 $R/scripts/pmc_summary.py $R/$out/sq >> $R/$out/pmc_summary.txt
find $R/$out -name "*.csv" -size +300k -delete
cat $R/$out/kernel_trace_by_grid.txt; grep -A3 k_call_v2 $R/$out/pmc_summary.txt
