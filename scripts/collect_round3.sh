#!/bin/bash
# Everything profiles/r03_* is assembled from, in one GPU call (every profiler run under its own timeout).
# usage: bash scripts/collect_round3.sh   (writes gpurun_out/r03final/)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r03final}; mkdir -p $O
cd $R
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
timeout 600 bash scripts/shapes_perf.sh 7 > $O/shapes.txt 2>&1
for spec in "2000 3000 60" "20000 1000 20" "500 58000 9"; do
  set -- $spec
  timeout 600 python3 scripts/e2e_perf.py $1 $2 $3 2>&1 | grep -v -E "amdgpu|smc_bam|collect_reads" > $O/e2e_$1.txt
  timeout 300 python3 scripts/bp_perf.py $1 $2 $3 2>&1 | tail -2 > $O/bp_$1.txt
done
cd /tmp; export TMPDIR=/tmp; export PYTHONPATH=$R
# kernel trace of the driver's command
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_under_trace.json 2>/dev/null
find $O/kt -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
python3 $R/scripts/kt_summary.py $O/kt > $O/kernel_trace_by_grid.txt
# HBM traffic of k_call_v2 (C3, C5, C2) and how the read requests are sized (the FETCH_SIZE x 2 question)
SHORT="--steps 5 --warmup 2 --blocks 1 --no-cpu-baseline --no-parity --no-other-configs --no-from-alignments"
for cfg in C3 C5 C2; do
  timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch_$cfg -- python3 $R/bench.py $SHORT --config $cfg > /dev/null 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write_$cfg -- python3 $R/bench.py $SHORT --config $cfg > /dev/null 2>&1
  echo "== $cfg" >> $O/pmc_summary.txt
  python3 $R/scripts/pmc_summary.py $O/fetch_$cfg $O/write_$cfg >> $O/pmc_summary.txt
done
timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d $O/rdreq_C3 -- python3 $R/bench.py $SHORT --config C3 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/wrreq_C3 -- python3 $R/bench.py $SHORT --config C3 > /dev/null 2>&1
echo "== C3 request sizes" >> $O/pmc_summary.txt
python3 $R/scripts/pmc_summary.py $O/rdreq_C3 $O/wrreq_C3 >> $O/pmc_summary.txt
# the from_alignments leg: traffic and SQ counters of the plane builder's kernels
FA="-m bench_fa --config C3 --steps 2 --warmup 1 --blocks 1 --parity-loci 0"
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" \
  "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
  "SQ_BUSY_CYCLES SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $O/fa_p$i -- python3 $FA > /dev/null 2>&1
done
python3 $R/scripts/pmc_summary.py $O/fa_p* > $O/fa_pmc_summary.txt
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/fa_kt -- python3 $FA > /dev/null 2>&1
python3 $R/scripts/kt_summary.py $O/fa_kt > $O/fa_kernels.txt
# kernel traces of two end-to-end runs
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_e2e -- python3 $R/scripts/e2e_perf.py 20000 1000 20 > /dev/null 2>&1
python3 $R/scripts/kt_summary.py $O/kt_e2e > $O/e2e_kernels.txt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_deep -- python3 $R/scripts/e2e_perf.py 500 58000 9 > /dev/null 2>&1
python3 $R/scripts/kt_summary.py $O/kt_deep > $O/e2e_deep_kernels.txt
find $O -name "*.csv" -size +300k -delete
tail -3 $O/shapes.txt; head -12 $O/fa_kernels.txt; grep -A4 "k_call_v2" $O/pmc_summary.txt | head -40
