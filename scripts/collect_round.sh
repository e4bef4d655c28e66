#!/bin/bash
# Everything profiles/rNN_* is assembled from, in one GPU call (every profiler run under its own timeout).
# usage: bash scripts/collect_round.sh [TAG]   (writes gpurun_out/TAG/, default r06final; then scripts/assemble_profiles.py TAG r06)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r06final}; mkdir -p $O
cd $R; ulimit -c 0
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
cp bench_detail.json $O/bench_detail.json
# the headline in two more fresh processes (the spread between processes on one box)
for i in 1 2; do
  timeout 400 python3 bench.py --no-cpu-baseline --no-other-configs --no-parity > $O/fresh_$i.json 2>/dev/null
done
timeout 600 python3 bench.py --scaling strong --config C4 --no-cpu-baseline --no-other-configs --steps 10 --warmup 3 > $O/bench_C4_strong_n1.json 2>/dev/null
cp bench_detail.json $O/bench_C4_strong_n1_detail.json
for spec in "2000 3000 60" "20000 1000 20" "500 58000 9" "2000 58000 9"; do
  set -- $spec
  timeout 600 python3 scripts/e2e_perf.py $1 $2 $3 2>&1 | grep -v -E "amdgpu|smc_bam|collect_reads" > $O/e2e_$1_$2.txt
done
cd /tmp; export TMPDIR=/tmp; export PYTHONPATH=$R
# kernel trace of the driver's command
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_under_trace.json 2>/dev/null
find $O/kt -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
python3 $R/scripts/kt_summary.py $O/kt > $O/kernel_trace_by_grid.txt
# the step's kernels alone, shape by shape: trace by grid, the timeline of one step
for c in C3 C5 X3 EX C2; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/fa_kt_$c -- python3 $R/bench_fa.py --config $c --steps 6 --warmup 2 --blocks 1 --parity-loci 0 --slots 1 > $O/fa_under_trace_$c.json 2>/dev/null
  python3 $R/scripts/kt_summary.py $O/fa_kt_$c > $O/fa_kernels_$c.txt
  python3 $R/scripts/kt_gaps.py $O/fa_kt_$c > $O/fa_timeline_$c.txt 2>&1
done
# counters: HBM-side request sizes and bytes for every shape (the traffic records), the instruction / wait counters on C3
for c in C3 C5 X3 EX C2; do
  i=0
  for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" \
    "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $set --output-format csv -d $O/pmc_${c}_$i -- python3 $R/bench_fa.py --config $c --steps 4 --warmup 2 --blocks 1 --parity-loci 0 --slots 1 > /dev/null 2>&1 || echo "$c pass $i failed"
  done
  python3 $R/scripts/pmc_summary.py $O/pmc_${c}_* > $O/pmc_summary_$c.txt
done
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
  "SQ_BUSY_CYCLES SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $O/sq_C3_$i -- python3 $R/bench_fa.py --config C3 --steps 4 --warmup 2 --blocks 1 --parity-loci 0 --slots 1 > /dev/null 2>&1 || echo "sq pass $i failed"
done
python3 $R/scripts/pmc_summary.py $O/sq_C3_* > $O/sq_summary_C3.txt
find $O -name "*.csv" -size +300k -delete
cd $R; SMC_BENCH_SHARE_GPU=1 timeout 600 python3 bench.py --gpus 2 --steps 10 --warmup 3 --loci-per-gpu 50000 --no-other-configs --no-cpu-baseline > $O/bench_2ranks_functional.json 2>/dev/null
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
sha256sum $R/smcounter_amd/libsmcounter_hip.so | cut -c1-16 > $O/lib_sha16.txt
tail -c 600 $O/bench.json; head -8 $O/fa_kernels_C3.txt; tail -2 $O/fa_timeline_C3.txt; tail -4 $O/e2e_500_58000.txt; wc -c $O/pmc_summary_*.txt
