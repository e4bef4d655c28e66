#!/bin/bash
# Everything profiles/r05_* is assembled from, in one GPU call (every profiler run under its own timeout).
# usage: bash scripts/collect_round5.sh [TAG]   (writes gpurun_out/TAG/, default r05final; then scripts/assemble_profiles_r05.py TAG)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r05final}; mkdir -p $O
cd $R
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
# the library's own placement against the bench-side trials of round 4, three fresh processes each way
for i in 1 2 3; do
  timeout 400 python bench.py --no-cpu-baseline --no-other-configs --no-parity > $O/place0_$i.json 2>/dev/null
  timeout 400 python bench.py --no-cpu-baseline --no-other-configs --no-parity --place 30 > $O/place30_$i.json 2>/dev/null
done
timeout 600 python bench.py --scaling strong --config C4 --no-cpu-baseline --no-other-configs --steps 10 --warmup 3 > $O/bench_C4_strong_n1.json 2>/dev/null
timeout 600 bash scripts/shapes_perf.sh 7 > $O/shapes.txt 2>&1
for spec in "2000 3000 60" "20000 1000 20" "500 58000 9" "2000 58000 9"; do
  set -- $spec
  timeout 600 python3 scripts/e2e_perf.py $1 $2 $3 2>&1 | grep -v -E "amdgpu|smc_bam|collect_reads" > $O/e2e_$1_$2.txt
done
cd /tmp; export TMPDIR=/tmp; export PYTHONPATH=$R
# kernel trace of the driver's command
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_under_trace.json 2>/dev/null
find $O/kt -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
python3 $R/scripts/kt_summary.py $O/kt > $O/kernel_trace_by_grid.txt
# the step's kernels alone (the leg bench.py's headline times): trace by grid, timeline of one step, counters of the walk and of the locus kernel
FA="-m bench_fa --config C3 --steps 6 --warmup 2 --blocks 1 --parity-loci 0 --slots 1"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/fa_kt -- python3 $FA > $O/fa_under_trace.json 2>/dev/null
python3 $R/scripts/kt_summary.py $O/fa_kt > $O/fa_kernels.txt
python3 $R/scripts/kt_gaps.py $O/fa_kt > $O/fa_timeline.txt
for c in C5 X3 EX C2; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/fa_kt_$c -- python3 -m bench_fa --config $c --steps 6 --warmup 2 --blocks 1 --parity-loci 0 --slots 1 > /dev/null 2>&1
  python3 $R/scripts/kt_gaps.py $O/fa_kt_$c > $O/fa_timeline_$c.txt 2>&1
done
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" \
  "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum" \
  "TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum TCC_BUSY_avr" \
  "TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" \
  "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
  "SQ_BUSY_CYCLES SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $O/fa_p$i -- python3 $FA > /dev/null 2>&1 || echo "fa pass $i failed"
done
python3 $R/scripts/pmc_summary.py $O/fa_p* > $O/fa_pmc_summary.txt
find $O -name "*.csv" -size +300k -delete
cd $R; SMC_BENCH_SHARE_GPU=1 timeout 600 python bench.py --gpus 2 --steps 10 --warmup 3 --loci-per-gpu 50000 --no-other-configs --no-cpu-baseline > $O/bench_2ranks_functional.json 2>/dev/null
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
sha256sum $R/smcounter_amd/libsmcounter_hip.so | cut -c1-16 > $O/lib_sha16.txt
tail -3 $O/shapes.txt; head -14 $O/fa_kernels.txt; tail -3 $O/fa_timeline.txt; cat $O/e2e_500_58000.txt | tail -8; wc -c $O/fa_pmc_summary.txt
