#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05c; mkdir -p $O
cd $R
timeout 400 python3 scripts/r05_chunk_sweep.py 200000 > $O/sweep1.txt 2>&1
timeout 400 python3 scripts/r05_chunk_sweep.py 200000 > $O/sweep2.txt 2>&1
SWEEP_CFG=C5 timeout 400 python3 scripts/r05_chunk_sweep.py 100000 > $O/sweep_c5.txt 2>&1
cd /tmp; export TMPDIR=/tmp; export PYTHONPATH=$R
# the step's timeline on the example run's shape (where does EX-from-alignments spend 2.2 ms?)
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ex_kt -- python3 -m bench_fa --config EX --steps 6 --warmup 2 --blocks 1 --parity-loci 0 --place 0 --slots 1 > $O/ex_under_trace.json 2>/dev/null
python3 $R/scripts/kt_summary.py $O/ex_kt > $O/ex_kernels.txt
python3 $R/scripts/kt_gaps.py $O/ex_kt > $O/ex_timeline.txt
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5_kt -- python3 -m bench_fa --config C5 --steps 4 --warmup 2 --blocks 1 --parity-loci 0 --place 0 --slots 1 > $O/c5_under_trace.json 2>/dev/null
python3 $R/scripts/kt_gaps.py $O/c5_kt > $O/c5_timeline.txt
# per-channel view of the write requests (json keeps the dimensions)
export R05_PMC=1
timeout 300 rocprofv3 --pmc TCC_EA0_WRREQ TCC_EA0_WRREQ_STALL --output-format json -d $O/pj -- python3 $R/scripts/r05_place_probe.py 200000 2 > $O/pj_run.txt 2>&1
ls -la $O/pj/*/ | head
find $O -name "*.csv" -size +300k -delete
cat $O/sweep1.txt; tail -40 $O/sweep2.txt; tail -40 $O/sweep_c5.txt; cat $O/ex_timeline.txt | tail -45
