#!/bin/bash
# the whole GPU suite, the from_alignments leg, the in-process end-to-end run
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-full}; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
timeout 300 python -m bench_fa --config C3 --loci 200000 --steps 10 --warmup 3 > $O/fa.txt 2>&1; grep -oE '"ms_per_step": [0-9.]+|"kernel_ms": [0-9.]+|"k_call_v2_ms": [0-9.]+|"mismatches": [0-9]+|"host_ms_per_step": \{[^}]*\}' $O/fa.txt | tr '\n' ' '; echo
timeout 600 python3 scripts/e2e_perf.py 20000 1000 20 2>&1 | grep -v -E "amdgpu|smc_bam|collect_reads" > $O/e2e.txt; grep -E "in process|stages|child" $O/e2e.txt | tail -6
