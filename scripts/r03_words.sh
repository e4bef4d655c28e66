#!/bin/bash
# the read-word layout: GPU tests, then from_alignments and the resident bench
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-words}; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -15 $O/pytest.txt
timeout 300 python -m smcounter_amd.fa_leg --config C3 --loci 200000 --steps 10 --warmup 3 > $O/fa.txt 2>&1; grep -oE '"ms_per_step": [0-9.]+|"kernel_ms": [0-9.]+|"k_call_v2_ms": [0-9.]+|"mismatches": [0-9]+' $O/fa.txt | tr '\n' ' '; echo
timeout 900 python bench.py --no-from-alignments --no-other-configs > $O/bench.txt 2>&1; tail -3 $O/bench.txt | cut -c1-1500
