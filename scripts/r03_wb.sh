#!/bin/bash
# A/B of a plane-builder change: the builder's tests, then from_alignments C3 and the two bp_perf shapes.  usage: bash scripts/r03_wb.sh TAG
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-wb}; mkdir -p $O
cd $R
if [ -z "$SKIP_TESTS" ]; then timeout 900 python -m pytest tests/test_gpu_devplanes.py tests/test_bam_golden.py -m gpu -x -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt; fi
timeout 300 python -m smcounter_amd.fa_leg --config C3 --loci 200000 --steps 10 --warmup 3 > $O/fa.txt 2>&1; tail -12 $O/fa.txt
timeout 300 python scripts/bp_perf.py 20000 1000 > $O/bp1.txt 2>&1; tail -6 $O/bp1.txt
timeout 300 python scripts/bp_perf.py 500 58000 > $O/bp2.txt 2>&1; tail -6 $O/bp2.txt
