#!/bin/bash
# round 3: the rewritten plane builder - GPU tests, a soak, kernel traces of the two end-to-end shapes.  usage: bash scripts/r03_bp_check.sh TAG
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r3a}; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
timeout 600 python scripts/bam_soak.py 1000 60 > $O/soak.txt 2>&1; tail -2 $O/soak.txt
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_e2e -- python3 $R/scripts/e2e_perf.py 20000 1000 20 > $O/e2e_20000.txt 2>&1
python3 $R/scripts/kt_summary.py $O/kt_e2e > $O/e2e_kernels.txt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_deep -- python3 $R/scripts/e2e_perf.py 500 58000 9 > $O/e2e_deep.txt 2>&1
python3 $R/scripts/kt_summary.py $O/kt_deep > $O/e2e_deep_kernels.txt
find $O -name "*.csv" -size +300k -delete
grep -E "device planes|stages" $O/e2e_20000.txt | tail -4; head -30 $O/e2e_kernels.txt
grep -E "device planes|stages" $O/e2e_deep.txt | tail -4; head -30 $O/e2e_deep_kernels.txt
