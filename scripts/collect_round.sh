#!/bin/bash
# Everything profiles/r0N_* is assembled from, in one GPU call: the bench line, the shapes, the phase counts, the end-to-end
# runs (20,000 loci x 1000x; 2,000 x 3000x; the reference's example depth 500 x 58,000x), the host-buffer rate and the
# kernel trace of an end-to-end run.  usage: bash scripts/collect_round.sh   (writes gpurun_out/r02final/)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02final; mkdir -p $O
cd $R
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
bash scripts/shapes_perf.sh 7 > $O/shapes.txt 2>&1
bash scripts/phase_valu.sh gpurun_out/r02final/pv 2>&1 | grep ablate > $O/phase.txt
python3 scripts/e2e_perf.py 2000 3000 60 2>&1 | grep -v -E "amdgpu|smc_bam|collect_reads" > $O/e2e_2000.txt
python3 scripts/e2e_perf.py 20000 1000 20 2>&1 | grep -v -E "amdgpu|smc_bam|collect_reads" > $O/e2e_20000.txt
python3 scripts/e2e_perf.py 500 58000 9 2>&1 | grep -v -E "amdgpu|smc_bam|collect_reads" > $O/e2e_deep.txt
SMC_BAM_TIMING=1 python3 scripts/decode_alignments_perf.py 20000 1000 20 2>&1 | tail -5 > $O/decode.txt
python3 scripts/host_path_perf.py C3 200000 > $O/host_path.txt 2>&1
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_e2e -- python3 $R/scripts/e2e_perf.py 20000 1000 20 > /dev/null 2>&1
python3 $R/scripts/kt_summary.py $O/kt_e2e > $O/e2e_kernels.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_deep -- python3 $R/scripts/e2e_perf.py 500 58000 9 > /dev/null 2>&1
python3 $R/scripts/kt_summary.py $O/kt_deep > $O/e2e_deep_kernels.txt
find $O -name "*.csv" -size +300k -delete
tail -3 $O/shapes.txt; cat $O/phase.txt; cat $O/host_path.txt | tail -3; head -8 $O/e2e_kernels.txt
