cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r2d
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2d/kt -- python3 $R/scripts/quick_perf.py --cfg X1 --loci 2000 --iters 3 > $R/gpurun_out/r2d/x1.txt 2>&1
find $R/gpurun_out/r2d/kt -name "*kernel_stats.csv" -exec cat {} \;
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/r2d/sq -- python3 $R/scripts/quick_perf.py --cfg X1 --loci 2000 --iters 1 > /dev/null 2>&1
python3 $R/scripts/pmc_summary.py $R/gpurun_out/r2d/sq | head -40
find $R/gpurun_out/r2d -name "*.csv" -size +200k -delete
