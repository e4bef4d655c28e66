#!/bin/bash
# usage: run_pmc.sh OUTDIR -- counter sets are separate passes (TCC slots; never mixed with traces)
out=$1; shift
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CMD="python3 $R/scripts/quick_perf.py --cfg ${PMC_CFG:-C3} --loci ${PMC_LOCI:-40000} --iters 3"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM --output-format csv -d $R/$out/p1 -- $CMD > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT --output-format csv -d $R/$out/p2 -- $CMD > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_I8 GRBM_GUI_ACTIVE --output-format csv -d $R/$out/p3 -- $CMD > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$out/p4 -- $CMD > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/$out/p5 -- $CMD > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/kt -- $CMD > /dev/null 2>&1
python3 $R/scripts/pmc_summary.py $R/$out | tee $R/$out/pmc_summary.txt
find $R/$out/kt -name "*kernel_stats.csv" -exec cat {} \; | tee $R/$out/kernel_stats.csv
find $R/$out -name "*.csv" -size +200k -delete
