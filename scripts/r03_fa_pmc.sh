#!/bin/bash
# PMC passes over the from_alignments leg (smcounter_amd.fa_leg) at a given size.  usage: bash scripts/r03_fa_pmc.sh TAG LOCI
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-fa_pmc}; N=${2:-200000}; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; export PYTHONPATH=$R
ARGS="-m smcounter_amd.fa_leg --config C3 --loci $N --steps 2 --warmup 1 --blocks 1 --parity-loci 0"
rocprofv3 -L > $O/counters.txt 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $ARGS > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $ARGS > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/sq -- python3 $ARGS > /dev/null 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM --output-format csv -d $O/sq2 -- python3 $ARGS > /dev/null 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d $O/tcc -- python3 $ARGS > /dev/null 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_STALL_sum TCC_WRITE_sum --output-format csv -d $O/tcc2 -- python3 $ARGS > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $ARGS > /dev/null 2>&1
python3 $R/scripts/pmc_summary.py $O/fetch $O/write $O/sq $O/sq2 $O/tcc $O/tcc2 > $O/pmc_summary.txt
python3 $R/scripts/kt_summary.py $O/kt > $O/kernels.txt
find $O -name "*.csv" -size +300k -delete
grep -A40 "k_bp_tiles<true>" $O/pmc_summary.txt | head -60; head -12 $O/kernels.txt
