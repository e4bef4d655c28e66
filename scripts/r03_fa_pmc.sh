#!/bin/bash
# PMC passes over the from_alignments leg (smcounter_amd.fa_leg) at a given size.  usage: bash scripts/r03_fa_pmc.sh TAG LOCI [quick]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-fa_pmc}; N=${2:-200000}; Q=${3:-}; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; export PYTHONPATH=$R
ARGS="-m bench_fa --config C3 --loci $N --steps 2 --warmup 1 --blocks 1 --parity-loci 0"
run() { timeout 240 rocprofv3 --pmc $2 --output-format csv -d $O/$1 -- python3 $ARGS > /dev/null 2>&1 || echo "pass $1 failed/timeout"; }
run sq "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
run sq2 "SQ_BUSY_CYCLES SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS"
run sq3 "SQ_IFETCH SQ_INSTS_BRANCH SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD"
if [ -z "$Q" ]; then
run fetch "FETCH_SIZE"
run write "WRITE_SIZE"
run tcc "TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"
run tcc2 "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_STALL_sum TCC_WRITE_sum"
fi
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $ARGS > /dev/null 2>&1
python3 $R/scripts/pmc_summary.py $O/fetch $O/write $O/sq $O/sq2 $O/sq3 $O/tcc $O/tcc2 > $O/pmc_summary.txt
python3 $R/scripts/kt_summary.py $O/kt > $O/kernels.txt
find $O -name "*.csv" -size +300k -delete
awk '/k_bp_emit/,/^k_bp_scan/' $O/pmc_summary.txt | head -40; head -8 $O/kernels.txt
