#!/bin/bash
# HBM-side counters of the from-alignments leg at a given LDS pad (wavefronts per CU of the walk).  usage: bash scripts/r04_traffic.sh TAG LOCI [PAD]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-traffic}; N=${2:-200000}; export SMC_BP_LDS_PAD=${3:-0}; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; export PYTHONPATH=$R
ARGS="-m bench_fa --config C3 --loci $N --steps 2 --warmup 1 --blocks 1 --parity-loci 0"
run() { timeout 240 rocprofv3 --pmc $2 --output-format csv -d $O/$1 -- python3 $ARGS > /dev/null 2>&1 || echo "pass $1 failed/timeout"; }
run fetch "FETCH_SIZE"
run write "WRITE_SIZE"
run m4 "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_BUSY_avr"
run m5 "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"
run m2 "TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum"
python3 $R/scripts/pmc_summary.py $O/fetch $O/write $O/m4 $O/m5 $O/m2 > $O/pmc_summary.txt
find $O -name "*.csv" -size +300k -delete
awk '/k_bp_emit2/{f=1;print;next} f&&/^[^ ]/{exit} f' $O/pmc_summary.txt
