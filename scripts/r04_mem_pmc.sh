#!/bin/bash
# memory-side PMC passes over the from_alignments leg (bench_fa).  usage: bash scripts/r04_mem_pmc.sh TAG LOCI
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-mem_pmc}; N=${2:-200000}; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; export PYTHONPATH=$R
ARGS="-m bench_fa --config C3 --loci $N --steps 2 --warmup 1 --blocks 1 --parity-loci 0"
run() { timeout 240 rocprofv3 --pmc $2 --output-format csv -d $O/$1 -- python3 $ARGS > /dev/null 2>&1 || echo "pass $1 failed/timeout"; }
run m1 "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum"
run m2 "TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum"
run m3 "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum"
run m4 "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_BUSY_avr"
run m5 "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"
run m6 "TCC_TAG_STALL_sum TCC_EA0_WRREQ_STALL_sum TCC_WRITE_sum TCC_READ_sum"
run m7 "GRBM_GUI_ACTIVE GRBM_COUNT"
python3 $R/scripts/pmc_summary.py $O/m1 $O/m2 $O/m3 $O/m4 $O/m5 $O/m6 $O/m7 > $O/pmc_summary.txt
find $O -name "*.csv" -size +300k -delete
grep -A40 "k_bp_emit" $O/pmc_summary.txt | head -60
