#!/bin/bash
# What binds k_call_v2: busy / stall counters of the SQ, TCP / TCC on a short bench.py run (every pass under its own timeout; the
# TA / TD / GRBM group hung the profiler once and is left out).  usage: bash scripts/r03_call_pmc.sh TAG [CFG]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-call_pmc}; CFG=${2:-C3}; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
SHORT="--steps 5 --warmup 2 --blocks 1 --no-cpu-baseline --no-parity --no-other-configs --no-from-alignments --config $CFG"
i=0
for set in \
  "SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" \
  "SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
  "SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_IFETCH SQ_BUSY_CU_CYCLES" \
  "SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SMEM SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_FLAT SQ_CYCLES SQ_INSTS" \
  "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" \
  "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum" \
  "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum" \
  "TCC_TAG_STALL_sum TCC_REQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_BUSY_sum" \
  "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 240 rocprofv3 --pmc $set --output-format csv -d $O/p$i -- python3 $R/bench.py $SHORT > /dev/null 2>&1 || echo "pass $i failed / timed out"
done
python3 $R/scripts/pmc_summary.py $O/p* > $O/pmc_summary.txt
find $O -name "*.csv" -size +300k -delete
grep -A70 "k_call_v2" $O/pmc_summary.txt | head -80
