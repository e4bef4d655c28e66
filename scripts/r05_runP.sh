#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05p; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; export PYTHONPATH=$R
i=0
for set in "TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_CYCLE_sum" \
  "TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_TAG_STALL_sum TCC_NORMAL_WRITEBACK_sum" \
  "TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCC_WRITE_sum TCC_NORMAL_EVICT_sum" \
  "TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_WRITE_DRAM_sum" \
  "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $O/p$i -- python3 $R/scripts/r05_probe_pmc.py 16 > $O/run$i.txt 2>&1 || echo "pass $i failed"
  echo "== pass $i: $(grep -E 'FINAL|probe times' $O/run$i.txt | tr '\n' ' ')" >> $O/fast_slow.txt
  python3 $R/scripts/pmc_last_dispatches.py k_mem_write_probe 8 $O/p$i >> $O/fast_slow.txt
done
find $O -name "*.csv" -size +300k -delete
cat $O/fast_slow.txt
