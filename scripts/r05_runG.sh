#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05g; mkdir -p $O
cd $R
timeout 400 python3 scripts/r05_window_probe.py 200000 > $O/window.txt 2>&1
cat $O/window.txt
cd /tmp; export TMPDIR=/tmp; export PYTHONPATH=$R; export R05_PMC=1; export SMC_VMM_CHUNK_MB=0
i=0
for set in "TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum" \
  "TCC_NORMAL_WRITEBACK_sum TCC_NORMAL_EVICT_sum TCC_WRITEBACK_sum TCC_WRITE_sum" \
  "TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" \
  "TCC_EA0_WRREQ_WRITE_DRAM_sum TCC_EA0_RDREQ_DRAM_sum TCC_BUBBLE_sum TCC_CYCLE_sum" \
  "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $O/p$i -- python3 $R/scripts/r05_place_probe.py 200000 6 > $O/pmc_run$i.txt 2>&1 || echo "pass $i failed"
  echo "== pass $i: $(grep FINAL $O/pmc_run$i.txt)" >> $O/pmc_fast_slow.txt
  python3 $R/scripts/pmc_last_dispatches.py k_bp_emit2 2 $O/p$i >> $O/pmc_fast_slow.txt
done
find $O -name "*.csv" -size +300k -delete
cat $O/pmc_fast_slow.txt
cd $R
SMC_VMM_CHUNK_MB=0 timeout 400 python3 scripts/ab_build.py 200000 libv_base.so libv_w0x2000.so libv_w0x4000.so libv_w0xA000.so libv_w0xC000.so libv_w1.so libv_x2.so libv_b16.so > $O/ab_wide.txt 2>&1
cat $O/ab_wide.txt
