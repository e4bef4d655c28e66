#!/bin/bash
# GPU call B of round 5: where do fast and slow allocations of the read words differ (timing, then counters), and what do the
# walk's stores cost - bytes or instructions (ablation builds)?
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05b; mkdir -p $O
cd $R
timeout 300 python3 scripts/r05_place_probe.py 200000 6 > $O/place.txt 2>&1
timeout 300 python3 scripts/r05_place_probe.py 200000 6 > $O/place2.txt 2>&1
timeout 300 python3 scripts/ab_build.py 200000 libv_base.so libv_b16.so libv_x2.so > $O/ab_store.txt 2>&1
cd /tmp; export TMPDIR=/tmp; export PYTHONPATH=$R; export R05_PMC=1
i=0
for set in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum" \
  "TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_TAG_STALL_sum" \
  "TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_LFIFO_FULL_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum" \
  "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum TCC_BUSY_avr" \
  "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
  "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum" \
  "GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $O/p$i -- python3 $R/scripts/r05_place_probe.py 200000 4 > $O/pmc_run$i.txt 2>&1 || echo "pass $i failed"
  echo "== pass $i: $(grep FINAL $O/pmc_run$i.txt)" >> $O/pmc_fast_slow.txt
  python3 $R/scripts/pmc_last_dispatches.py k_bp_emit2 2 $O/p$i >> $O/pmc_fast_slow.txt
done
find $O -name "*.csv" -size +300k -delete
cat $O/place.txt; echo; tail -14 $O/place2.txt; echo; cat $O/ab_store.txt; echo; cat $O/pmc_fast_slow.txt
