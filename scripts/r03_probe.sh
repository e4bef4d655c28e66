#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-probe}; mkdir -p $O; cd $R
for i in 1 2 3; do timeout 600 python scripts/bp_addr_probe.py 200000 3 > $O/p$i.txt 2>&1; grep -E "copy|k_bp_emit|Error|error" $O/p$i.txt; rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|Power|Temperature \(Sensor junction|hotspot" | head -4; done
