#!/bin/bash
# Dev tool: the hot path on every synthetic shape at its own size (what profiles/r0N_other_shapes.txt holds).
# usage: bash scripts/shapes_perf.sh [iters]
R=${GRAFT_REPO_ROOT:-.}
it=${1:-5}
for spec in C3:40000 C2:10000 C5:20000 X3:40000 X6:20000 X2:20000 X7:15000 X4:10000 X5:8000 X8:6000 X1:2000 X9:1000 X10:300; do
  c=${spec%%:*}; n=${spec##*:}
  python3 $R/scripts/quick_perf.py --cfg $c --loci $n --iters $it 2>&1 | grep -E "^$c:|Error|error" | head -2
done
