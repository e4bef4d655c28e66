#!/bin/bash
# Dev tool: scripts/ab_variants.py over several shapes.  usage: bash scripts/ab_shapes.sh "C3:40000 X1:2000" libA.so libB.so ...
specs=$1; shift
for spec in $specs; do
  c=${spec%%:*}; n=${spec##*:}
  echo "== $c $n"
  python3 ${GRAFT_REPO_ROOT:-.}/scripts/ab_variants.py $c $n "$@" 2>&1 | grep -E "median|identical to first variant: False"
done
