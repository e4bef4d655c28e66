#!/bin/bash
# a variant build of the HIP library for side-by-side runs (scripts/ab_build.py).  usage: bash scripts/mkvariant.sh NAME.so -DFLAG ...
R=$(cd $(dirname $0)/.. && pwd); OUT=$R/smcounter_amd/$1; shift
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -shared -std=c++17 -ffp-contract=off -mllvm -disable-machine-licm -I$R/include -I$R/smcounter_amd/csrc "$@" -o $OUT $R/smcounter_amd/csrc/smcounter_hip.hip
