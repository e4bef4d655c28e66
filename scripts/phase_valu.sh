#!/bin/bash
# Dev tool: VALU / SALU wave-instruction counts of k_call_v2 per phase, from ablation builds (SMC_ABLATE=n returns
# after phase n) under rocprofv3 --pmc.  usage: [CFG=X9 LOCI=1000] bash scripts/phase_valu.sh gpurun_out/phase_valu
out=$1
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
F="-O3 --offload-arch=gfx950 -fPIC -shared -std=c++17 -ffp-contract=off -mllvm -disable-machine-licm -I$R/include"
cp $R/smcounter_amd/libsmcounter_hip.so /tmp/lib_keep.so
for ab in ${ABLATES:-1 2 3 4 0}; do
  hipcc $F -DSMC_ABLATE=$ab -o $R/smcounter_amd/libsmcounter_hip.so $R/smcounter_amd/csrc/smcounter_hip.hip
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d $R/$out/a$ab -- python3 $R/scripts/quick_perf.py --cfg ${CFG:-C3} --loci ${LOCI:-40000} --iters 1 > /dev/null 2>&1
  python3 - $R/$out/a$ab $ab <<'PY'
import sys, glob, csv, collections
d, ab = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_call_v2" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
import os
nl = int(os.environ.get("LOCI", "40000"))
print("ablate %s:" % ab, {k: round(sum(v) / len(v) / nl) for k, v in sorted(acc.items())}, "per locus")
PY
done
cp /tmp/lib_keep.so $R/smcounter_amd/libsmcounter_hip.so
find $R/$out -name "*.csv" -size +100k -delete
