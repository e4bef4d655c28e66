#!/bin/bash
# does a run whose read words fit the 256 MiB Infinity Cache walk and call faster per locus? (C3's shape at 4k..200k loci; dev tool)
mkdir -p gpurun_out/mall
for n in 4000 8000 12000 16000 24000 48000 200000; do
  python3 -m bench_fa --config C3 --loci $n --slots 1 --steps 20 --warmup 3 --blocks 3 --parity-loci 0 > gpurun_out/mall/fa_$n.json 2> gpurun_out/mall/fa_$n.err
  python3 - <<PY
import json
d=json.loads([l for l in open("gpurun_out/mall/fa_$n.json") if l.startswith("{")][-1])
n=$n
print("loci %6d: step %.1f us (%.2f ns/locus), emit2 %.1f us (%.3f ns/locus), call %.1f us (%.3f ns/locus)" % (n, 1e3*d["ms_per_step"], 1e6*d["ms_per_step"]/n, 1e3*d["k_bp_emit2_ms"], 1e6*d["k_bp_emit2_ms"]/n, 1e3*d.get("k_call_v2_ms",0), 1e6*d.get("k_call_v2_ms",0)/n))
PY
done
