for p in 128 192 256 320 384 512; do
  SMC_BP_PART=$p timeout 200 python -m bench_fa --config C3 --loci 200000 --steps 10 --warmup 3 --parity-loci 0 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('part $p', 'step', round(d['ms_per_step'],3), 'emit', round(d['roofline']['kernel_ms'],3))"
done
