#!/bin/bash
# round 3, GPU call 8: does the qkv Linear prefer shorter passes than the other Linears?  (pass-size sweep, exact mode)
O=gpurun_out/r3c8; mkdir -p $O
for c in 0 15 19 11 0; do
  timeout 300 python bench.py --steps 2 --warmup 1 --chunk-seqs $c --no-cpu-baseline --no-other-leg --no-parity 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['kernel_ms_per_step']
print('chunk $c:', round(d['value'], 2), round(d['ms_per_step'], 1), {n: round(v) for n, v in k.items() if v > 50})" >> $O/sweep.log
done
cat $O/sweep.log
