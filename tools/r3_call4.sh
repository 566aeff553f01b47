#!/bin/bash
# round 3, GPU call 4: chunk-size sweep (tile-round quantisation) and small A/Bs of the h2i Linear (loader priority, cache policy)
O=gpurun_out/r3c4; mkdir -p $O
export TMPDIR=/tmp
V=$PWD/d3dp_amd/lib/variants
run() {  # name lib chunk
  L=$V/libd3dp_$2.so; [ $2 = default ] && L=$PWD/d3dp_amd/lib/libd3dp_hip.so
  D3DP_LIB=$L timeout 600 python bench.py --steps 2 --warmup 1 --chunk-seqs $3 --no-cpu-baseline --no-other-leg --no-parity 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['kernel_ms_per_step']
print('$1', round(d['value'], 2), round(d['ms_per_step'], 1), {n: round(v) for n, v in k.items() if v > 50})" >> $O/bench.log
}
run c30 default 30; run c47 default 47; run c59 default 59; run c71 default 71; run c79 default 79; run c30b default 30
run lp1 lp1 30; run lp3 lp3 30; run ant ant 30; run wnt wnt 30; run c30c default 30
cat $O/bench.log
