#!/bin/bash
# round 3, GPU call 9: temporal split-fp16 attention with both query tiles of a wave in one pass over K -- tests and A/B
O=gpurun_out/r3c9; mkdir -p $O
export TMPDIR=/tmp
V=$PWD/d3dp_amd/lib/variants
( timeout 600 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "attention or g3_full or g4_sampler or full_size_properties" 2>&1 | tail -6 ) > $O/pytest.log
run() {
  L=$V/libd3dp_$2.so; [ $2 = default ] && L=$PWD/d3dp_amd/lib/libd3dp_hip.so
  D3DP_LIB=$L timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-leg --no-parity 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['kernel_ms_per_step']
print('$1', round(d['value'], 2), round(d['ms_per_step'], 1), {n: round(v) for n, v in k.items() if v > 50})" >> $O/bench.log
}
run pair default; run nopair nopair; run pair default; run nopair nopair
tail -n 12 $O/*.log | cut -c1-400
