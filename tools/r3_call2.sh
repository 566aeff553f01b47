#!/bin/bash
# round 3, GPU call 2: L2 prefetch touches by the compute waves of the EXACT Linear (D3DP_X2_PFD) -- A/B on one box
O=gpurun_out/r3c2; mkdir -p $O
export TMPDIR=/tmp
V=$PWD/d3dp_amd/lib/variants
( D3DP_LIB=$V/libd3dp_pf4.so timeout 600 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "linear_split_f16 or qkv_linear_packed or (linear_all_epilogues and exact) or g3_full_width" 2>&1 | tail -5 ) > $O/pytest_pf4.log
for v in default pf4 pf8 pf2 pf4a; do
  L=$V/libd3dp_$v.so; [ $v = default ] && L=$PWD/d3dp_amd/lib/libd3dp_hip.so
  D3DP_LIB=$L timeout 300 python tools/gemm_bench.py --x2 --m 123930 --iters 15 > $O/gemm_$v.log 2>&1
done
for v in default pf4 pf8 pf2 pf4a default; do
  L=$V/libd3dp_$v.so; [ $v = default ] && L=$PWD/d3dp_amd/lib/libd3dp_hip.so
  D3DP_LIB=$L timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-leg --no-parity 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['kernel_ms_per_step']
print('$v', round(d['value'], 2), round(d['ms_per_step'], 1), {n: round(v) for n, v in k.items() if v > 50})" >> $O/bench.log
done
tail -n 12 $O/*.log | cut -c1-400
