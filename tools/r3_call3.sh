#!/bin/bash
# round 3, GPU call 3: line-interleaved (h2i) operand layout of the EXACT Linear vs the two-plane layout -- tests + A/B
O=gpurun_out/r3c3; mkdir -p $O
export TMPDIR=/tmp
V=$PWD/d3dp_amd/lib/variants
( timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "linear or layernorm or attention or g2_tiny or g3_full or g4_sampler or cross_check or residual_adds" 2>&1 | tail -15 ) > $O/pytest.log
for v in default old; do
  L=$V/libd3dp_$v.so; [ $v = default ] && L=$PWD/d3dp_amd/lib/libd3dp_hip.so
  D3DP_LIB=$L timeout 300 python tools/gemm_bench.py --x2 --m 123930 --iters 15 > $O/gemm_$v.log 2>&1
done
timeout 300 python tools/gemm_bench.py --x2 --check --m 61965 --iters 5 > $O/gemm_check.log 2>&1
for v in default old default old; do
  L=$V/libd3dp_$v.so; [ $v = default ] && L=$PWD/d3dp_amd/lib/libd3dp_hip.so
  D3DP_LIB=$L timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-leg --no-parity 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['kernel_ms_per_step']
print('$v', round(d['value'], 2), round(d['ms_per_step'], 1), {n: round(v) for n, v in k.items() if v > 50})" >> $O/bench.log
done
tail -n 14 $O/*.log | cut -c1-400
