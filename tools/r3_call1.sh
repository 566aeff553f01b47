#!/bin/bash
# round 3, GPU call 1: 32x32x16 vs 16x16x32 form of the EXACT Linear -- correctness, micro-benchmark, bare MFMA stream, step
O=gpurun_out/r3c1; mkdir -p $O
export TMPDIR=/tmp
V=d3dp_amd/lib/variants
( timeout 600 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "linear_split_f16 or qkv_linear_packed or (linear_all_epilogues and exact)" 2>&1 | tail -15 ) > $O/pytest32.log
( D3DP_X2_SHAPE=16 timeout 600 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "linear_split_f16_is or qkv_linear_packed" 2>&1 | tail -5 ) > $O/pytest16.log
for sh in 32 16; do
  D3DP_X2_SHAPE=$sh timeout 300 python tools/gemm_bench.py --x2 --check --m 61965 --iters 15 > $O/gemm_$sh.log 2>&1
  D3DP_X2_SHAPE=$sh timeout 300 python tools/gemm_bench.py --x2 --m 123930 --iters 15 >> $O/gemm_$sh.log 2>&1
  D3DP_X2_SHAPE=$sh D3DP_LIB=$V/libd3dp_bare.so timeout 300 python tools/gemm_bench.py --x2 --m 123930 --iters 15 > $O/bare_$sh.log 2>&1
done
D3DP_X2_SHAPE=32 D3DP_LIB=$V/libd3dp_sbend.so timeout 300 python tools/gemm_bench.py --x2 --m 123930 --iters 15 > $O/sbend_32.log 2>&1
for sh in 32 16 32; do
  D3DP_X2_SHAPE=$sh timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-leg --no-parity 2>&1 | tail -1 >> $O/bench_$sh.log
done
tail -n 30 $O/*.log | cut -c1-600
