#!/bin/bash
# A/B of builds of gemm_x2.hip (tools/build_variant.sh): micro-benchmark with result check, then the sampler parity tests on the winner candidates
mkdir -p gpurun_out/probe
for v in $VARIANTS; do
  echo "== $v"
  D3DP_LIB=$PWD/d3dp_amd/lib/variants/libd3dp_$v.so python tools/gemm_bench.py --x2 --check --shapes qkv,proj,fc1,fc2 --m 61965 --iters 15 2>&1 | grep "x2\|check"
  D3DP_LIB=$PWD/d3dp_amd/lib/variants/libd3dp_$v.so python tools/gemm_bench.py --x2 --shapes qkv --m 123930 --iters 15 2>&1 | grep "^x2"
done 2>&1 | tee gpurun_out/probe/probe.log
