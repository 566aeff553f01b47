#!/bin/bash
# A/B: write-through (sc1) GEMM output stores vs plain stores; EXACT-mode chunk sweep
mkdir -p gpurun_out
P=$PWD/d3dp_amd/lib/libd3dp_hip_plainstores.so
{
echo "== sc1 stores"; python tools/gemm_bench.py --x2; python tools/gemm_bench.py
echo "== plain stores"; D3DP_LIB=$P python tools/gemm_bench.py --x2; D3DP_LIB=$P python tools/gemm_bench.py
} > gpurun_out/r2_gemm4.log 2>&1
B="--steps 1 --warmup 1 --no-other-leg --no-cpu-baseline --no-parity"
{
echo "== exact chunk 15: sc1 vs plain"; bash tools/ab_bench.sh default $P $B
for c in 5 8 10; do echo "== exact chunk $c (sc1)"; bash tools/ab_bench.sh default default $B --chunk-seqs $c | head -1; done
echo "== fast chunk 15: sc1 vs plain"; bash tools/ab_bench.sh default $P $B --numerics fast
} > gpurun_out/r2_ab4.log 2>&1
cat gpurun_out/r2_gemm4.log gpurun_out/r2_ab4.log | grep -v amdgpu.ids
