#!/bin/bash
mkdir -p gpurun_out/r6
R=$PWD; P=$R/d3dp_amd/lib/libd3dp_plainout.so
{ echo "== nt out"; python tools/gemm_bench.py --x2 --iters 10; echo "== plain out"; D3DP_LIB=$P python tools/gemm_bench.py --x2 --iters 10; } 2>&1 | grep -v amdgpu > gpurun_out/r6/time.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/r6/nt_f -- python $R/tools/gemm_bench.py --x2 --shapes qkv --iters 4 > /dev/null 2>&1
D3DP_LIB=$P rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/r6/plain_f -- python $R/tools/gemm_bench.py --x2 --shapes qkv --iters 4 > /dev/null 2>&1
cd $R
for d in gpurun_out/r6/*_f/; do f=$(find $d -name "*counter_collection.csv" | head -1); echo "== $d"; [ -n "$f" ] && python tools/pmc_summary.py $f gemm | grep FETCH; done > gpurun_out/r6/fetch.log 2>&1
B="--steps 1 --warmup 1 --no-other-leg --no-cpu-baseline --no-parity"
bash tools/ab_bench.sh default $P $B 2>&1 | grep -v amdgpu > gpurun_out/r6/ab.log
cat gpurun_out/r6/time.log gpurun_out/r6/fetch.log gpurun_out/r6/ab.log
