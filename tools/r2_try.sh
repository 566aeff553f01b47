#!/bin/bash
mkdir -p gpurun_out/try
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "attention or g3 or c2_full" 2>&1 | tail -4 | tee gpurun_out/try/tests.log
V=$PWD/d3dp_amd/lib/variants
A="--steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-other-leg"
( LIBS="$V/libd3dp_noprio.so default" timeout 1200 bash tools/ab_bench.sh $A ) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/try/ab.log
