#!/bin/bash
# correctness subset, then step timing on ONE box: HEAD build vs the working tree (boxes differ by 2-3 % between calls)
mkdir -p gpurun_out/try
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "attention or qkv_linear_packed or g2 or g3 or g4 or c2_full or cross_check or linear_split_f16" 2>&1 | tail -4 | tee gpurun_out/try/tests.log
V=$PWD/d3dp_amd/lib/variants
LIBS="${LIBS:-$V/libd3dp_head.so default $V/libd3dp_head.so default}" timeout 900 bash tools/ab_bench.sh --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-other-leg 2>&1 | grep -v amdgpu.ids | tee gpurun_out/try/ab.log
