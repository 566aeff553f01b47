#!/bin/bash
mkdir -p gpurun_out
( python tools/gemm_bench.py --x2 --check ) > gpurun_out/r2_gemm2.log 2>&1
python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "split_f16 or g3_full or g4_sampler or g2_tiny or c2_full or cross_check" 2>&1 | tail -15 > gpurun_out/r2_tests2.log
python bench.py --steps 1 --warmup 1 --no-other-leg --no-cpu-baseline > gpurun_out/r2_bench2.log 2> gpurun_out/r2_bench2.err
cat gpurun_out/r2_gemm2.log; tail -5 gpurun_out/r2_tests2.log; tail -c 1500 gpurun_out/r2_bench2.log
