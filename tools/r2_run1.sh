#!/bin/bash
# GPU-box run 1 of round 2: parity tests, GEMM micro-benchmarks (fp16x2 vs bf16x3 vs bf16), subnormal probe, bench line.
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -x --durations=15 2>&1 | tail -60 > gpurun_out/r2_tests1.log
python -m pytest tests/test_hip_parity.py -m gpu -q -s -k "split_f16 or split_bf16 or c2_full or c3_full or bf16_emulating or g3_full or g4_sampler or scale_and" 2>&1 | grep -E "mean \|err\||MPJPE|exact|fast|passed|failed|checksum" > gpurun_out/r2_numbers1.log
python tools/probe_f16_denorm.py > gpurun_out/r2_denorm.log 2>&1
( python tools/gemm_bench.py --x2 --check; python tools/gemm_bench.py --x3; python tools/gemm_bench.py ) > gpurun_out/r2_gemm1.log 2>&1
python bench.py --steps 2 --warmup 1 > gpurun_out/r2_bench1.log 2> gpurun_out/r2_bench1.err
tail -3 gpurun_out/r2_tests1.log; cat gpurun_out/r2_gemm1.log; cat gpurun_out/r2_denorm.log; tail -c 1500 gpurun_out/r2_bench1.log
