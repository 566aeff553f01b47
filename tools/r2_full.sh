#!/bin/bash
# full GPU check of the round-2 build: every -m gpu test, smoke(), the bench line (exact headline + fast leg)
mkdir -p gpurun_out
python -m pytest tests -m gpu -q --durations=8 2>&1 | tail -25 > gpurun_out/r2_tests_full.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r2_smoke.log 2>&1
python bench.py --steps 3 --warmup 1 > gpurun_out/r2_bench_full.log 2> gpurun_out/r2_bench_full.err
tail -12 gpurun_out/r2_tests_full.log; cat gpurun_out/r2_smoke.log | tail -3; head -c 1500 gpurun_out/r2_bench_full.log
