#!/bin/bash
mkdir -p gpurun_out
python tools/gemm_bench.py --check --iters 10 2>&1 | grep -v amdgpu.ids > gpurun_out/r2_gemm2.log
python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "linear_all or profile_counters or full_size_prop" 2>&1 | tail -3 >> gpurun_out/r2_gemm2.log
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity > gpurun_out/r2_bench2.log 2> gpurun_out/r2_bench2.err
cat gpurun_out/r2_gemm2.log; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2_bench2.log').read().strip().splitlines()[-1])
print('exact value',round(d['value'],2),'ms',round(d['ms_per_step'],1), 'roof', round(d['roofline']['achieved'],1), d['roofline']['kernel'])
print(d['kernel_ms_per_step'])
f=d['fast_mode']; print('fast value', round(f['value'],2), f['roofline']['kernel'], round(f['roofline']['achieved'],1)); print(f['kernel_ms_per_step'])
PY
