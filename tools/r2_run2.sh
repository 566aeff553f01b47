#!/bin/bash
mkdir -p gpurun_out
python tools/gemm_bench.py --x2 --check --iters 10 2>&1 | grep -v amdgpu.ids > gpurun_out/r2_gemm2.log
python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "split_f16" 2>&1 | tail -2 >> gpurun_out/r2_gemm2.log
python bench.py --steps 1 --warmup 1 --no-other-leg --no-cpu-baseline --no-parity > gpurun_out/r2_bench2.log 2> gpurun_out/r2_bench2.err
cat gpurun_out/r2_gemm2.log; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2_bench2.log').read().strip().splitlines()[-1])
print('value',round(d['value'],2),'ms',round(d['ms_per_step'],1), 'roof', round(d['roofline']['achieved'],1), d['roofline']['kernel'])
print(d['kernel_ms_per_step'])
PY
