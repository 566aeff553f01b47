#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "attention or g3_full or c2_full" 2>&1 | tail -3 > gpurun_out/r2_gemm2.log
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-other-leg > gpurun_out/r2_bench2.log 2> gpurun_out/r2_bench2.err
cat gpurun_out/r2_gemm2.log; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2_bench2.log').read().strip().splitlines()[-1])
print('exact value',round(d['value'],2),'ms',round(d['ms_per_step'],1), 'roof', round(d['roofline']['achieved'],1), d['roofline']['kernel'])
print(d['kernel_ms_per_step'])
PY
