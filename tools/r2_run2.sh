#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_hip_parity.py -m gpu -q -x -s -k "attention or g2_tiny or g3_full or c2_full" 2>&1 | grep -E "impl=2|passed|failed|rror" | tail -12 > gpurun_out/r2_gemm2.log
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-leg --no-parity > gpurun_out/r2_bench2.log 2> gpurun_out/r2_bench2.err
cat gpurun_out/r2_gemm2.log; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2_bench2.log').read().strip().splitlines()[-1])
print('exact value',round(d['value'],2),'ms',round(d['ms_per_step'],1))
print(d['kernel_ms_per_step'])
PY
