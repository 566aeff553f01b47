#!/bin/bash
# round 3, GPU call 7: norm2 folded into proj / fc1 (EPI_RESID_LN / EPI_GELU_LN) -- parity tests and A/B against the row kernel
O=gpurun_out/r3c7; mkdir -p $O
export TMPDIR=/tmp
( timeout 1200 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "norm2_folded or residual_adds or range_guard or g2_tiny or g3_full or g4_sampler or c2_full or linear or cross_check or full_size_properties or scale_and_live" 2>&1 | tail -25 ) > $O/pytest.log
run() {
  env $2 timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-leg --no-parity 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['kernel_ms_per_step']
print('$1', round(d['value'], 2), round(d['ms_per_step'], 1), {n: round(v) for n, v in k.items() if v > 20})" >> $O/bench.log
}
run folded A=1; run kernel D3DP_NO_FOLD_LN=1; run folded A=1; run kernel D3DP_NO_FOLD_LN=1
tail -n 30 $O/*.log | cut -c1-600
