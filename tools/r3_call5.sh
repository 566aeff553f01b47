#!/bin/bash
# round 3, GPU call 5: new tests (range guard, backward version check), planned pass sizes, residual-tile touch (PFR) A/B
O=gpurun_out/r3c5; mkdir -p $O
export TMPDIR=/tmp
V=$PWD/d3dp_amd/lib/variants
( timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "range_guard or backward_after or deferred_backward or full_size_properties or rejects_k or layernorm or g4_sampler" 2>&1 | tail -15 ) > $O/pytest.log
run() {  # name lib chunk
  L=$V/libd3dp_$2.so; [ $2 = default ] && L=$PWD/d3dp_amd/lib/libd3dp_hip.so
  D3DP_LIB=$L timeout 600 python bench.py --steps 2 --warmup 1 --chunk-seqs $3 --no-cpu-baseline --no-other-leg --no-parity 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['kernel_ms_per_step']
print('$1', round(d['value'], 2), round(d['ms_per_step'], 1), {n: round(v) for n, v in k.items() if v > 50})" >> $O/bench.log
}
run plan31 default 0; run pfr pfr 0; run uni30 default -30; run plan31 default 0; run pfr pfr 0; run plan47 default 47
( D3DP_LIB=$V/libd3dp_pfr.so timeout 600 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "residual_epilogue or linear_all_epilogues or g3_full_width" 2>&1 | tail -4 ) > $O/pytest_pfr.log
tail -n 20 $O/*.log | cut -c1-500
