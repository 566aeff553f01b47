#!/bin/bash
mkdir -p gpurun_out
{
for l in hip probe1 probe2 probe4 probe7; do echo "== lib $l"; D3DP_LIB=$PWD/d3dp_amd/lib/libd3dp_$l.so python tools/gemm_bench.py --x2 --shapes qkv,fc2 --iters 10; done
} 2>&1 | grep -v amdgpu.ids > gpurun_out/r2_probe.log
python -m pytest tests/test_hip_parity.py -m gpu -q -s -k "c3_full or deferred or bf16_emulating" 2>&1 | grep -E "MPJPE|fast F|passed|failed|Error|assert" | tail -20 > gpurun_out/r2_tests2.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 >> gpurun_out/r2_tests2.log
cat gpurun_out/r2_probe.log gpurun_out/r2_tests2.log
