#!/bin/bash
# hardware counters for the probe builds of the split-fp16 Linear (qkv shape, micro-benchmark)
R=$PWD; mkdir -p gpurun_out/probe
cd /tmp && export TMPDIR=/tmp
for v in p0 p1 p4 p5 p7 p8; do
  D3DP_LIB=$R/d3dp_amd/lib/variants/libd3dp_$v.so rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES \
    --kernel-include-regex f16x2_kernel --output-format csv -d $R/gpurun_out/probe/pmc_$v -- python $R/tools/gemm_bench.py --x2 --shapes qkv --m 61965 --iters 6 > /dev/null 2>&1
done
cd $R
for v in p0 p1 p4 p5 p7 p8; do f=$(find gpurun_out/probe/pmc_$v -name "*counter_collection.csv" | head -1); echo "== $v"; [ -n "$f" ] && python tools/pmc_summary.py $f f16x2 | grep -v "^$"; done 2>&1 | tee gpurun_out/probe/pmc.log
f=$(find gpurun_out/probe/pmc_p0 -name "*counter_collection.csv" | head -1); head -2 $f
