#!/bin/bash
# PMC passes over the EXACT-mode GEMM variants (qkv shape): where do the wave cycles go?
mkdir -p gpurun_out/pmc3
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in "--x2" "--x2 --x2-tile" "--x3"; do
  tag=$(echo $v | tr -d ' -')
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --output-format csv -d $R/gpurun_out/pmc3/$tag -- python $R/tools/gemm_bench.py $v --shapes qkv --iters 4 > $R/gpurun_out/pmc3/$tag.log 2>&1
  rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc3/${tag}_b -- python $R/tools/gemm_bench.py $v --shapes qkv --iters 4 > $R/gpurun_out/pmc3/${tag}_b.log 2>&1
done
cd $R
for d in gpurun_out/pmc3/*/; do f=$(find $d -name "*counter_collection.csv" | head -1); echo "== $d"; [ -n "$f" ] && python tools/pmc_summary.py $f gemm | grep -v "^$" | head -14; done
