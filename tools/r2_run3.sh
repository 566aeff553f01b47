#!/bin/bash
# PMC passes over the EXACT-mode GEMM (qkv shape) and the FAST GEMM: where do the wave cycles go?
mkdir -p gpurun_out/pmc5
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in "--x2" "--fast"; do
  tag=$(echo $v | tr -d ' -'); arg=$v; [ "$v" = "--fast" ] && arg=""
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --output-format csv -d $R/gpurun_out/pmc5/$tag -- python $R/tools/gemm_bench.py $arg --shapes qkv --iters 4 > $R/gpurun_out/pmc5/$tag.log 2>&1
  rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/pmc5/${tag}_b -- python $R/tools/gemm_bench.py $arg --shapes qkv --iters 4 > $R/gpurun_out/pmc5/${tag}_b.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc5/${tag}_f -- python $R/tools/gemm_bench.py $arg --shapes qkv --iters 4 > $R/gpurun_out/pmc5/${tag}_f.log 2>&1
  rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/pmc5/${tag}_w -- python $R/tools/gemm_bench.py $arg --shapes qkv --iters 4 > $R/gpurun_out/pmc5/${tag}_w.log 2>&1
done
cd $R
for d in gpurun_out/pmc5/*/; do f=$(find $d -name "*counter_collection.csv" | head -1); echo "== $d"; [ -n "$f" ] && python tools/pmc_summary.py $f gemm | grep -v "^$" | head -14; done
