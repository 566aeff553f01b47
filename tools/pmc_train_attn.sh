#!/bin/bash
# rocprofv3 --pmc passes over the training attention kernels (train_attn.hip) inside tools/train_bench.py: where their wave cycles
# go (parked / issue-stalled / issuing), LDS bank conflicts, HBM-side bytes.  usage: tools/pmc_train_attn.sh TAG [REGEX [OUT]]
# (REGEX: kernels to count, default "tattn"; "gemm_f16x2_dyn|gemm_f16x2_tn" = the training Linears; OUT: name of the summary)
set -u
TAG=${1:-pmc_ta}; RX=${2:-tattn}; OUT=${3:-pmc_train_attn}; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
i=0
for c in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU" "FETCH_SIZE WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM"; do
  i=$((i + 1))
  ( cd /tmp && timeout 300 rocprofv3 --pmc $c --kernel-include-regex "$RX" --output-format csv -d $O/p$i -- python $R/tools/train_bench.py 2 > $O/p$i.log 2>&1 )
  f=$(find $O/p$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python $R/tools/pmc_summary.py $f "$RX" >> $O/$OUT.md
done
find $O -name "*.csv" -size +5M -delete
cat $O/$OUT.md | cut -c1-160
