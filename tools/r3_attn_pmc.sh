#!/bin/bash
# round 3: counters of the EXACT attention kernels inside the denoiser (B = 4 step), for the next round's planning
R=$PWD; O=$R/gpurun_out/r3attn; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="$R/bench.py --steps 1 --warmup 0 --batch 4 --no-other-leg --no-cpu-baseline --no-parity --no-profile --numerics exact"
timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
  --kernel-include-regex "attn_(temporal|spatial)_x2" --output-format csv -d $O/sq -- python $B > /dev/null 2>&1
timeout 200 rocprofv3 --pmc TCP_TCC_READ_REQ TCP_PENDING_STALL_CYCLES \
  --kernel-include-regex "attn_(temporal|spatial)_x2" --output-format csv -d $O/tc -- python $B > /dev/null 2>&1
cd $R
for p in sq tc; do f=$(find $O/$p -name "*counter_collection.csv" | head -1); echo "== $p"; [ -n "$f" ] && python tools/pmc_summary.py $f attn_ | grep -v "^$"; done 2>&1 | tee $O/pmc.log
find $O -name "*.csv" -size +5M -delete
