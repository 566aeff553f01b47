#!/bin/bash
# round 3: hardware counters of the EXACT qkv Linear, round-2 operand layout (two planes) against h2i, micro-benchmark
R=$PWD; O=$R/gpurun_out/r3pmc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 60 rocprofv3 -L 2>/dev/null | grep -oE "\b(TA_[A-Z0-9_]+|TCP_[A-Z0-9_]+|TCC_[A-Z0-9_]+)\b" | sort -u > $O/counter_names.txt
export D3DP_LIB_ANY_ABI=1
for v in r2 h2i; do
  L=$R/d3dp_amd/lib/variants/libd3dp_$v.so; [ $v = h2i ] && L=$R/d3dp_amd/lib/libd3dp_hip.so
  B="python $R/tools/gemm_bench.py --x2 --shapes qkv --m 123930 --iters 6"
  D3DP_LIB=$L timeout 150 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
    --kernel-include-regex f16x2_kernel --output-format csv -d $O/sq_$v -- $B > /dev/null 2>&1
  D3DP_LIB=$L timeout 150 rocprofv3 --pmc TA_TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TA_BUFFER_WAVEFRONTS \
    --kernel-include-regex f16x2_kernel --output-format csv -d $O/ta_$v -- $B > /dev/null 2>&1
  D3DP_LIB=$L timeout 150 rocprofv3 --pmc TCP_TCC_READ_REQ TCP_PENDING_STALL_CYCLES \
    --kernel-include-regex f16x2_kernel --output-format csv -d $O/tc_$v -- $B > /dev/null 2>&1
done
cd $R
for v in r2 h2i; do for p in sq ta tc; do f=$(find $O/${p}_$v -name "*counter_collection.csv" | head -1); echo "== $v $p"; [ -n "$f" ] && python tools/pmc_summary.py $f f16x2 | grep -v "^$\|^###"; done; done 2>&1 | tee $O/pmc.log
find $O -name "*.csv" -size +5M -delete
