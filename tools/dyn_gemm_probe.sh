#!/bin/bash
# gemm_f16x2_dyn_kernel at shapes that separate its costs: rounds of tiles (N / 128 x M / 256 tiles on 256 CUs) x k-steps (K / 32).
# usage: tools/dyn_gemm_probe.sh TAG     -> gpurun_out/TAG/dyn_probe.md
set -u
TAG=${1:-dynp}; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
echo "| M | N | K | rounds x k-steps | gemm_f16x2_dyn_kernel avg us |" > $O/dyn_probe.md
echo "|---:|---:|---:|---|---:|" >> $O/dyn_probe.md
for s in "16384 512 512" "16384 1024 512" "16384 1536 512" "16384 2048 512" "16384 512 1024" "16384 512 2048" "16384 1024 1024" "16524 512 512" "16524 1536 512"; do
  set -- $s
  ( cd /tmp && timeout 120 rocprofv3 --kernel-trace --stats -d $O/p -- python $R/tools/dyn_gemm_probe.py $1 $2 $3 > /dev/null 2>&1 )
  db=$(find $O/p -name "*.db" | head -1)
  us=$(python $R/tools/rocprof_summary.py $db /tmp/dp.md 2>/dev/null | grep gemm_f16x2_dyn | head -1 | awk -F'|' '{print $5}')
  rm -rf $O/p
  echo "| $1 | $2 | $3 | $(( ($2 / 128 * (($1 + 255) / 256) + 255) / 256 )) x $(( $3 / 32 )) | $us |" >> $O/dyn_probe.md
done
cat $O/dyn_probe.md
