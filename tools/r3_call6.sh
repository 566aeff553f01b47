#!/bin/bash
# round 3, GPU call 6: timing probes of the h2i EXACT Linear (parts compiled out one at a time; results invalid)
O=gpurun_out/r3c6; mkdir -p $O
export TMPDIR=/tmp
V=$PWD/d3dp_amd/lib/variants
for v in default p1 p4 p5 p32 p37 default; do
  L=$V/libd3dp_$v.so; [ $v = default ] && L=$PWD/d3dp_amd/lib/libd3dp_hip.so
  echo "== $v" >> $O/probes.log
  D3DP_LIB=$L timeout 300 python tools/gemm_bench.py --x2 --m 123930 --iters 15 2>&1 | grep "^x2" >> $O/probes.log
done
cat $O/probes.log
