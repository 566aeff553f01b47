#!/bin/bash
mkdir -p gpurun_out/try
V=$PWD/d3dp_amd/lib/variants
A="--steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-other-leg"
( LIBS="$V/libd3dp_cur.so $V/libd3dp_pk.so" timeout 900 bash tools/ab_bench.sh $A
  echo "chunk 31:"; LIBS="$V/libd3dp_pk.so" timeout 900 bash tools/ab_bench.sh $A --chunk-seqs 31
  echo "chunk 27:"; LIBS="$V/libd3dp_pk.so" timeout 900 bash tools/ab_bench.sh $A --chunk-seqs 27 ) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/try/ab.log
