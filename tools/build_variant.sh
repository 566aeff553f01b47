#!/bin/bash
# tools/build_variant.sh NAME FILE.hip "EXTRA FLAGS": libd3dp_hip.so with ONE translation unit rebuilt with extra flags,
# written to d3dp_amd/lib/variants/libd3dp_NAME.so (git-ignored; travels with gpurun) for A/B runs via D3DP_LIB=...
set -e
cd "$(dirname "$0")/../d3dp_amd/csrc"
NAME=$1; FILE=$2; FLAGS=$3
make -s >/dev/null
mkdir -p ../lib/variants/obj
O=../lib/variants/obj/${NAME}_${FILE%.hip}.o
EXTRA=""; case $FILE in gemm*.hip) EXTRA=-fno-slp-vectorize;; jpma.hip) EXTRA=-ffp-contract=off;; esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -fvisibility-inlines-hidden -Wall -Wno-unused-function $EXTRA $FLAGS -c $FILE -o $O
OBJS=""; for f in gemm gemm_x2 attention pointwise sampler jpma caller train train_g train_attn capi; do
  if [ "$f.hip" = "$FILE" ]; then OBJS="$OBJS $O"; else OBJS="$OBJS ../lib/obj/$f.o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=exports.map -o ../lib/variants/libd3dp_$NAME.so $OBJS
echo built d3dp_amd/lib/variants/libd3dp_$NAME.so
