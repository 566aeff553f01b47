#!/bin/bash
# A/B of the training step's switches on ONE box: tools/train_ab.sh "name:ENV=V ENV=V" ... (each entry run REPS times, interleaved)
R=${REPS:-2}; N=${STEPS:-20}
for rep in $(seq $R); do
  for e in "$@"; do
    name=${e%%:*}; envs=${e#*:}
    ( for kv in $envs; do export $kv; done; echo -n "$name  "; timeout 200 python tools/train_bench.py $N 2>&1 | grep "train step" )
  done
done
