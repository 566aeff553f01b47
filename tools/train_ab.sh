#!/bin/bash
# Same-box interleaved A/B of the configs[4] training step: tools/train_ab.sh "name:ENV=V ENV=V;name2:..." [rounds]
# (the variants library honours the superseded-form switches; LIB=path inside an entry loads another build)
R=$PWD; V=$R/d3dp_amd/lib/variants/libd3dp_variants.so
IFS=';' read -ra ENTRIES <<< "${1:-base:}"
for round in $(seq 1 ${2:-2}); do
  for e in "${ENTRIES[@]}"; do
    name=${e%%:*}; envs=${e#*:}
    ( export D3DP_LIB=$V; for kv in $envs; do case $kv in LIB=*) export D3DP_LIB=${kv#LIB=};; *) export $kv;; esac; done
      python bench.py --train-only --steps 20 --warmup 3 --no-profile 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())['c5_train_step']
print('$name', round(d['ms_per_step'], 3), 'ms', d.get('clock_mhz_mean'), 'MHz', d.get('power_w_mean'), 'W')" )
  done
done
