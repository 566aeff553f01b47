#!/bin/bash
# Same-box interleaved A/B of the configs[4] training step with the library's per-class times beside the step:
#   tools/train_ab_classes.sh "name:ENV=V ENV=V;name2:..." [rounds] [classes, comma-separated]
# (the product library unless an entry carries LIB=path)
IFS=';' read -ra ENTRIES <<< "${1:-base:}"
CLS=${3:-train_ln_fwd,train_ln_bwd,train_operand_pass,train_wgrad,train_linear}
for round in $(seq 1 ${2:-2}); do
  for e in "${ENTRIES[@]}"; do
    name=${e%%:*}; envs=${e#*:}
    ( for kv in $envs; do case $kv in LIB=*) export D3DP_LIB=${kv#LIB=};; *) export $kv;; esac; done
      python bench.py --train-only --steps 20 --warmup 3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())['c5_train_step']
r = d.get('roofline_by_kernel', {})
print('$name', round(d['ms_per_step'], 3), 'ms', d.get('clock_mhz_mean'), 'MHz |', ' '.join(f\"{k[6:]} {r[k]['ms_per_step']:.3f}\" for k in '$CLS'.split(',') if k in r))" )
  done
done
