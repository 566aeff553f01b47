#!/bin/bash
# final measurements of the round-3 build: tests, smoke, bench line, rocprofv3 kernel stats, in-pipeline PMC of the qkv GEMMs
mkdir -p gpurun_out/final
R=$PWD
[ -n "$SKIP_TESTS" ] || timeout 900 python -m pytest tests -m gpu -q --durations=8 -rs 2>&1 | tail -24 > gpurun_out/final/tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final/smoke.log 2>&1
timeout 600 python bench.py --steps ${STEPS:-5} --warmup ${WARMUP:-2} > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/final/stats -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-parity > $R/gpurun_out/final/bench_prof.json 2> /dev/null
for mode in exact fast; do
  rx="f16x2_kernelILi0ELi1E"; [ $mode = fast ] && rx="Li8ELi4ELi1E"
  B="$R/bench.py --steps 1 --warmup 0 --batch 4 --no-other-leg --no-cpu-baseline --no-parity --no-profile --numerics $mode"
  for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    tag=$(echo $c | cut -d' ' -f1)
    timeout 200 rocprofv3 --pmc $c --kernel-include-regex $rx --output-format csv -d $R/gpurun_out/final/pmc_${mode}_$tag -- python $B > /dev/null 2>&1
  done
done
cd $R
db=$(find gpurun_out/final/stats -name "*.db" | head -1); [ -n "$db" ] && python tools/rocprof_summary.py $db gpurun_out/final/kernel_stats.md > /dev/null
for d in gpurun_out/final/pmc_*/; do f=$(find $d -name "*counter_collection.csv" | head -1); echo "== $d"; [ -n "$f" ] && python tools/pmc_summary.py $f gemm | grep -v "^$"; done > gpurun_out/final/pmc.log 2>&1
sha256sum d3dp_amd/lib/libd3dp_hip.so > gpurun_out/final/lib.sha256
find gpurun_out/final -name "*.csv" -size +10M -delete; rm -rf gpurun_out/final/stats
tail -6 gpurun_out/final/tests.log; tail -2 gpurun_out/final/smoke.log; head -c 600 gpurun_out/final/bench.json; echo; cat gpurun_out/final/pmc.log | grep -E "==|FETCH|WRITE|MFMA|GRBM"
# whole-step counters, both modes (FETCH_SIZE / WRITE_SIZE / matrix-pipe busy, one pass each)
[ -n "$SKIP_STEP_PMC" ] || timeout 600 bash tools/r2_pmc_step.sh > gpurun_out/final/step_pmc.log 2>&1
# pass-size sweeps, both modes (cheap; after everything that must exist)
if [ -z "$SKIP_SWEEP" ]; then
  for spec in "exact 23" "exact 27" "exact 0" "fast 0" "fast 23" "fast 31" "fast 47" "fast 0"; do
    set -- $spec
    timeout 300 python bench.py --steps 2 --warmup 1 --numerics $1 --chunk-seqs $2 --no-cpu-baseline --no-other-leg --no-parity 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['kernel_ms_per_step']
print('$1 chunk $2:', round(d['value'], 2), round(d['ms_per_step'], 1), {n: round(v) for n, v in k.items() if v > 50})" >> gpurun_out/final/sweep.log
  done
  cat gpurun_out/final/sweep.log
fi
