#!/bin/bash
# chunk sweep (exact), FETCH_SIZE calibration, rocprofv3 kernel stats of the exact-mode bench
mkdir -p gpurun_out/r5
B="--steps 1 --warmup 1 --no-other-leg --no-cpu-baseline --no-parity"
{ for c in 20 30; do echo "== exact chunk $c"; bash tools/ab_bench.sh default default $B --chunk-seqs $c | head -1; done; } > gpurun_out/r5/chunks.log 2>&1
R=$PWD
cd /tmp && export TMPDIR=/tmp
for v in "--x2" ""; do tag=x2; [ -z "$v" ] && tag=fast
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/r5/cal_${tag}_f -- python $R/tools/gemm_bench.py $v --shapes cal,qkv --iters 4 > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/r5/cal_${tag}_w -- python $R/tools/gemm_bench.py $v --shapes cal,qkv --iters 4 > /dev/null 2>&1
done
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r5/stats -- python $R/bench.py --steps 1 --warmup 1 --no-other-leg --no-cpu-baseline > $R/gpurun_out/r5/bench_prof.log 2>&1
cd $R
for d in gpurun_out/r5/cal_*/; do f=$(find $d -name "*counter_collection.csv" | head -1); echo "== $d"; [ -n "$f" ] && python tools/pmc_summary.py $f gemm | grep -v "^$"; done > gpurun_out/r5/cal.log 2>&1
db=$(find gpurun_out/r5/stats -name "*.db" | head -1); [ -n "$db" ] && python tools/rocprof_summary.py $db gpurun_out/r5/kernel_stats.md > /dev/null
cat gpurun_out/r5/chunks.log gpurun_out/r5/cal.log; head -20 gpurun_out/r5/kernel_stats.md; tail -c 600 gpurun_out/r5/bench_prof.log
