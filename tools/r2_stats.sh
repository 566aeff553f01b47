#!/bin/bash
# rocprofv3 kernel stats of one bench step (both legs) for the current library
mkdir -p gpurun_out/final; R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/final/stats -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-parity > $R/gpurun_out/final/bench_prof.json 2> /dev/null
cd $R
db=$(find gpurun_out/final/stats -name "*.db" | head -1); [ -n "$db" ] && python tools/rocprof_summary.py $db gpurun_out/final/kernel_stats.md > /dev/null
rm -rf gpurun_out/final/stats; head -8 gpurun_out/final/kernel_stats.md | cut -c1-160
