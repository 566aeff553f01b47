O=$PWD/gpurun_out/c11; mkdir -p $O; R=$PWD
timeout 600 python -m pytest tests -m gpu -q -x -k "train or g6 or range_guard or g3_full" -s 2>&1 | grep -E "config-5|passed|failed|Error|error|worst" | tail -8 | tee $O/train_tests.log
timeout 200 python tools/train_bench.py 10 2>&1 | tail -1 | tee $O/train_bench.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats -- python $R/tools/train_bench.py 5 > $O/tb.log 2>&1
cd $R
db=$(find $O/stats -name "*.db" | head -1); [ -n "$db" ] && python tools/rocprof_summary.py $db $O/train_kernel_stats.md > /dev/null; rm -rf $O/stats
head -14 $O/train_kernel_stats.md
