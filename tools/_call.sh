O=gpurun_out/c9; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -q -x -k "train or g6 or resume or epoch or 3dhp or cli" -s 2>&1 | grep -E "config-5|passed|failed|Error|error" | tail -12 | tee $O/train_tests.log
timeout 200 python tools/train_bench.py 10 2>&1 | tail -2 | tee $O/train_bench.log
D3DP_TRAIN_IMPL=f32 timeout 200 python tools/train_bench.py 10 2>&1 | tail -1 | tee -a $O/train_bench.log
