O=gpurun_out/c8; mkdir -p $O
timeout 300 python -m pytest tests/test_hip_parity.py -q -x -k "pingpong" 2>&1 | tail -4 | tee $O/pp_test.log
G="python tools/gemm_bench.py --x2 --real-epi --m 128960 --iters 10"
for spec in "plain" "pp" "plain" "pp"; do
  echo "== $spec" >> $O/gemm.log
  if [ $spec = pp ]; then timeout 120 $G --pp 2>&1 | grep "^x2" >> $O/gemm.log; else timeout 120 $G 2>&1 | grep "^x2" >> $O/gemm.log; fi
done
cat $O/gemm.log
TAG=c8 AB='pp1:D3DP_X2_PP=1;pp0:D3DP_X2_PP=0;pp1b:D3DP_X2_PP=1;pp0b:D3DP_X2_PP=0' bash tools/gpu_round.sh ab
