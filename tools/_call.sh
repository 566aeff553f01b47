O=gpurun_out/c5; mkdir -p $O
T="python -m pytest tests/test_hip_parity.py -q -s -k g3_full_width_denoiser"
for v in dbgA dbgB dbgC; do
  echo "== $v" >> $O/num.log
  ( export D3DP_LIB=$PWD/d3dp_amd/lib/variants/libd3dp_$v.so D3DP_X2_SKEW=0; timeout 200 $T 2>&1 | grep -E "^\[F=|^F\[F=|passed|failed|Error" | head -8 >> $O/num.log )
done
cat $O/num.log
