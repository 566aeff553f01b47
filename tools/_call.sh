R=$PWD; O=$R/gpurun_out/final; mkdir -p $O
B="$R/bench.py --steps 1 --warmup 0 --no-profile --numerics exact --no-cpu-baseline --no-other-leg --no-parity --no-configs"
cd /tmp; export TMPDIR=/tmp
for k in f16x2_kernelILi2ELi0E f16x2_kernelILi1ELi0E "ln_kernel<"; do
  kk=$(echo "$k" | tr -d '<')
  for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    t=$(echo $c | cut -d' ' -f1)
    timeout 300 rocprofv3 --pmc $c --kernel-include-regex "$k" --output-format csv -d $O/cls_${kk}_$t -- python $B > $O/cls_${kk}_$t.log 2>&1
    echo "$k $t rc=$? $(find $O/cls_${kk}_$t -name '*counter_collection.csv' 2>/dev/null | wc -l) csv"
  done
done
cd $R
python tools/pmc_step_summary.py $O/step_pmc_exact_b.md $(find $O/cls_f16x2* $O/cls_ln_kernel* -name "*counter_collection.csv") > /dev/null
cat $O/step_pmc_exact_b.md
find $O -name "*.csv" -size +4M -delete; rm -f $O/cls_*.log
