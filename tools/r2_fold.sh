#!/bin/bash
# one box: tests of the folded-residual build; A/B of the folded epilogue against the same library with D3DP_NO_FOLD=1;
# only if folding wins: the qkv counter passes and the default bench line for the new library
mkdir -p gpurun_out/fold gpurun_out/final; R=$PWD
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "residual_epilogue or qkv_linear_packed or g2 or g3 or g4 or c2_full or cross_check or full_size_properties" 2>&1 | tail -3 | tee gpurun_out/fold/tests.log
grep -q "failed\|error" gpurun_out/fold/tests.log && exit 1
A="--steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-other-leg"
val() { python - "$1" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
k = d["kernel_ms_per_step"]
print(round(d["value"], 3), round(d["ms_per_step"], 1), {n: round(v) for n, v in k.items() if v > 100}, file=sys.stderr)
print(d["value"])
PY
}
D3DP_NO_FOLD=1 python bench.py $A > gpurun_out/fold/nofold.json 2>/dev/null; a=$(val gpurun_out/fold/nofold.json 2>> gpurun_out/fold/ab.log)
python bench.py $A > gpurun_out/fold/fold.json 2>/dev/null; b=$(val gpurun_out/fold/fold.json 2>> gpurun_out/fold/ab.log)
cat gpurun_out/fold/ab.log
win=$(python -c "print(int($b > 1.005 * $a))")
echo "nofold $a fold $b win $win" | tee -a gpurun_out/fold/ab.log
[ "$win" = 1 ] || exit 0
cd /tmp && export TMPDIR=/tmp
for mode in exact fast; do
  rx="f16x2_kernelILi0ELi1E"; [ $mode = fast ] && rx="Li8ELi4ELi1E"
  B="$R/bench.py --steps 1 --warmup 0 --batch 4 --no-other-leg --no-cpu-baseline --no-parity --no-profile --numerics $mode"
  for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    tag=$(echo $c | cut -d' ' -f1)
    rocprofv3 --pmc $c --kernel-include-regex $rx --output-format csv -d $R/gpurun_out/final/pmc_${mode}_$tag -- python $B > /dev/null 2>&1
  done
done
cd $R
for d in gpurun_out/final/pmc_*/; do f=$(find $d -name "*counter_collection.csv" | sort | tail -1); echo "== $d"; [ -n "$f" ] && python tools/pmc_summary.py $f gemm | grep -v "^$"; done > gpurun_out/final/pmc.log 2>&1
sha256sum d3dp_amd/lib/libd3dp_hip.so > gpurun_out/final/lib.sha256
python bench.py > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err
find gpurun_out/final -name "*.csv" -size +10M -delete
head -c 400 gpurun_out/final/bench.json; echo; grep -E "==|FETCH|WRITE|MFMA|GRBM" gpurun_out/final/pmc.log
