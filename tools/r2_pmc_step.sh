#!/bin/bash
# whole-step counters of the EXACT-mode bench step (and the FAST one): FETCH_SIZE, WRITE_SIZE, matrix-pipe busy, one pass each
mkdir -p gpurun_out/pmc_step
R=$PWD
cd /tmp && export TMPDIR=/tmp
for mode in ${MODES:-exact fast}; do
  B="$R/bench.py --steps 1 --warmup 0 --batch 4 --no-other-leg --no-cpu-baseline --no-parity --no-profile --numerics $mode"
  timeout 180 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_step/${mode}_f -- python $B > /dev/null 2>&1
  timeout 180 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_step/${mode}_w -- python $B > /dev/null 2>&1
  timeout 180 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmc_step/${mode}_m -- python $B > /dev/null 2>&1
done
cd $R
for mode in ${MODES:-exact fast}; do
  python tools/pmc_step_summary.py gpurun_out/pmc_step/$mode.md $(find gpurun_out/pmc_step/${mode}_f gpurun_out/pmc_step/${mode}_w gpurun_out/pmc_step/${mode}_m -name "*counter_collection.csv") > /dev/null
  echo "== $mode"; cat gpurun_out/pmc_step/$mode.md
done
find gpurun_out/pmc_step -name "*.csv" -size +20M -delete
