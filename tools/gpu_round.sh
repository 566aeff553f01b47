#!/bin/bash
# One parametrised driver for everything this repo runs on the GPU box (replaces the per-call scratch scripts of rounds 2-3):
#   tools/gpu_round.sh STAGE [STAGE ...]      results under gpurun_out/$TAG/ (TAG defaults to "r06")
# stages
#   guard        the new kernels once, small, under a short timeout (a hang here must not take the rest of the call with it);
#                on failure the remaining stages run with D3DP_X2_SKEW=0
#   tests        pytest -m gpu (PYTEST_ARGS to narrow)
#   parity       pytest -s of the fixture / full-size / training / width tests: the per-test distances, kept as profiles/<RN>_parity.log
#   variants     the tests marked `variants` against lib/variants/libd3dp_variants.so (the experiment kernels)
#   smoke        __graft_entry__.smoke()
#   ab           short bench runs, one per entry of AB (";"-separated "name:ENV=V ENV=V ..." entries; LIB=path selects a variant build)
#   bench        the driver-style bench line (STEPS / WARMUP)
#   stats        rocprofv3 --kernel-trace --stats of one bench step -> kernel_stats.md
#   pmc_gemm     FETCH_SIZE / WRITE_SIZE / MFMA-busy of the qkv Linear inside the denoiser at the bench's own pass sizes
#   pmc_step     the same counters over every kernel class of one step (one filtered pass per class and counter group)
#   gemm         tools/gemm_bench.py micro-benchmark of the four Linear shapes (GEMM_ARGS)
#   train        tools/train_bench.py
#   c5stats      rocprofv3 --kernel-trace --stats of bench.py's OWN configs[4] step (bench.py --train-only), two streams and one
#   c5pmc        FETCH_SIZE / WRITE_SIZE / MFMA-busy over every kernel of that step (3 steps per pass: 1 warm-up + 2 timed)
set -u
TAG=${TAG:-r06}; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
QUICK="--no-cpu-baseline --no-other-leg --no-parity --no-configs"
json_line() { python -c "
import sys, json
l = [x for x in sys.stdin.read().splitlines() if x.startswith('{')]
d = json.loads(l[-1]); k = d.get('kernel_ms_per_step', {})
print('$1', round(d['value'], 2), round(d['ms_per_step'], 1), {n: round(v) for n, v in k.items() if v > 40})"; }
for stage in "$@"; do
  case $stage in
    guard)
      # the kernels that changed this round, small, under a short timeout: a hang here must not take the rest of the call with it
      if ! timeout ${GUARD_TIMEOUT:-420} python -m pytest tests/test_hip_parity.py -q -x -s -k "${GUARD_K:-training_step_matches_reference or bit_reproducible or train_branch_forward}" > $O/guard.log 2>&1; then
        echo "guard FAILED" | tee -a $O/guard.log; tail -40 $O/guard.log; exit 1
      fi; tail -5 $O/guard.log ;;
    parity)
      # the parity printout of the final library, kept under profiles/ 
      timeout 1200 python -m pytest tests -m gpu -q -s -k "g3 or g4 or c2_full or c3_full or scale or config5 or longer_than_256 or bit_reproducible or outside_the_instantiated_set or more_than_32_joints or went_through_training" > $O/parity.log 2>&1; grep -E "mm|passed|failed|error|worst" $O/parity.log | tail -40 ;;
    variants)
      # the tests marked `variants` against the library that carries the experiment kernels
      D3DP_LIB=$R/d3dp_amd/lib/variants/libd3dp_variants.so timeout 900 python -m pytest tests -m "gpu and variants" -q -rs > $O/variants.log 2>&1; tail -6 $O/variants.log ;;
    trainstats)
      ( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $O/tstats -- python $R/tools/train_bench.py ${TRAIN_ARGS:-5} > $O/train_prof.log 2>&1 )
      db=$(find $O/tstats -name "*.db" | head -1); [ -n "$db" ] && python tools/rocprof_summary.py $db $O/train_kernel_stats.md > /dev/null; rm -rf $O/tstats; head -45 $O/train_kernel_stats.md; tail -2 $O/train_prof.log ;;
    tests)
      timeout 1500 python -m pytest tests -m gpu -q --durations=8 -rs ${PYTEST_ARGS:-} > $O/tests_full.log 2>&1; tail -30 $O/tests_full.log > $O/tests.log; tail -12 $O/tests.log ;;
    smoke)
      timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -3 $O/smoke.log ;;
    ab)
      IFS=';' read -ra ENTRIES <<< "${AB:-base:}"
      for e in "${ENTRIES[@]}"; do
        name=${e%%:*}; envs=${e#*:}
        ( for kv in $envs; do case $kv in LIB=*) export D3DP_LIB=$R/d3dp_amd/lib/variants/libd3dp_${kv#LIB=}.so;; *) export $kv;; esac; done
          timeout 400 python bench.py --steps ${AB_STEPS:-2} --warmup 1 $QUICK ${AB_ARGS:-} 2>$O/ab_$name.err | json_line $name ) >> $O/ab.log
      done; cat $O/ab.log ;;
    bench)
      timeout 900 python bench.py --steps ${STEPS:-20} --warmup ${WARMUP:-5} ${BENCH_ARGS:-} > $O/bench.json 2> $O/bench.err; head -c 900 $O/bench.json; echo ;;
    stats)
      ( cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats -d $O/stats -- python $R/bench.py --steps 1 --warmup 1 $QUICK > $O/bench_prof.json 2>/dev/null )
      db=$(find $O/stats -name "*.db" | head -1); [ -n "$db" ] && python tools/rocprof_summary.py $db $O/kernel_stats.md > /dev/null; rm -rf $O/stats; head -30 $O/kernel_stats.md ;;
    pmc_gemm)
      for mode in ${MODES:-exact}; do
        rx="f16x2_kernelILi0ELi1E"; [ $mode = fast ] && rx="Li8ELi4ELi1E"
        B="$R/bench.py --steps 1 --warmup 0 --no-profile --numerics $mode $QUICK"
        for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
          t=$(echo $c | cut -d' ' -f1)
          ( cd /tmp && timeout 300 rocprofv3 --pmc $c --kernel-include-regex "$rx" --output-format csv -d $O/pmc_${mode}_$t -- python $B > /dev/null 2>&1 )
        done
      done
      for d in $O/pmc_*_*/; do f=$(find $d -name "*counter_collection.csv" | head -1); echo "== $d"; [ -n "$f" ] && python tools/pmc_summary.py $f gemm | grep -v "^$"; done > $O/pmc.log 2>&1
      rm -rf $O/pmc_*_*/; grep -E "==|FETCH|WRITE|MFMA|GRBM" $O/pmc.log ;;          # (raw counter dumps: gpurun brings back at most 64 MiB)
    pmc_step)
      # One filtered pass per kernel class and counter group: rocprofv3 7.2 segfaults on this run (batch 16: 60 k dispatches)
      # unfiltered, with an alternation in the regex, or with a regex that matches all three Linear instantiations at once.
      for mode in ${MODES:-exact}; do
        B="$R/bench.py --steps 1 --warmup 0 --no-profile --numerics $mode $QUICK"
        csvs=""
        i=0
        RX=("f16x2_kernelILi0ELi1E" "f16x2_kernelILi2ELi0E" "f16x2_kernelILi1ELi0E" "attn_temporal_x2" "attn_spatial_x2" "ln2_kernel" "ln_kernel<" "embed_ln" "head_kernel")
        # (FAST mode's kernels carry other names: the four instantiations of the streaming bf16 Linear -- qkv, fc1 + GELU, fc2, proj --, the bf16 attentions)
        [ $mode = fast ] && RX=("stream_kernelILi0EDF16bLi8ELi4ELi1E" "stream_kernelILi1EDF16bLi8ELi4ELi0E" "stream_kernelILi0EDF16bLi16ELi4ELi0E" "stream_kernelILi0EDF16bLi8ELi4ELi0E" "attn_temporal2_bf16" "attn_spatial_bf16" "ln2_kernel" "ln_kernelILi512" "embed_ln" "head_kernel")
        for rx in "${RX[@]}"; do
          i=$((i + 1))
          for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
            t=$(echo $c | cut -d' ' -f1); d=$O/step_${mode}_${i}_$t
            ( cd /tmp && timeout 200 rocprofv3 --pmc $c --kernel-include-regex "$rx" --output-format csv -d $d -- python $B > $d.log 2>&1 )
            f=$(find $d -name "*counter_collection.csv" 2>/dev/null | head -1); [ -n "$f" ] && csvs="$csvs $f" || echo "no counters: $rx $t (see $d.log)"
          done
        done
        python tools/pmc_step_summary.py $O/step_pmc_$mode.md $csvs > /dev/null
        head -14 $O/step_pmc_$mode.md
        for d in $O/step_${mode}_*/; do rm -rf $d; done           # (raw counter dumps: gpurun brings back at most 64 MiB; the .log files stay)
      done ;;
    gemm)
      timeout 300 python tools/gemm_bench.py --x2 ${GEMM_ARGS:---m 128960 --iters 10} > $O/gemm_bench.log 2>&1; cat $O/gemm_bench.log ;;
    train)
      timeout 300 python tools/train_bench.py ${TRAIN_ARGS:-10} > $O/train_bench.log 2>&1; tail -3 $O/train_bench.log ;;
    c5stats)
      for mode in two one; do
        ( cd /tmp && [ $mode = one ] && export D3DP_TRAIN_OVERLAP=0; timeout 400 rocprofv3 --kernel-trace --stats -d $O/c5st_$mode -- python $R/bench.py --train-only --steps 10 --warmup 2 --no-profile > $O/c5_prof_$mode.json 2>/dev/null )
        db=$(find $O/c5st_$mode -name "*.db" | head -1); [ -n "$db" ] && python tools/rocprof_summary.py $db $O/c5_kernel_stats_$mode.md > /dev/null; rm -rf $O/c5st_$mode
      done; head -24 $O/c5_kernel_stats_one.md | cut -c1-150 ;;
    c5pmc)
      csvs=""
      for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
        t=$(echo $c | cut -d' ' -f1); d=$O/c5pmc_$t
        ( cd /tmp && D3DP_TRAIN_OVERLAP=0 timeout 300 rocprofv3 --pmc $c --output-format csv -d $d -- python $R/bench.py --train-only --steps 2 --warmup 1 --no-profile > $d.log 2>&1 )
        f=$(find $d -name "*counter_collection.csv" 2>/dev/null | head -1); [ -n "$f" ] && csvs="$csvs $f" || echo "no counters: $t (see $d.log)"
      done
      python tools/pmc_step_summary.py $O/c5_pmc.md $csvs > /dev/null; head -20 $O/c5_pmc.md | cut -c1-150; for d in $O/c5pmc_*/; do rm -rf $d; done ;;
    *) echo "unknown stage $stage" ;;
  esac
done
sha256sum d3dp_amd/lib/libd3dp_hip.so > $O/lib.sha256
