#!/usr/bin/env python3
"""Turn gpurun_out/<TAG>/ (written by tools/gpu_round.sh on the GPU box) into the committed artefacts under profiles/:
  <RN>_bench_c3.json, <RN>_bench_c3_kernel_stats.md, <RN>_gemm_traffic.json (qkv Linear: HBM bytes per launch, keyed by the
  library's sha256 and the rows per launch), <RN>_step_pmc_<mode>.md + <RN>_step_pmc.json (whole step: HBM bytes, matrix-pipe
  busy fraction), <RN>_gpu_tests.log.      usage: write_profiles.py TAG [RN]"""
import json
import os
import re
import shutil
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = sys.argv[1]
RN = sys.argv[2] if len(sys.argv) > 2 else "r06"
F = os.path.join(R, "gpurun_out", TAG)
P = os.path.join(R, "profiles")
F_, J_, C_ = 243, 17, 512


def last_json(path):
    return json.loads([l for l in open(path).read().splitlines() if l.startswith("{")][-1])


sha = open(os.path.join(F, "lib.sha256")).read().split()[0]
d = last_json(os.path.join(F, "bench.json"))
json.dump(d, open(os.path.join(P, RN + "_bench_c3.json"), "w"), indent=1)
B, Hh, K = d["config"]["global_batch"], d["config"]["hypotheses_total"], 10
for name in ("tests.log", "smoke.log"):
    if os.path.exists(os.path.join(F, name)):
        with open(os.path.join(P, RN + "_gpu_tests.log"), "a" if name == "smoke.log" else "w") as o:
            o.write(open(os.path.join(F, name)).read())

# ---- qkv Linear counters: "== dir" lines followed by "| NAME | mean | dispatches |"
ctr, cur = {}, None
if os.path.exists(os.path.join(F, "pmc.log")):
    for l in open(os.path.join(F, "pmc.log")):
        m = re.match(r"== .*pmc_(exact|fast)_(\w+)/", l)
        if m:
            cur = m.group(1)
            continue
        m = re.match(r"\| (\w+) \| (\d+) \| (\d+) \|", l)
        if m and cur:
            ctr.setdefault(cur, {})[m.group(1)] = (float(m.group(2)), int(m.group(3)))


def entry(mode, elem):
    c = ctr[mode]
    n = c["FETCH_SIZE"][1]                                  # qkv launches in one bench step
    rows = 2 * B * Hh * F_ * J_ * K * 16 / n                # token rows per launch (2: flip TTA, 16 blocks)
    fetch, write = c["FETCH_SIZE"][0] * 1024 * 2, c["WRITE_SIZE"][0] * 1024
    alg_in = rows * C_ * elem + 3 * C_ * C_ * elem
    alg_out = rows * 3 * C_ * elem
    busy, cyc = c["SQ_VALU_MFMA_BUSY_CYCLES"][0] / 1024, c["GRBM_GUI_ACTIVE"][0] / 8
    return {"lib_sha256": sha, "fetch_size_kib_raw": c["FETCH_SIZE"][0], "fetch_bytes_corrected_x2": fetch,
            "write_bytes": write, "hbm_bytes_per_launch": fetch + write, "algorithmic_bytes_per_launch": int(alg_in + alg_out),
            "algorithmic_read_bytes": int(alg_in), "read_amplification": round(fetch / alg_in, 3),
            "mfma_busy_cycles_per_simd": busy, "kernel_cycles": cyc, "mfma_util_hw": busy / cyc,
            "launches_profiled": n, "mean_rows_per_launch": rows, "batch": B}


if ctr:
    t = {"source": "rocprofv3 --pmc <counter> --kernel-include-regex <qkv Linear symbol> -- python bench.py --steps 1 --warmup 0 "
                   "--no-profile --numerics <mode> (the bench's own batch: 16 clips); tools/gpu_round.sh pmc_gemm, one pass per "
                   "counter set; FETCH_SIZE doubled per MI355X_MICROARCH.md (calibrated in profiles/r02_gemm_pmc.md)"}
    for mode, elem in (("exact", 4), ("fast", 2)):
        if mode in ctr and "FETCH_SIZE" in ctr[mode]:
            t[mode] = {"gemm_qkv": entry(mode, elem)}
    json.dump(t, open(os.path.join(P, RN + "_gemm_traffic.json"), "w"), indent=1)
    print(json.dumps(t.get("exact", {}), indent=1))

# ---- whole-step counters
steps = {}
for mode in ("exact", "fast"):
    src = os.path.join(F, f"step_pmc_{mode}.md")
    if not os.path.exists(src):
        continue
    shutil.copy(src, os.path.join(P, f"{RN}_step_pmc_{mode}.md"))
    rows = [l for l in open(src) if l.startswith("| `")]
    head = [h.strip() for h in open(src).readline().strip().strip("|").split("|")]
    tot = {h: 0.0 for h in head[2:]}
    for l in rows:
        cells = [x.strip() for x in l.strip().strip("|").split("|")]
        for h, v in zip(head[2:], cells[2:]):
            tot[h] += float(v)
    hbm = tot.get("FETCH_SIZE", 0) * 1024 * 2 + tot.get("WRITE_SIZE", 0) * 1024
    busy = (tot["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024) / (tot["GRBM_GUI_ACTIVE"] / 8) if tot.get("GRBM_GUI_ACTIVE") else None
    tokens_blocks = 2 * B * Hh * F_ * J_ * K * 16
    steps[mode] = {"lib_sha256": sha, "batch": B, "hbm_tb_per_step": hbm / 1e12, "mfma_busy_hw": busy,
                   "hbm_kb_per_token_block": hbm / tokens_blocks / 1e3,
                   "source": f"profiles/{RN}_step_pmc_{mode}.md: sum over every kernel of one bench step (tools/gpu_round.sh pmc_step)"}
if steps:
    json.dump(steps, open(os.path.join(P, RN + "_step_pmc.json"), "w"), indent=1)
    print(json.dumps(steps, indent=1))

if os.path.exists(os.path.join(F, "kernel_stats.md")):
    prof = last_json(os.path.join(F, "bench_prof.json")) if os.path.exists(os.path.join(F, "bench_prof.json")) else {}
    hdr = ["# rocprofv3 --kernel-trace --stats -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-other-leg --no-parity "
           "--no-configs   (tools/gpu_round.sh stats)",
           "# warm-up + timed step + the profiled step of BASELINE configs[2] in EXACT numerics (the headline).",
           "# gemm_f16x2_kernel<0,1> = the EXACT qkv Linear (own symbol); <2,0> = proj and fc2 (x += ... in place), <1,0> = fc1 + GELU.",
           f"# bench.py's roofline.avg_launch_ms (HIP events on the launch stream) in the un-profiled run of the same build: "
           f"{d['roofline']['avg_launch_ms'] * 1e3:.1f} us"
           + (f", in this profiled run: {prof['roofline']['avg_launch_ms'] * 1e3:.1f} us." if prof.get("roofline") else "."), ""]
    open(os.path.join(P, RN + "_bench_c3_kernel_stats.md"), "w").write("\n".join(hdr) + open(os.path.join(F, "kernel_stats.md")).read())
# ---- round 5: the training step's kernel table, the parity printout, the variants' test log
tk = os.path.join(F, "train_kernel_stats.md")
if os.path.exists(tk):
    tl = [l.strip() for l in open(os.path.join(F, "train_prof.log")) if l.startswith("train step")] if os.path.exists(os.path.join(F, "train_prof.log")) else []
    c5 = d.get("configs", {}).get("c5_train_step", {})
    hdr = ["# rocprofv3 --kernel-trace --stats -- python tools/train_bench.py 5   (7 steps of BASELINE configs[4]: B=4, F=243, cs=512, dep=8, incl. AdamW;",
           "# tools/gpu_round.sh trainstats).  Per-kernel durations are measured with the weight-gradient products running on a second",
           "# stream beside the dgrad products (d3dp_ctx::aux): the two GEMM kernels' and sum_partials' times below are CONCURRENT",
           "# durations and do not add up to the step (D3DP_TRAIN_OVERLAP=0 serialises them).",
           f"# step time in the profiled run: {tl[-1] if tl else 'n/a'}; in bench.py's configs block (no optimizer step): "
           f"{c5.get('ms_per_step', float('nan')):.2f} ms.", ""]
    open(os.path.join(P, RN + "_train_step_kernel_stats.md"), "w").write("\n".join(hdr) + open(tk).read())
# ---- round 6: the configs[4] step as bench.py itself runs it: rocprofv3 --stats on two streams and on one, and
# the whole step's counters
c5 = d.get("configs", {}).get("c5_train_step", {})
for mode, what in (("two", "the weight-gradient products on the library's second stream, as un-profiled: the two GEMM kernels' and ln_bwd2<true>'s durations are CONCURRENT and contain waits for compute units"),
                   ("one", "D3DP_TRAIN_OVERLAP=0: everything on the caller's stream, so that no duration contains a wait -- the durations add up to the step")):
    src = os.path.join(F, f"c5_kernel_stats_{mode}.md")
    if os.path.exists(src):
        pj = os.path.join(F, f"c5_prof_{mode}.json")
        ms = json.load(open(pj))["c5_train_step"]["ms_per_step"] if os.path.exists(pj) and os.path.getsize(pj) else float("nan")
        hdr = [f"# rocprofv3 --kernel-trace --stats -- python bench.py --train-only --steps 10 --warmup 2 --no-profile   (tools/gpu_round.sh c5stats; 12 steps",
               f"# of BASELINE configs[4]: q_sample + MixSTE2 forward / backward + MPJPE loss, B=4 F=243, no optimizer step -- bench.py's own c5 leg).",
               f"# {what}.", f"# step time in this profiled run: {ms:.2f} ms; in the un-profiled bench line of the same build: {c5.get('ms_per_step', float('nan')):.2f} ms.", ""]
        open(os.path.join(P, f"{RN}_c5_kernel_stats_{'one_stream' if mode == 'one' else 'two_streams'}.md"), "w").write("\n".join(hdr) + open(src).read())
src = os.path.join(F, "c5_pmc.md")
if os.path.exists(src):
    rows = [l for l in open(src) if l.startswith("| `")]
    head = [h.strip() for h in open(src).readline().strip().strip("|").split("|")]
    tot = {h: 0.0 for h in head[2:]}
    for l in rows:
        cells = [x.strip() for x in l.strip().strip("|").split("|")]
        for h, v in zip(head[2:], cells[2:]):
            tot[h] += float(v)
    nsteps = 3
    hbm = (tot.get("FETCH_SIZE", 0) * 1024 * 2 + tot.get("WRITE_SIZE", 0) * 1024) / nsteps
    busy = (tot["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024) / (tot["GRBM_GUI_ACTIVE"] / 8) if tot.get("GRBM_GUI_ACTIVE") else None
    hdr = ["# rocprofv3 --pmc <counter set> -- python bench.py --train-only --steps 2 --warmup 1 --no-profile, D3DP_TRAIN_OVERLAP=0 (tools/gpu_round.sh c5pmc):",
           f"# every kernel of {nsteps} configs[4] steps (incl. torch's own small kernels), one pass per counter set; sums over the {nsteps} steps.",
           f"# per step: HBM-side bytes (FETCH_SIZE x 2 + WRITE_SIZE, MI355X_MICROARCH.md) {hbm / 1e9:.2f} GB; hardware matrix-pipe busy (sum of",
           f"# SQ_VALU_MFMA_BUSY_CYCLES / 1024 over sum of GRBM_GUI_ACTIVE / 8) {busy if busy is None else round(busy, 4)}.", ""]
    open(os.path.join(P, RN + "_c5_pmc.md"), "w").write("\n".join(hdr) + open(src).read())
    json.dump({"lib_sha256": sha, "hbm_gb_per_step": hbm / 1e9, "mfma_busy_hw": busy, "steps_counted": nsteps,
               "source": f"profiles/{RN}_c5_pmc.md"}, open(os.path.join(P, RN + "_c5_pmc.json"), "w"), indent=1)
for src, dst in (("parity.log", "_parity.log"), ("variants.log", "_variants_tests.log"), ("tests_full.log", "_gpu_tests_full.log")):
    if os.path.exists(os.path.join(F, src)):
        shutil.copy(os.path.join(F, src), os.path.join(P, RN + dst))
print("value", d["value"], "sha", sha[:12])
