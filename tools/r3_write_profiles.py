#!/usr/bin/env python3
"""Turn gpurun_out/final/ (written by tools/r2_final.sh on the GPU box) into the committed artefacts under profiles/:
rNN_bench_c3.json, rNN_bench_c3_kernel_stats.md, rNN_gemm_traffic.json (keyed by the library's sha256), rNN_step_pmc_{exact,fast}.md;
usage: r3_write_profiles.py [r03]"""
import json
import os
import re
import shutil
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F = os.path.join(R, "gpurun_out", "final")
P = os.path.join(R, "profiles")
RN = sys.argv[1] if len(sys.argv) > 1 else "r03"


def last_json(path):
    return json.loads([l for l in open(path).read().splitlines() if l.startswith("{")][-1])


d = last_json(os.path.join(F, "bench.json"))
json.dump(d, open(os.path.join(P, RN + "_bench_c3.json"), "w"), indent=1)
for mode in ("exact", "fast"):
    src = os.path.join(R, "gpurun_out", "pmc_step", mode + ".md")
    if os.path.exists(src):
        shutil.copy(src, os.path.join(P, f"{RN}_step_pmc_{mode}.md"))
sha = open(os.path.join(F, "lib.sha256")).read().split()[0]

# counters: "== dir" lines followed by "| NAME | mean | dispatches |"
ctr, cur = {}, None
for l in open(os.path.join(F, "pmc.log")):
    m = re.match(r"== .*pmc_(exact|fast)_(\w+)/", l)
    if m:
        cur = m.group(1)
        continue
    m = re.match(r"\| (\w+) \| (\d+) \| (\d+) \|", l)
    if m and cur:
        ctr.setdefault(cur, {})[m.group(1)] = (float(m.group(2)), int(m.group(3)))


def entry(mode, elem, chunk):
    c = ctr[mode]
    n = c["FETCH_SIZE"][1]
    # B=4, H=20, flip: 160 sequences per denoiser call, 16 qkv launches per pass, 10 DDIM steps: passes = n / 160
    rows = 160 * 4131 / (n / 160.0)
    fetch, write = c["FETCH_SIZE"][0] * 1024 * 2, c["WRITE_SIZE"][0] * 1024
    alg_in = rows * 512 * elem + 1536 * 512 * elem
    alg_out = rows * 1536 * elem
    busy, cyc = c["SQ_VALU_MFMA_BUSY_CYCLES"][0] / 1024, c["GRBM_GUI_ACTIVE"][0] / 8
    return {"lib_sha256": sha, "fetch_size_kib_raw": c["FETCH_SIZE"][0], "fetch_bytes_corrected_x2": fetch,
            "write_bytes": write, "hbm_bytes_per_launch": fetch + write, "algorithmic_bytes_per_launch": int(alg_in + alg_out),
            "algorithmic_read_bytes": int(alg_in), "read_amplification": round(fetch / alg_in, 3),
            "mfma_busy_cycles_per_simd": busy, "kernel_cycles": cyc, "mfma_util_hw": busy / cyc,
            "launches_profiled": n, "mean_rows_per_launch": rows}


t = {"source": "rocprofv3 --pmc <counter> --kernel-include-regex <qkv GEMM symbol> -- python bench.py --steps 1 --warmup 0 "
               "--batch 4 --no-other-leg --no-cpu-baseline --no-parity --no-profile --numerics <mode>  (tools/r3_final.sh; one "
               "pass per counter set; FETCH_SIZE doubled per MI355X_MICROARCH.md, calibrated in profiles/r02_gemm_pmc.md)",
     "exact": {"gemm_qkv": entry("exact", 4, 31)}, "fast": {"gemm_qkv": entry("fast", 2, 15)}}
json.dump(t, open(os.path.join(P, RN + "_gemm_traffic.json"), "w"), indent=1)

prof = last_json(os.path.join(F, "bench_prof.json")) if os.path.exists(os.path.join(F, "bench_prof.json")) else {}
hdr = ["# rocprofv3 --kernel-trace --stats -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-parity   (tools/r3_final.sh)",
       "# warm-up + timed step + the profiled step of BASELINE configs[2] in EXACT numerics (the headline), then the same three steps",
       "# in FAST numerics (the secondary leg).  gemm_f16x2_kernel<0,1> = the EXACT qkv Linear (own symbol); bench.py's",
       f"# roofline.avg_launch_ms (HIP events on the launch stream) in the un-profiled run of the same build: {d['roofline']['avg_launch_ms'] * 1e3:.1f} us"
       + (f", in this profiled run: {prof['roofline']['avg_launch_ms'] * 1e3:.1f} us." if prof else "."),
       "# <2,0> = proj and fc2 (EXACT: x += ... in place), <1,0> = fc1 + GELU (EXACT); gemm_bf16_stream_kernel<...> = the FAST Linears.", ""]
open(os.path.join(P, RN + "_bench_c3_kernel_stats.md"), "w").write("\n".join(hdr) + open(os.path.join(F, "kernel_stats.md")).read())
print("value", d["value"], "fast", d["fast_mode"]["value"], "sha", sha[:12])
print(json.dumps(t["exact"]["gemm_qkv"], indent=1))
