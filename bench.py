#!/usr/bin/env python3
"""Headline benchmark of the D3DP hot path on MI355X (contract: see the round prompt / DESIGN.md §Measurement).

One "step" = one D3DP.forward (ddim_sample_flip: K DDIM steps x 2 flip-TTA denoiser passes) over one synthetic batch
of B clips x H hypotheses at F=243, J=17.  N=1 workload = BASELINE.json configs[2] (B=16, H=20, K=10), the
configuration the metric is quoted on.  With N>1 ranks every rank samples its own H=20 hypotheses of the same clips
(weak scaling, no data-path collective during sampling) and one RCCL all-gather assembles (B,K,N*20,F,17,3) on every
rank inside the timed region.  Inputs are resident in HBM before the timed region starts.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from types import SimpleNamespace

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

import torch  # noqa: E402

F_, J_, C_, DEPTH = 243, 17, 512, 8
PEAK_BF16_TFLOPS = 2500.0      # dense bf16 MFMA, MI355X_MICROARCH.md (2.5 PFLOP/s)
PEAK_F32_TFLOPS = 157.3        # fp32 MFMA / vector peak


def flops_per_denoiser_call(frames=F_, joints=J_, c=C_, depth=DEPTH):
    """Algorithmic FLOP of one MixSTE2.forward per (clip, hypothesis) -- SURVEY.md §8 D4 (multiply-add = 2)."""
    tseq = frames * joints
    per_tok = depth * 2 * 8 * c * c * 2              # 16 blocks x (3+1+2+2) C^2 MACs x 2 = 16*16*C^2
    attn = depth * 4 * joints * c + depth * 4 * frames * c
    return tseq * (per_tok + attn + 2 * 5 * c + 2 * c * 3) + 2 * 2 * c * 2 * c


def build_model(H, K, numerics, chunk_seqs):
    from d3dp_amd import D3DP
    from d3dp_amd.weights import H36M_JOINTS_LEFT, H36M_JOINTS_RIGHT, make_state_dict
    args = SimpleNamespace(number_of_frames=F_, test_time_augmentation=True, timestep=1000, scale=1.0, cs=C_,
                           dep=DEPTH, chunk_seqs=chunk_seqs)
    m = D3DP(args, H36M_JOINTS_LEFT, H36M_JOINTS_RIGHT, is_train=False, num_proposals=H, sampling_timesteps=K,
             numerics=numerics)
    m.load_state_dict(make_state_dict(7, C_, DEPTH, F_), strict=False)
    return m.cuda().eval()


def cpu_baseline(budget_s=12.0):
    """The CPU oracle (a port of the reference path; the reference's own files cannot travel to the GPU box and
    hard-code CUDA) timed on this host's cores: F=243, B=1, H=1, K=1 (= 2 denoiser calls), repeated for ~budget_s.
    Extrapolated linearly in K to the K=10 unit of the metric."""
    from d3dp_amd.weights import H36M_JOINTS_LEFT, H36M_JOINTS_RIGHT, flip_2d, make_state_dict, synthetic_inputs_2d, synthetic_noise
    from oracle import d3dp_oracle as orc
    p = orc.strip_prefix(make_state_dict(7, C_, DEPTH, F_))
    sched = orc.cosine_schedule(1000)
    x2d = synthetic_inputs_2d(1234, 1, F_)
    nz = [torch.from_numpy(synthetic_noise(1, (1, 1, F_, J_, 3)))]
    a, b = torch.from_numpy(x2d), torch.from_numpy(flip_2d(x2d))
    times = []
    t_end = time.perf_counter() + budget_s
    while not times or (time.perf_counter() < t_end and len(times) < 8):
        t0 = time.perf_counter()
        with torch.no_grad():
            orc.ddim_sample_flip(p, sched, a, b, 1, 1, DEPTH, H36M_JOINTS_LEFT, H36M_JOINTS_RIGHT, nz)
        times.append(time.perf_counter() - t0)
    t_k1 = sorted(times)[len(times) // 2]
    return {"value": 1.0 / (t_k1 * 10.0), "unit": "hypothesis-clips/s", "cores": torch.get_num_threads(),
            "kind": "port",
            "sample": f"oracle ddim_sample_flip F=243 B=1 H=1 K=1 (2 denoiser calls), median of {len(times)} runs = "
                      f"{t_k1:.3f} s; x10 DDIM steps extrapolated linearly to the K=10 unit",
            "gflops": 2 * flops_per_denoiser_call() / t_k1 / 1e9}


def quick_parity():
    """MPJPE (mm) of both numerics modes vs the CPU oracle on a small full-width problem (F=27,B=1,H=2,K=2)."""
    from d3dp_amd import D3DP
    from d3dp_amd.weights import H36M_JOINTS_LEFT, H36M_JOINTS_RIGHT, flip_2d, make_state_dict, synthetic_inputs_2d, synthetic_noise
    from oracle import d3dp_oracle as orc
    Fr, B, H, K = 27, 1, 2, 2
    sd = make_state_dict(7, C_, DEPTH, Fr)
    x2d = synthetic_inputs_2d(11, B, Fr)
    x2f = flip_2d(x2d)
    nz = [torch.from_numpy(synthetic_noise(20 + k, (B, H, Fr, J_, 3))) for k in range(K)]
    ref = orc.ddim_sample_flip(orc.strip_prefix(sd), orc.cosine_schedule(1000), torch.from_numpy(x2d),
                               torch.from_numpy(x2f), H, K, DEPTH, H36M_JOINTS_LEFT, H36M_JOINTS_RIGHT, nz)
    out = {}
    for numerics in ("exact", "fast"):
        args = SimpleNamespace(number_of_frames=Fr, test_time_augmentation=True, timestep=1000, scale=1.0, cs=C_, dep=DEPTH)
        m = D3DP(args, H36M_JOINTS_LEFT, H36M_JOINTS_RIGHT, is_train=False, num_proposals=H, sampling_timesteps=K,
                 numerics=numerics)
        m.load_state_dict(sd, strict=False)
        m = m.cuda().eval()
        o = m(torch.from_numpy(x2d).cuda(), None, input_2d_flip=torch.from_numpy(x2f).cuda(), noise=nz)
        out[numerics + "_mpjpe_mm"] = orc.mpjpe_mm(o.cpu(), ref)
    out["workload"] = "F=27 B=1 H=2 K=2 cs=512 dep=8, identical weights/inputs/noise, vs CPU oracle"
    out["tolerance_mm"] = 1e-3
    return out


def timed_steps(model, x2d, x2f, steps, warmup, gen, gather):
    import torch.distributed as dist
    from d3dp_amd.dist import all_gather_hypotheses

    def one():
        preds = model(x2d, None, input_2d_flip=x2f, generator=gen)
        return all_gather_hypotheses(preds) if gather else preds

    for _ in range(warmup):
        one()
    if gather:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = one()
    torch.cuda.synchronize()
    if gather:
        dist.barrier()
    dt = time.perf_counter() - t0
    if gather:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=16, help="clips per step (BASELINE config: 16)")
    ap.add_argument("--hyps", type=int, default=20, help="hypotheses per GPU (BASELINE config: 20)")
    ap.add_argument("--ksteps", type=int, default=10, help="DDIM sampling timesteps (BASELINE config: 10)")
    ap.add_argument("--numerics", default="fast", choices=["fast", "exact"])
    ap.add_argument("--chunk-seqs", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-exact-leg", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    a = ap.parse_args()

    from d3dp_amd.dist import init_from_env, rank_generator
    from d3dp_amd.weights import flip_2d, synthetic_inputs_2d
    rank, world, local = init_from_env()
    assert world == a.gpus or world == 1, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local)
    B, H, K = a.batch, a.hyps, a.ksteps

    x2d_np = synthetic_inputs_2d(1234, B, F_)
    x2d = torch.from_numpy(x2d_np).cuda()
    x2f = torch.from_numpy(flip_2d(x2d_np)).cuda()
    gen = rank_generator(1, rank, "cuda")
    model = build_model(H, K, a.numerics, a.chunk_seqs)
    dt, out = timed_steps(model, x2d, x2f, a.steps, a.warmup, gen, gather=world > 1)
    assert out.shape == (B, K, H * world, F_, J_, 3) and bool(torch.isfinite(out).all())
    units = B * H * world * a.steps
    value = units / dt
    flop_per_unit = 2 * K * flops_per_denoiser_call()
    peak = PEAK_BF16_TFLOPS if a.numerics == "fast" else PEAK_F32_TFLOPS

    res = {
        "metric": "pose-hypotheses/sec (H x clips) at F=243, J=17, H=20, K=10",
        "value": value, "unit": "hypothesis-clips/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16" if a.numerics == "fast" else "f32", "data": "synthetic",
        "config": {"workload": f"BASELINE configs[2]: ddim_sample_flip F=243 J=17 B={B} H={H}/GPU K={K} flip-TTA, "
                               f"MixSTE2 cs=512 dep=8 (34.8M params, seed-generated), numerics={a.numerics}",
                   "global_batch": B, "hypotheses_total": H * world, "parallelism": f"hshard{world}",
                   "frame_pose_hypotheses_per_sec": value * F_,
                   "algorithmic_tflop_per_step": flop_per_unit * B * H * world / 1e12,
                   "whole_path_tflops": value * flop_per_unit / 1e12,
                   "whole_path_frac_of_mfma_peak": value * flop_per_unit / 1e12 / (peak * world)},
    }

    if rank == 0 and not a.no_profile:
        # per-kernel HIP-event timing (own stream = torch's current stream) over ONE extra untimed step
        pe = model.pose_estimator
        pe.profile_enable(True)
        model(x2d, None, input_2d_flip=x2f, generator=gen)
        prof = pe.profile_read()
        pe.profile_enable(False)
        total = sum(ms for _, ms in prof.values())
        T_all = 2 * B * H * F_ * J_ * K                       # token-rows through each GEMM class over the step
        gemm_flops = {"gemm_qkv": 2 * 3 * C_ * C_, "gemm_proj": 2 * C_ * C_, "gemm_fc1": 2 * 2 * C_ * C_, "gemm_fc2": 2 * 2 * C_ * C_}
        dom = max(gemm_flops, key=lambda k: prof[k][1])
        n, ms = prof[dom]
        fl = gemm_flops[dom] * T_all * 2 * DEPTH              # 16 blocks
        res["roofline"] = {"bound": "mfma", "kernel": dom, "achieved": fl / (ms * 1e-3) / 1e12, "peak": peak,
                           "unit": "TFLOP/s", "frac": fl / (ms * 1e-3) / 1e12 / peak, "traffic": None,
                           "launches": n, "avg_launch_ms": ms / max(n, 1),
                           "algorithmic_gflop_per_launch": fl / max(n, 1) / 1e9}
        # HBM bytes per launch of the dominant kernel come from a SEPARATE rocprofv3 --pmc pass (FETCH_SIZE doubled per
        # MI355X_MICROARCH.md, WRITE_SIZE), committed under profiles/; valid for the default 15-sequence chunk shape.
        tpath = os.path.join(REPO, "profiles", "r01_gemm_traffic.json")
        if dom == "gemm_qkv" and a.numerics == "fast" and a.chunk_seqs in (0, 15) and os.path.exists(tpath):
            tj = json.load(open(tpath))
            res["roofline"]["traffic"] = tj["hbm_bytes_per_launch"]
            res["roofline"]["traffic_note"] = ("bytes/launch at M=61965 (15-sequence chunk), separate rocprofv3 --pmc pass; "
                                               f"algorithmic bytes/launch {tj['algorithmic_bytes_per_launch']}; "
                                               f"hardware MFMA busy {tj['mfma_util_hw']:.3f} of kernel cycles")
        res["kernel_time_share"] = {k: round(ms / total, 4) for k, (_, ms) in prof.items() if ms > 0}
        res["kernel_ms_per_step"] = {k: round(ms, 3) for k, (_, ms) in prof.items() if ms > 0}

    if rank == 0 and world == 1:
        if not a.no_exact_leg and a.numerics == "fast":
            Be = max(1, B // 8)
            me = build_model(H, K, "exact", a.chunk_seqs)
            dte, _ = timed_steps(me, x2d[:Be].contiguous(), x2f[:Be].contiguous(), 1, 1, gen, gather=False)
            res["exact_mode"] = {"value": Be * H / dte, "unit": "hypothesis-clips/s", "dtype": "f32",
                                 "workload": f"same path, numerics=exact (split-bf16 MFMA Linears, fp32 elsewhere), B={Be} H={H} K={K}, 1 step after 1 warmup",
                                 "whole_path_tflops": Be * H / dte * flop_per_unit / 1e12}
            del me
        res["parity"] = quick_parity()
        if not a.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline()
    if rank == 0:
        print(json.dumps(res))
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
