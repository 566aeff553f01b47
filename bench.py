#!/usr/bin/env python3
"""Headline benchmark of the D3DP hot path on MI355X (contract: see the round prompt / DESIGN.md §Measurement).

One "step" = one D3DP.forward (ddim_sample_flip: K DDIM steps x 2 flip-TTA denoiser passes) over one synthetic batch
of B clips x H hypotheses at F=243, J=17.  N=1 workload = BASELINE.json configs[2] (B=16, H=20, K=10), the
configuration the metric is quoted on.

The HEADLINE (`value`) is the parity-compliant numerics mode, `exact`: the mode that meets north_star's <= 1e-3 mm
against the reference (Linears on split-fp16 operands, three fp16-MFMA passes, fp32 accumulate; fp32 elsewhere).  The
single-pass bf16 mode (`fast`, millimetres away from the reference on random weights) is timed on the same workload
and reported beside it as `fast_mode`, never as `value`.

`python bench.py --gpus N` launches its own N ranks (re-executes itself under torch.distributed.run) when it is not
already inside a torchrun job; every rank samples its own H=20 hypotheses of the same clips (weak scaling, no
data-path collective during sampling) and one RCCL all-gather assembles (B,K,N*20,F,17,3) on every rank inside the
timed region.  Inputs are resident in HBM before the timed region starts.  Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import socket
import sys
import time
from types import SimpleNamespace

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

F_, J_, C_, DEPTH = 243, 17, 512, 8
PEAK_MFMA_TFLOPS = 2500.0      # dense bf16 / fp16 MFMA, MI355X_MICROARCH.md (2.5 PFLOP/s)
MFMA_STREAM_PFLOPS = 1.5    # measured rate of the EXACT Linear's MFMA stream alone: 1.57 (r02_gemm_probes.md), 1.45-1.60 (r03_gemm_mfma_shape.md)
EXACT_PASSES = 3               # fp16-MFMA products per algorithmic multiply-add in exact mode (hi.hi, hi.lo, lo.hi)


def flops_per_denoiser_call(frames=F_, joints=J_, c=C_, depth=DEPTH):
    """Algorithmic FLOP of one MixSTE2.forward per (clip, hypothesis) -- SURVEY.md §8 D4 (multiply-add = 2)."""
    tseq = frames * joints
    per_tok = depth * 2 * 8 * c * c * 2              # 16 blocks x (3+1+2+2) C^2 MACs x 2 = 16*16*C^2
    attn = depth * 4 * joints * c + depth * 4 * frames * c
    return tseq * (per_tok + attn + 2 * 5 * c + 2 * c * 3) + 2 * 2 * c * 2 * c


def build_model(H, K, numerics, chunk_seqs, frames=F_):
    from d3dp_amd import D3DP
    from d3dp_amd.weights import H36M_JOINTS_LEFT, H36M_JOINTS_RIGHT, make_state_dict
    args = SimpleNamespace(number_of_frames=frames, test_time_augmentation=True, timestep=1000, scale=1.0, cs=C_,
                           dep=DEPTH, chunk_seqs=chunk_seqs)
    m = D3DP(args, H36M_JOINTS_LEFT, H36M_JOINTS_RIGHT, is_train=False, num_proposals=H, sampling_timesteps=K,
             numerics=numerics)
    m.load_state_dict(make_state_dict(7, C_, DEPTH, frames), strict=False)
    return m.cuda().eval()


def cpu_baseline(budget_s=24.0):
    """The CPU oracle (a port of the reference path; the reference's own files cannot travel to the GPU box and
    hard-code CUDA) timed on this host's cores on a BOUNDED sample of the workload: F=243, B=1, H=1, K=1 (= 2 denoiser
    calls of the K=10 unit's 20).  The thread count is swept (oversubscribing a 128-thread host halves the rate) and the
    best is reported with its count; the K=10 unit is K=1 x 10, stated as such."""
    import torch
    from d3dp_amd.weights import H36M_JOINTS_LEFT, H36M_JOINTS_RIGHT, flip_2d, make_state_dict, synthetic_inputs_2d, synthetic_noise
    from oracle import d3dp_oracle as orc
    p = orc.strip_prefix(make_state_dict(7, C_, DEPTH, F_))
    sched = orc.cosine_schedule(1000)
    x2d = synthetic_inputs_2d(1234, 1, F_)
    nz = [torch.from_numpy(synthetic_noise(1, (1, 1, F_, J_, 3)))]
    a, b = torch.from_numpy(x2d), torch.from_numpy(flip_2d(x2d))

    def one():
        t0 = time.perf_counter()
        with torch.no_grad():
            orc.ddim_sample_flip(p, sched, a, b, 1, 1, DEPTH, H36M_JOINTS_LEFT, H36M_JOINTS_RIGHT, nz)
        return time.perf_counter() - t0

    ncpu = os.cpu_count() or 1
    t_start = time.perf_counter()
    old = torch.get_num_threads()
    sweep = {}
    one()                                                     # page in the weights / thread pool
    for nt in [n for n in (8, 16, 32, 64, 128) if n <= ncpu] or [ncpu]:
        torch.set_num_threads(nt)
        sweep[nt] = one()
        if sweep[nt] < 3.0 and time.perf_counter() - t_start < budget_s * 0.6:
            sweep[nt] = min(sweep[nt], one())
        # oversubscription only gets worse from here (256-thread hosts: 0.7 s at 16 threads, 4 s at 128, 100 s at 256)
        if time.perf_counter() - t_start > budget_s or sweep[nt] > 2.5 * min(sweep.values()):
            break
    best = min(sweep, key=sweep.get)
    torch.set_num_threads(best)
    times = [sweep[best]]
    while time.perf_counter() - t_start < budget_s and len(times) < 5:
        times.append(one())
    torch.set_num_threads(old)
    t_k1 = sorted(times)[len(times) // 2]
    return {"value": 1.0 / (t_k1 * 10.0), "unit": "hypothesis-clips/s", "cores": best, "kind": "port",
            "sample": f"oracle ddim_sample_flip F=243 B=1 H=1 K=1 (2 of the unit's 20 denoiser calls), median of "
                      f"{len(times)} runs at the best thread count = {t_k1:.3f} s; x10 DDIM steps to the K=10 unit",
            "host_cpus": ncpu, "thread_sweep_s": {str(k): round(v, 3) for k, v in sweep.items()},
            "gflops": 2 * flops_per_denoiser_call() / t_k1 / 1e9,
            "full_config_run": full_config_cpu_run()}


def full_config_cpu_run():
    """The UN-extrapolated CPU run of BASELINE configs[1] (B=4 H=5 K=5 F=243; tools/cpu_c2.py, committed under
    profiles/): per FLOP it is 2.3x SLOWER than the B=1 H=1 sample above (the oracle, like the reference, materialises
    the attention scores of the whole batch), i.e. the bounded sample OVERSTATES the CPU at a real batch size.  Its
    hypothesis-clips/s are in K=5 units; `k10_units_per_s` rescales by FLOP to this benchmark's K=10 unit."""
    path = os.path.join(REPO, "profiles", "r02_cpu_c2.json")
    if not os.path.exists(path):
        return None
    j = json.load(open(path))
    return {"source": "profiles/r02_cpu_c2.json (tools/cpu_c2.py, measured on a GPU box's host, not in this run)",
            "workload": j["workload"], "threads": j["threads"], "seconds": j["seconds"], "gflops": j["gflops"],
            "hypothesis_clips_per_s_K5_units": j["hypothesis_clips_per_s"],
            "k10_units_per_s": j["hypothesis_clips_per_s"] / 2.0}


def quick_parity():
    """MPJPE (mm) of both numerics modes on a small full-width problem (F=27,B=1,H=2,K=2) against the fp32 CPU oracle
    (the parity target) and, for fast mode, against the oracle's bf16-rounding emulation of the FAST kernels."""
    import torch
    from d3dp_amd import D3DP
    from d3dp_amd.weights import H36M_JOINTS_LEFT, H36M_JOINTS_RIGHT, flip_2d, make_state_dict, synthetic_inputs_2d, synthetic_noise
    from oracle import d3dp_oracle as orc
    Fr, B, H, K = 27, 1, 2, 2
    sd = make_state_dict(7, C_, DEPTH, Fr)
    x2d = synthetic_inputs_2d(11, B, Fr)
    x2f = flip_2d(x2d)
    nz = [torch.from_numpy(synthetic_noise(20 + k, (B, H, Fr, J_, 3))) for k in range(K)]
    p = orc.strip_prefix(sd)
    run = lambda q: orc.ddim_sample_flip(q, orc.cosine_schedule(1000), torch.from_numpy(x2d), torch.from_numpy(x2f), H, K,
                                         DEPTH, H36M_JOINTS_LEFT, H36M_JOINTS_RIGHT, nz)
    ref, ref16 = run(p), run(orc.emulate_bf16(p))
    out = {}
    for numerics in ("exact", "fast"):
        args = SimpleNamespace(number_of_frames=Fr, test_time_augmentation=True, timestep=1000, scale=1.0, cs=C_, dep=DEPTH)
        m = D3DP(args, H36M_JOINTS_LEFT, H36M_JOINTS_RIGHT, is_train=False, num_proposals=H, sampling_timesteps=K,
                 numerics=numerics)
        m.load_state_dict(sd, strict=False)
        m = m.cuda().eval()
        o = m(torch.from_numpy(x2d).cuda(), None, input_2d_flip=torch.from_numpy(x2f).cuda(), noise=nz).cpu()
        out[numerics + "_mpjpe_mm"] = orc.mpjpe_mm(o, ref)
        if numerics == "fast":
            out["fast_vs_bf16_oracle_mm"] = orc.mpjpe_mm(o, ref16)
    out["workload"] = "F=27 B=1 H=2 K=2 cs=512 dep=8, identical weights/inputs/noise, vs CPU oracle"
    out["tolerance_mm"] = 1e-3
    out["exact_meets_tolerance"] = out["exact_mpjpe_mm"] <= 1e-3
    out["fast_meets_tolerance"] = out["fast_mpjpe_mm"] <= 1e-3
    return out


def device_sync(dev):
    import torch
    if dev == "cuda":
        torch.cuda.synchronize()


def timed_steps(model, x2d, x2f, steps, warmup, gen, gather, dev="cuda"):
    import torch
    import torch.distributed as dist
    from d3dp_amd.dist import all_gather_hypotheses

    def one():
        preds = model(x2d, None, input_2d_flip=x2f, generator=gen)
        return all_gather_hypotheses(preds) if gather else preds

    for _ in range(warmup):
        one()
    if gather:
        dist.barrier()
    device_sync(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        out = one()
    device_sync(dev)
    if gather:
        dist.barrier()
    dt = time.perf_counter() - t0
    if gather:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt, out


class DryRunSampler:
    """NOT the product path: a CPU stand-in with the sampler's call signature and independence structure (hypothesis h
    depends only on x2d and noise[:, h]; tests/test_dist_cpu.py uses the same form), used ONLY by `--dist-dry-run N` to
    drive this file's multi-rank control flow -- self-launch under torch.distributed.run, WORLD_SIZE check, shard_check,
    timed loop with the all-gather inside, MAX-reduce of the time, the `multi_gpu` block -- over gloo on a box without
    GPUs.  Its numbers say nothing about the library."""

    def __init__(self, H, K, frames):
        self.H, self.K, self.frames = H, K, frames

    def __call__(self, x2d, _x3d, input_2d_flip=None, generator=None, noise=None):
        import torch
        B = x2d.shape[0]
        if noise is None:
            noise = [torch.randn(B, self.H, self.frames, J_, 3, generator=generator) for _ in range(self.K)]
        m = (x2d.mean(dim=-1, keepdim=True) - input_2d_flip.mean(dim=-1, keepdim=True))[:, None]
        return torch.stack([torch.tanh(n * 0.3 + m * (k + 1)) for k, n in enumerate(noise)], dim=1)


def shard_check(rank, world, numerics, dev="cuda", make=None):
    """N ranks, each sampling its slice of GLOBAL noise draws (dist.shard_noise), must reproduce the 1-rank run with
    H_total = N * H_local hypotheses bit for bit (small problem: F=27, B=2, H_local=2, K=2; rank 0 runs the 1-rank form)."""
    import torch
    from d3dp_amd.dist import all_gather_hypotheses, shard_noise
    from d3dp_amd.weights import flip_2d, synthetic_inputs_2d, synthetic_noise
    Fr, B, Hl, K = 27, 2, 2, 2
    x2d_np = synthetic_inputs_2d(77, B, Fr)
    make = make or (lambda H_, K_, fr: build_model(H_, K_, numerics, 0, frames=fr))
    x2d, x2f = torch.from_numpy(x2d_np).to(dev), torch.from_numpy(flip_2d(x2d_np)).to(dev)
    noise = [torch.from_numpy(synthetic_noise(90 + k, (B, world * Hl, Fr, J_, 3))) for k in range(K)]
    if dev != "cuda":
        noise = [n.to(dev) for n in noise]
    local = make(Hl, K, Fr)(x2d, None, input_2d_flip=x2f, noise=shard_noise(noise, rank, world))
    got = all_gather_hypotheses(local)
    ok = True
    if rank == 0:
        want = make(world * Hl, K, Fr)(x2d, None, input_2d_flip=x2f, noise=noise)
        ok = bool(torch.equal(got, want))
    return ok


def roofline_from_profile(prof, numerics, B, H, K):
    T_all = 2 * B * H * F_ * J_ * K                       # token-rows through each GEMM class over the step
    gemm_flops = {"gemm_qkv": 2 * 3 * C_ * C_, "gemm_proj": 2 * C_ * C_, "gemm_fc1": 2 * 2 * C_ * C_, "gemm_fc2": 2 * 2 * C_ * C_}
    dom = max(gemm_flops, key=lambda k: prof[k][1])
    n, ms = prof[dom]
    fl = gemm_flops[dom] * T_all * 2 * DEPTH              # 16 blocks
    ach = fl / (ms * 1e-3) / 1e12
    passes = EXACT_PASSES if numerics == "exact" else 1
    note = "peak = dense fp16 / bf16 MFMA peak of MI355X_MICROARCH.md (2.5 PFLOP/s); achieved = ALGORITHMIC FLOP / HIP-event time"
    if numerics == "exact":
        note += (f"; every algorithmic product costs {EXACT_PASSES} fp16-MFMA products (split-fp16 operands: hi.hi + hi.lo + "
                 f"lo.hi), so the matrix pipes do {EXACT_PASSES} x achieved = `matrix_pipe_work_tflops` -- reported beside "
                 f"`frac`, never instead of it.  Under this instruction stream the chip clocks at 1.7-1.9 GHz, not 2.4; the "
                 f"kernel's bare MFMA stream (no loads, LDS reads, stores or barriers) runs at {MFMA_STREAM_PFLOPS} PFLOP/s "
                 f"of matrix work (profiles/r02_gemm_probes.md, profiles/r03_gemm_mfma_shape.md)")
    r = {"bound": "mfma", "kernel": dom, "achieved": ach, "peak": PEAK_MFMA_TFLOPS, "unit": "TFLOP/s",
         "frac": ach / PEAK_MFMA_TFLOPS, "frac_algorithmic_of_dense_peak": ach / PEAK_MFMA_TFLOPS,
         "mfma_passes_per_product": passes, "matrix_pipe_work_tflops": ach * passes,
         "matrix_pipe_work_frac_of_dense_peak": ach * passes / PEAK_MFMA_TFLOPS,
         "traffic": None, "launches": n, "avg_launch_ms": ms / max(n, 1), "algorithmic_gflop_per_launch": fl / max(n, 1) / 1e9,
         "peak_note": note}
    if numerics == "exact":
        r["matrix_pipe_work_frac_of_measured_mfma_stream"] = ach * EXACT_PASSES / (MFMA_STREAM_PFLOPS * 1e3)
    return r


def lib_sha256():
    from d3dp_amd import _lib
    h = hashlib.sha256()
    with open(_lib.LIB_PATH, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


def attach_traffic(roof, numerics, chunk_seqs):
    """HBM bytes per launch of the dominant kernel come from SEPARATE rocprofv3 --pmc passes (FETCH_SIZE doubled per
    MI355X_MICROARCH.md, WRITE_SIZE) stored under profiles/ next to the hash of the library they were measured on: the
    number is printed only when this run uses that very build."""
    sha, tj, seen = lib_sha256(), None, []
    for name in ("r03_gemm_traffic.json", "r02_gemm_traffic.json"):        # newest first; keyed by the library's hash
        tpath = os.path.join(REPO, "profiles", name)
        if os.path.exists(tpath):
            cand = json.load(open(tpath)).get(numerics, {}).get(roof["kernel"])
            if cand:
                seen.append(name)
                if cand.get("lib_sha256") == sha:
                    tj = cand
                    break
    if chunk_seqs not in (0, 15) or not seen:
        return
    if tj is None:
        roof["traffic_note"] = f"profiles/{seen[0]} was measured on a different build of libd3dp_hip.so: not reported"
        return
    roof["traffic"] = tj["hbm_bytes_per_launch"]
    roof["traffic_note"] = (f"HBM bytes/launch (FETCH_SIZE x2 + WRITE_SIZE) at a mean of {tj.get('mean_rows_per_launch', 0):.0f} "
                            f"rows per launch, separate rocprofv3 --pmc passes of this build inside the denoiser; algorithmic "
                            f"bytes/launch {tj['algorithmic_bytes_per_launch']} (read amplification "
                            f"{tj.get('read_amplification', float('nan')):.2f}x: the weight matrix is re-fetched by every XCD "
                            f"every tile round); hardware MFMA busy {tj.get('mfma_util_hw', float('nan')):.3f} of kernel cycles")


def profile_step(model, x2d, x2f, gen):
    pe = model.pose_estimator
    pe.profile_enable(True)
    model(x2d, None, input_2d_flip=x2f, generator=gen)
    prof = pe.profile_read()
    pe.profile_enable(False)
    return prof


def relaunch_under_torchrun(n):
    """`python bench.py --gpus N` outside a torchrun job: become `python -m torch.distributed.run ... bench.py <same args>`."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    os.execv(sys.executable, cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=16, help="clips per step (BASELINE config: 16)")
    ap.add_argument("--hyps", type=int, default=20, help="hypotheses per GPU (BASELINE config: 20)")
    ap.add_argument("--ksteps", type=int, default=10, help="DDIM sampling timesteps (BASELINE config: 10)")
    ap.add_argument("--numerics", default="exact", choices=["exact", "fast"], help="mode timed as the headline")
    ap.add_argument("--chunk-seqs", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-leg", action="store_true", help="skip timing the other numerics mode")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--dist-dry-run", type=int, default=0, metavar="N",
                    help="no GPU needed: drive the N-rank control flow of this file (self-launch, WORLD_SIZE check, shard "
                         "check, timed loop with the all-gather, MAX-reduce, multi_gpu block) over gloo with a CPU "
                         "stand-in for the sampler; the JSON line is marked dry_run and measures nothing")
    a = ap.parse_args()
    dry = a.dist_dry_run > 0
    if dry:
        a.gpus = a.dist_dry_run
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        relaunch_under_torchrun(a.gpus)

    import torch
    from d3dp_amd.dist import init_from_env, rank_generator
    from d3dp_amd.weights import flip_2d, synthetic_inputs_2d
    if int(os.environ.get("WORLD_SIZE", "1")) != a.gpus:      # before any rendezvous: a mismatched job must not hang
        raise SystemExit(f"bench.py: --gpus {a.gpus} but the job has WORLD_SIZE={os.environ.get('WORLD_SIZE', '1')}")
    rank, world, local = init_from_env("gloo" if dry else None)
    dev = "cpu" if dry else "cuda"
    if not dry:
        torch.cuda.set_device(local)
    B, H, K = a.batch, a.hyps, a.ksteps
    frames = F_
    if dry:
        B, H, K, frames = min(B, 2), min(H, 2), min(K, 2), 27
        a.no_profile = True
    make = (lambda H_, K_, fr: DryRunSampler(H_, K_, fr)) if dry else None

    x2d_np = synthetic_inputs_2d(1234, B, frames)
    x2d = torch.from_numpy(x2d_np).to(dev)
    x2f = torch.from_numpy(flip_2d(x2d_np)).to(dev)
    gen = rank_generator(1, rank, dev)
    sharding_ok = shard_check(rank, world, a.numerics, dev, make) if world > 1 else None
    model = DryRunSampler(H, K, frames) if dry else build_model(H, K, a.numerics, a.chunk_seqs)
    dt, out = timed_steps(model, x2d, x2f, a.steps, a.warmup, gen, gather=world > 1, dev=dev)
    assert out.shape == (B, K, H * world, frames, J_, 3) and bool(torch.isfinite(out).all())
    units = B * H * world * a.steps
    value = units / dt
    flop_per_unit = 2 * K * flops_per_denoiser_call()
    peak = PEAK_MFMA_TFLOPS
    dtypes = {"exact": "f16x2-split operands on fp16 MFMA, f32 accumulate (fp32-class: meets 1e-3 mm)",
              "fast": "bf16 operands, f32 accumulate (does NOT meet 1e-3 mm)"}

    res = {
        "metric": "pose-hypotheses/sec (H x clips) at F=243, J=17, H=20, K=10",
        "value": value, "unit": "hypothesis-clips/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": dtypes[a.numerics], "data": "synthetic",
        "config": {"workload": f"BASELINE configs[2]: ddim_sample_flip F=243 J=17 B={B} H={H}/GPU K={K} flip-TTA, "
                               f"MixSTE2 cs=512 dep=8 (34.8M params, seed-generated), numerics={a.numerics}",
                   "numerics": a.numerics, "parity_gate_mm": 1e-3 if a.numerics == "exact" else None,
                   "global_batch": B, "hypotheses_total": H * world, "parallelism": f"hshard{world}",
                   "frame_pose_hypotheses_per_sec": value * F_,
                   "algorithmic_tflop_per_step": flop_per_unit * B * H * world / 1e12,
                   "whole_path_tflops": value * flop_per_unit / 1e12,
                   "whole_path_frac_of_mfma_peak": value * flop_per_unit / 1e12 / (peak * world),
                   "mfma_peak_used_tflops": peak},
    }
    if world > 1:
        import torch.distributed as dist
        from d3dp_amd.dist import all_gather_hypotheses
        local_preds = out[:, :, rank * H:(rank + 1) * H].contiguous()
        device_sync(dev); dist.barrier()
        t0 = time.perf_counter()
        for _ in range(5):
            all_gather_hypotheses(local_preds)
        device_sync(dev)
        ag_ms = (time.perf_counter() - t0) / 5 * 1e3
        flag = torch.tensor([1 if sharding_ok else 0], device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        res["multi_gpu"] = {"backend": "gloo (dry run)" if dry else "nccl (RCCL)", "world_size": dist.get_world_size(),
                            "all_gather_bytes_per_rank": local_preds.numel() * 4, "all_gather_ms": ag_ms,
                            "all_gather_gbps_per_rank_out": local_preds.numel() * 4 * (world - 1) / (ag_ms * 1e-3) / 1e9,
                            "sharded_equals_single_rank": bool(flag.item()),
                            "shard_check": "F=27 B=2 H_local=2 K=2: N ranks on sliced global noise == 1 rank with H=2N, torch.equal"}
        assert bool(flag.item()), "N-rank sampling does not reproduce the 1-rank run"

    if rank == 0 and not a.no_profile:
        # per-kernel HIP-event timing (on the launch stream = torch's current stream) over ONE extra untimed step
        prof = profile_step(model, x2d, x2f, gen)
        total = sum(ms for _, ms in prof.values())
        res["roofline"] = roofline_from_profile(prof, a.numerics, B, H, K)
        attach_traffic(res["roofline"], a.numerics, a.chunk_seqs)
        res["kernel_time_share"] = {k: round(ms / total, 4) for k, (_, ms) in prof.items() if ms > 0}
        res["kernel_ms_per_step"] = {k: round(ms, 3) for k, (_, ms) in prof.items() if ms > 0}

    if dry:
        res.update({"dry_run": True, "value": None, "ms_per_step": None, "data": "none (dry run)",
                    "dtype": "none: CPU stand-in sampler over gloo, control flow only",
                    "config": {"workload": f"DRY RUN of the {world}-rank control flow (B={B} H={H}/rank K={K} F={frames}); "
                                           f"measures nothing", "parallelism": f"hshard{world}"}})
    if rank == 0 and world == 1 and not dry:
        if not a.no_other_leg:
            other = "fast" if a.numerics == "exact" else "exact"
            del model
            torch.cuda.empty_cache()
            mo = build_model(H, K, other, a.chunk_seqs)
            st, wu = max(1, min(a.steps, 5)), max(1, min(a.warmup, 2))
            dto, _ = timed_steps(mo, x2d, x2f, st, wu, gen, gather=False)
            vo = B * H * st / dto
            leg = {"value": vo, "unit": "hypothesis-clips/s", "numerics": other, "dtype": dtypes[other], "steps": st,
                   "warmup": wu, "ms_per_step": dto / st * 1e3,
                   "workload": f"the same BASELINE configs[2] workload (B={B} H={H} K={K}) in numerics={other}",
                   "whole_path_tflops": vo * flop_per_unit / 1e12}
            if not a.no_profile:
                po = profile_step(mo, x2d, x2f, gen)
                leg["roofline"] = roofline_from_profile(po, other, B, H, K)
                attach_traffic(leg["roofline"], other, a.chunk_seqs)
                leg["kernel_ms_per_step"] = {k: round(ms, 3) for k, (_, ms) in po.items() if ms > 0}
            res[other + "_mode"] = leg
            del mo
        if not a.no_parity:
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
