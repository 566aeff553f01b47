"""CPU ORACLE for the D3DP hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this module, and only as the checker / the CPU number
reported beside the GPU one.  ``d3dp_amd`` never imports it.

What it is: a from-scratch functional restatement (plain torch-CPU tensor ops on a
flat ``{name: tensor}`` dict, no nn.Module) of the reference algorithm

  * cosine schedule + DDIM sampler with flip test-time augmentation
    (reference common/diffusionpose.py:42-52, 75-117, 129-133, 147-169, 214-256,
    260-267, 290-320)
  * MixSTE2 denoiser, eval (B,H,F,J,*) and train (B,F,J,*) branches
    (reference common/mixste.py:24-43, 46-82, 113-115, 127-139, 213-298)

Parity pinning: the reference ships no tests and no golden vectors,
so this oracle is pinned against outputs of the reference ITSELF, produced in the
authoring container by ``tools/make_goldens.py`` (imports /root/reference under a
``timm`` stub, records every random draw) and committed as ``tests/golden/*.npz``.
``tests/test_oracle_vs_goldens.py`` checks the oracle against those fixtures
(bit-exact for the schedule and time pairs, <=1e-3 mm for tensors; observed 0.0).

``dtype=torch.float64`` runs the same algorithm in double precision; tests use it
to measure the fp32 rounding floor, never as the parity target.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
NUM_HEADS = 8          # diffusionpose.py:124
EPS_BLOCK = 1e-6       # mixste.py:163 (partial(nn.LayerNorm, eps=1e-6))
EPS_HEAD = 1e-5        # mixste.py:208 (nn.LayerNorm default)


# --------------------------------------------------------------------------------------
# schedule (diffusionpose.py:42-52, 75-117)
# --------------------------------------------------------------------------------------
def cosine_schedule(timesteps: int = 1000, s: float = 0.008) -> Dict[str, Tensor]:
    """fp64 buffers actually used on the path, keyed by the reference buffer names."""
    steps = timesteps + 1
    x = torch.linspace(0, timesteps, steps, dtype=torch.float64)
    ac = torch.cos(((x / timesteps) + s) / (1 + s) * math.pi * 0.5) ** 2
    ac = ac / ac[0]
    betas = torch.clip(1 - (ac[1:] / ac[:-1]), 0, 0.999)
    alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
    return {
        "betas": betas,
        "alphas_cumprod": alphas_cumprod,
        "sqrt_alphas_cumprod": torch.sqrt(alphas_cumprod),
        "sqrt_one_minus_alphas_cumprod": torch.sqrt(1.0 - alphas_cumprod),
        "sqrt_recip_alphas_cumprod": torch.sqrt(1.0 / alphas_cumprod),
        "sqrt_recipm1_alphas_cumprod": torch.sqrt(1.0 / alphas_cumprod - 1),
    }


def time_pairs(sampling_timesteps: int, total_timesteps: int = 1000):
    """[(T-1, ...), ..., (.., -1)] exactly as diffusionpose.py:221-223."""
    times = torch.linspace(-1, total_timesteps - 1, steps=sampling_timesteps + 1)
    times = list(reversed(times.int().tolist()))
    return list(zip(times[:-1], times[1:]))


# --------------------------------------------------------------------------------------
# denoiser (mixste.py)
# --------------------------------------------------------------------------------------
def sinusoidal_embedding(t: Tensor, dim: int) -> Tensor:
    """mixste.py:132-139.  ``t`` int64 (B,) -> (B, dim) fp32."""
    half = dim // 2
    k = math.log(10000) / (half - 1)
    freq = torch.exp(torch.arange(half) * -k)
    e = t[:, None] * freq[None, :]
    return torch.cat((e.sin(), e.cos()), dim=-1)


def time_mlp(p: Dict[str, Tensor], t: Tensor, cs: int, dtype) -> Tensor:
    """mixste.py:179-184: sinusoid -> Linear(C,2C) -> GELU(erf) -> Linear(2C,C)."""
    e = sinusoidal_embedding(t, cs).to(dtype)
    e = F.linear(e, p["time_mlp.1.weight"], p["time_mlp.1.bias"])
    e = F.gelu(e)
    return F.linear(e, p["time_mlp.3.weight"], p["time_mlp.3.bias"])


# --------------------------------------------------------------------------------------
# bf16 emulation of the library's FAST numerics mode (SURVEY.md §7 hard part 1(b), §8 C4): the SAME algorithm with a
# bf16 rounding wherever the FAST kernels store or consume bf16 -- weights, Linear inputs (LayerNorm / attention /
# GELU outputs), Linear outputs (q|k|v, the two branch outputs), the softmax probabilities entering P.V -- and the
# kernels' polynomial GELU.  fp32 everywhere else (residual stream, LayerNorm statistics, softmax, accumulation).
# It is the tight end-to-end gate of the FAST mode; the fp32 path above stays the parity target.
# --------------------------------------------------------------------------------------
def _r16(x: Tensor) -> Tensor:
    return x.to(torch.bfloat16).to(torch.float32)


def gelu_fast(x: Tensor) -> Tensor:
    """d3dp_amd/csrc/gemm.hip gelu_fast: 0.5 x (1 + u P(u^2)), u = clamp(x, +-3.8) (odd minimax polynomial for erf)."""
    u = x.clamp(-3.8, 3.8)
    t = u * u
    pl = t * 7.331557583256654e-08 + -4.5449246499629226e-06
    for c in (0.0001213696159538813, -0.0018630953272804618, 0.018633270636200905, -0.13143958151340485,
              0.7973535060882568):
        pl = t * pl + c
    h = 0.5 * x
    return h * (u * pl) + h


def attention_bf16(x: Tensor, p: Dict[str, Tensor], pre: str) -> Tensor:
    S, N, C = x.shape
    hd = C // NUM_HEADS
    qkv = _r16(F.linear(_r16(x), _r16(p[pre + "attn.qkv.weight"]), p[pre + "attn.qkv.bias"]))
    qkv = qkv.reshape(S, N, 3, NUM_HEADS, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    a = (q @ k.transpose(-2, -1)) * (hd ** -0.5)
    e = torch.exp(a - a.amax(dim=-1, keepdim=True))           # kernel: exp2 of the scaled difference, fp32
    o = (_r16(e) @ v) / e.sum(dim=-1, keepdim=True)             # bf16 probabilities into P.V, fp32 denominator
    o = _r16(o).transpose(1, 2).reshape(S, N, C)
    return _r16(F.linear(o, _r16(p[pre + "attn.proj.weight"]), p[pre + "attn.proj.bias"]))


def mlp_bf16(x: Tensor, p: Dict[str, Tensor], pre: str) -> Tensor:
    h = _r16(gelu_fast(F.linear(_r16(x), _r16(p[pre + "mlp.fc1.weight"]), p[pre + "mlp.fc1.bias"])))
    return _r16(F.linear(h, _r16(p[pre + "mlp.fc2.weight"]), p[pre + "mlp.fc2.bias"]))


def attention(x: Tensor, p: Dict[str, Tensor], pre: str) -> Tensor:
    """mixste.py:63-82 with comb=False: softmax(q k^T * hd^-0.5) v, then proj."""
    if p.get("__emulate_bf16__") is not None:
        return attention_bf16(x, p, pre)
    S, N, C = x.shape
    hd = C // NUM_HEADS
    qkv = F.linear(x, p[pre + "attn.qkv.weight"], p[pre + "attn.qkv.bias"])
    qkv = qkv.reshape(S, N, 3, NUM_HEADS, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    a = (q @ k.transpose(-2, -1)) * (hd ** -0.5)
    a = a.softmax(dim=-1)
    o = (a @ v).transpose(1, 2).reshape(S, N, C)
    return F.linear(o, p[pre + "attn.proj.weight"], p[pre + "attn.proj.bias"])


def mlp(x: Tensor, p: Dict[str, Tensor], pre: str) -> Tensor:
    """mixste.py:37-43."""
    if p.get("__emulate_bf16__") is not None:
        return mlp_bf16(x, p, pre)
    h = F.gelu(F.linear(x, p[pre + "mlp.fc1.weight"], p[pre + "mlp.fc1.bias"]))
    return F.linear(h, p[pre + "mlp.fc2.weight"], p[pre + "mlp.fc2.bias"])


def block(x: Tensor, p: Dict[str, Tensor], pre: str, keep: Optional[Sequence[Tensor]] = None) -> Tensor:
    """mixste.py:113-115.  ``keep`` = two DropPath masks/(1-rate) of shape (S,1,1) for the
    train path (timm DropPath semantics); None = eval (identity)."""
    C = x.shape[-1]
    a = attention(F.layer_norm(x, (C,), p[pre + "norm1.weight"], p[pre + "norm1.bias"], EPS_BLOCK), p, pre)
    x = x + (a if keep is None else a * keep[0])
    m = mlp(F.layer_norm(x, (C,), p[pre + "norm2.weight"], p[pre + "norm2.bias"], EPS_BLOCK), p, pre)
    return x + (m if keep is None else m * keep[1])


def mixste_forward(p: Dict[str, Tensor], x_2d: Tensor, x_3d: Tensor, t: Tensor, depth: int,
                   droppath: Optional[Dict[str, Sequence[Tensor]]] = None,
                   taps: Optional[Dict[str, Tensor]] = None) -> Tensor:
    """MixSTE2.forward (mixste.py:278-298).

    eval branch: x_2d (B,F,J,2), x_3d (B,H,F,J,3) -> (B,H,F,J,3)
    train branch: x_3d (B,F,J,3) (no H axis)      -> (B,F,J,3)
    ``p`` holds the parameters without the ``pose_estimator.`` prefix.
    ``taps`` (optional dict) receives per-block intermediates for fixture G2.
    """
    dtype = p["head.1.weight"].dtype
    train = x_3d.dim() == 4
    if train:
        x3 = x_3d[:, None]
    else:
        x3 = x_3d
    B, H, Fr, J, _ = x3.shape
    cs = p["Spatial_patch_to_embedding.weight"].shape[0]
    x2 = x_2d[:, None].expand(B, H, Fr, J, 2)
    x = torch.cat((x2, x3), dim=-1).reshape(B * H * Fr, J, 5).to(dtype)      # (b h f) n c
    x = F.linear(x, p["Spatial_patch_to_embedding.weight"], p["Spatial_patch_to_embedding.bias"])
    x = x + p["Spatial_pos_embed"]
    te = time_mlp(p, t, cs, dtype)                                              # (B, C)
    x = x + te[:, None, None, None, :].expand(B, H, Fr, J, cs).reshape(B * H * Fr, J, cs)

    BH = B * H

    def to_temporal(y):   # '(bh f) n c -> (bh n) f c'
        return y.reshape(BH, Fr, J, cs).permute(0, 2, 1, 3).reshape(BH * J, Fr, cs)

    def to_spatial(y):    # '(bh n) f c -> (bh f) n c'
        return y.reshape(BH, J, Fr, cs).permute(0, 2, 1, 3).reshape(BH * Fr, J, cs)

    def dp(name):
        return None if droppath is None else droppath.get(name)

    for i in range(depth):
        x = block(x, p, f"STEblocks.{i}.", dp(f"STEblocks.{i}"))
        x = F.layer_norm(x, (cs,), p["Spatial_norm.weight"], p["Spatial_norm.bias"], EPS_BLOCK)
        if taps is not None:
            taps[f"ste{i}"] = x.reshape(BH, Fr, J, cs).clone()
        x = to_temporal(x)
        if i == 0:
            x = x + p["Temporal_pos_embed"]
        x = block(x, p, f"TTEblocks.{i}.", dp(f"TTEblocks.{i}"))
        x = F.layer_norm(x, (cs,), p["Temporal_norm.weight"], p["Temporal_norm.bias"], EPS_BLOCK)
        x = to_spatial(x)
        if taps is not None:
            taps[f"tte{i}"] = x.reshape(BH, Fr, J, cs).clone()

    x = F.layer_norm(x, (cs,), p["head.0.weight"], p["head.0.bias"], EPS_HEAD)
    x = F.linear(x, p["head.1.weight"], p["head.1.bias"])
    x = x.reshape(B, H, Fr, J, 3)
    return x[:, 0] if train else x


# --------------------------------------------------------------------------------------
# sampler (diffusionpose.py:129-169, 214-256)
# --------------------------------------------------------------------------------------
def flip_pose(x: Tensor, joints_left: List[int], joints_right: List[int]) -> Tensor:
    """x-coordinate negated + left/right joints swapped on the joint axis (dim -2)
    (diffusionpose.py:150-153 / 158-160)."""
    y = x.clone()
    y[..., 0] *= -1
    y[..., joints_left + joints_right, :] = y[..., joints_right + joints_left, :]
    return y


def predict_noise_from_start(sched, x_t: Tensor, t: Tensor, x0: Tensor) -> Tensor:
    """diffusionpose.py:129-133 -- runs in fp64 because the buffers are fp64."""
    shp = (t.shape[0],) + (1,) * (x_t.dim() - 1)
    a = sched["sqrt_recip_alphas_cumprod"].gather(-1, t).reshape(shp)
    b = sched["sqrt_recipm1_alphas_cumprod"].gather(-1, t).reshape(shp)
    return (a * x_t - x0) / b


def model_predictions_flip(p, sched, x, x2d, x2d_flip, t, depth, jl, jr, scale):
    """diffusionpose.py:147-169.  Returns (pred_noise fp32, x_start)."""
    x_t = torch.clamp(x, min=-1.1 * scale, max=1.1 * scale) / scale
    x_t_flip = flip_pose(x_t, jl, jr)
    pred = mixste_forward(p, x2d, x_t, t, depth)
    pred_flip = flip_pose(mixste_forward(p, x2d_flip, x_t_flip, t, depth), jl, jr)
    pred = (pred + pred_flip) / 2
    x_start = torch.clamp(pred * scale, min=-1.1 * scale, max=1.1 * scale)
    pred_noise = predict_noise_from_start(sched, x, t, x_start).to(x.dtype)
    return pred_noise, x_start


def ddim_sample_flip(p, sched, x2d: Tensor, x2d_flip: Tensor, num_proposals: int,
                     sampling_timesteps: int, depth: int, joints_left, joints_right,
                     noises: Sequence[Tensor], scale: float = 1.0, eta: float = 1.0,
                     total_timesteps: int = 1000) -> Tensor:
    """diffusionpose.py:214-256.  ``noises[0]`` replaces ``randn(shape)`` (:225) and
    ``noises[k]`` the k-th ``randn_like(img)`` (:250).  Returns (B, K, H, F, J, 3)."""
    B, Fr = x2d.shape[0], x2d.shape[1]
    img = noises[0].clone()
    assert img.shape == (B, num_proposals, Fr, 17, 3)
    ac = sched["alphas_cumprod"]
    preds = []
    draw = 1
    for time, time_next in time_pairs(sampling_timesteps, total_timesteps):
        t = torch.full((B,), time, dtype=torch.long)
        pred_noise, x_start = model_predictions_flip(p, sched, img, x2d, x2d_flip, t, depth,
                                                     joints_left, joints_right, scale)
        preds.append(x_start)
        if time_next < 0:
            img = x_start
            continue
        alpha, alpha_next = ac[time], ac[time_next]
        sigma = eta * ((1 - alpha / alpha_next) * (1 - alpha_next) / (1 - alpha)).sqrt()
        c = (1 - alpha_next - sigma ** 2).sqrt()
        noise = noises[draw]
        draw += 1
        img = x_start * alpha_next.sqrt() + c * pred_noise + sigma * noise
    return torch.stack(preds, dim=1)


def q_sample(sched, x_start: Tensor, t: Tensor, noise: Tensor) -> Tensor:
    """diffusionpose.py:260-267 (fp64 result)."""
    shp = (t.shape[0],) + (1,) * (x_start.dim() - 1)
    a = sched["sqrt_alphas_cumprod"].gather(-1, t).reshape(shp)
    b = sched["sqrt_one_minus_alphas_cumprod"].gather(-1, t).reshape(shp)
    return a * x_start + b * noise


def prepare_targets(sched, targets: Tensor, ts: Tensor, noises: Tensor, scale: float = 1.0):
    """diffusionpose.py:290-320 with recorded draws: per sample t (1,), noise (F,17,3)."""
    outs = []
    for i in range(targets.shape[0]):
        x = q_sample(sched, targets[i] * scale, ts[i].reshape(1), noises[i])
        x = torch.clamp(x, min=-1.1 * scale, max=1.1 * scale) / scale
        outs.append(x)
    return torch.stack(outs).float()


def emulate_bf16(p: Dict[str, Tensor]) -> Dict[str, Tensor]:
    """Copy of a parameter dict that makes every function above follow the FAST mode's bf16 roundings."""
    q = dict(p)
    q["__emulate_bf16__"] = torch.ones(())
    return q


def strip_prefix(sd: Dict[str, Tensor], prefix: str = "pose_estimator.", dtype=None) -> Dict[str, Tensor]:
    out = {}
    for k, v in sd.items():
        if k.startswith("module."):
            k = k[len("module."):]
        if k.startswith(prefix):
            out[k[len(prefix):]] = v if dtype is None else v.to(dtype)
    return out


def mpjpe_mm(a: Tensor, b: Tensor) -> float:
    """Mean per-joint position error between two pose tensors in metres, reported in mm
    (loss.py:13 times 1000)."""
    return float(torch.mean(torch.norm(a.double() - b.double(), dim=-1)) * 1000.0)
