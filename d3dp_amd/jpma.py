"""Caller-side steps either side of the hot path, device-agnostic torch code.

These are the "next" rows of the scope table.  The JPMA selection itself (root zeroing + trajectory add + projection +
per-joint argmin over hypotheses + gather) is a fused HIP kernel (``jpma_hip`` -> d3dp_jpma); the remaining functions
are small torch steps over (B,K,H,F,17,*) tensors that run on whatever device the sampler output lives on and double
as the readable statement of what the kernel computes.  They exist so that
the `--evaluate` entrypoint and the multi-GPU path have their consumer, and they are pinned against the
reference by fixture g5 (tests/test_caller_side.py).

  eval_data_prepare   reference main.py:267-299   (clip chunking; last clip = last F frames, replicate-pad short)
  project_to_2d       reference common/camera.py:30-60
  jpma_metrics        reference main.py:700-718 + common/loss.py:22-107  (J_Best, P_Best, P_Agg, J_Agg errors)
  jpma_aggregate      reference main_3dhp.py:781-835 semantics (per-joint selection by reprojection error -> poses)
"""
from __future__ import annotations

from typing import Dict, Tuple

import torch


def clip_starts(n_frames: int, receptive_field: int):
    """Start index of every clip main.py:267-299 cuts a sequence into."""
    if n_frames <= receptive_field:
        return [0]
    n = n_frames // receptive_field
    starts = [i * receptive_field for i in range(n)]
    if n_frames % receptive_field:
        starts.append(n_frames - receptive_field)
    return starts


def eval_data_prepare(receptive_field: int, inputs_2d: torch.Tensor, inputs_3d: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """(1,N,J,2),(1,N,J,3) or (N,J,*) -> (n_clips, F, J, *) like main.py:267-299."""
    assert inputs_2d.shape[:-1] == inputs_3d.shape[:-1], "2d and 3d inputs shape must be same!"
    a, b = inputs_2d.reshape(-1, *inputs_2d.shape[-2:]), inputs_3d.reshape(-1, *inputs_3d.shape[-2:])
    n = a.shape[0]
    if n < receptive_field:      # replicate-pad on the right (main.py:284-295)
        pad = receptive_field - n
        a = torch.cat((a, a[-1:].expand(pad, -1, -1)), dim=0)
        b = torch.cat((b, b[-1:].expand(pad, -1, -1)), dim=0)
    starts = clip_starts(n, receptive_field)
    return (torch.stack([a[s:s + receptive_field] for s in starts]),
            torch.stack([b[s:s + receptive_field] for s in starts]))


def project_to_2d(X: torch.Tensor, camera_params: torch.Tensor) -> torch.Tensor:
    """Human3.6M projection with radial + tangential distortion (camera.py:30-60).
    X (..., J, 3) in camera space, camera_params (9,) or broadcastable (..., 9): f(2) c(2) k(3) p(2)."""
    cam = camera_params
    while cam.dim() < X.dim():
        cam = cam.unsqueeze(-2)
    f, c, k, p = cam[..., :2], cam[..., 2:4], cam[..., 4:7], cam[..., 7:]
    XX = torch.clamp(X[..., :2] / X[..., 2:], min=-1, max=1)
    r2 = torch.sum(XX ** 2, dim=-1, keepdim=True)
    radial = 1 + torch.sum(k * torch.cat((r2, r2 ** 2, r2 ** 3), dim=-1), dim=-1, keepdim=True)
    tan = torch.sum(p * XX, dim=-1, keepdim=True)
    return f * (XX * (radial + tan) + p * r2) + c


def reproject(pred: torch.Tensor, traj: torch.Tensor, cam: torch.Tensor) -> torch.Tensor:
    """pred (B,K,H,F,J,3) root-relative, traj (B,F,1,3), cam (9,) -> (B,K,H,F,J,2)   (main.py:706-712)."""
    absol = pred + traj[:, None, None]
    return project_to_2d(absol, cam.reshape(-1)[:9])


def jpma_metrics(pred: torch.Tensor, gt: torch.Tensor, reproj: torch.Tensor, gt_2d: torch.Tensor) -> Dict[str, torch.Tensor]:
    """Per-step errors (K,) in metres for the four aggregation modes of loss.py:22-107.
    pred (B,K,H,F,J,3), gt (B,F,J,3), reproj (B,K,H,F,J,2), gt_2d (B,F,J,2)."""
    K = pred.shape[1]
    err = torch.norm(pred - gt[:, None, None], dim=-1)                     # (B,K,H,F,J)
    err2d = torch.norm(reproj - gt_2d[:, None, None], dim=-1)
    j_best = err.min(dim=2).values.permute(1, 0, 2, 3).reshape(K, -1).mean(-1)                 # loss.py:38-43
    p_best = err.permute(1, 2, 0, 3, 4).reshape(K, err.shape[2], -1).mean(-1).min(dim=1).values  # loss.py:93-96
    p_agg = torch.norm(pred.mean(dim=2) - gt[:, None], dim=-1).permute(1, 0, 2, 3).reshape(K, -1).mean(-1)  # :45-52
    sel = err2d.min(dim=2, keepdim=True).indices                           # first minimal h, like torch.min
    j_agg = torch.gather(err, 2, sel).permute(1, 2, 0, 3, 4).reshape(K, -1).mean(-1)             # loss.py:70-76
    return {"J_Best": j_best, "P_Best": p_best, "P_Agg": p_agg, "J_Agg": j_agg}


def jpma_hip(pred: torch.Tensor, traj: torch.Tensor, cam: torch.Tensor, gt_2d: torch.Tensor,
             gt_3d: torch.Tensor = None, zero_root: bool = True, want_errors: bool = False):
    """The fused HIP kernel (include/d3dp_hip.h: d3dp_jpma) for GPU tensors: returns the aggregated poses (B,K,F,J,3),
    the selected hypothesis index (B,K,F,J) and, with ``want_errors``, the per-joint J_Agg / J_Best errors."""
    import ctypes as C
    from . import _lib
    lib = _lib.load()
    assert pred.is_cuda, "jpma_hip needs GPU tensors (no CPU fallback); use jpma_aggregate for host tensors"
    B, K, H, Fr, J, _ = pred.shape
    pred = pred.to(torch.float32).contiguous()
    traj = traj.to(torch.float32).reshape(B, Fr, 3).contiguous()
    cam = cam.to(device=pred.device, dtype=torch.float32).reshape(-1)[:9].contiguous()
    gt_2d = gt_2d.to(torch.float32).contiguous()
    agg = torch.empty((B, K, Fr, J, 3), dtype=torch.float32, device=pred.device)
    sel = torch.empty((B, K, Fr, J), dtype=torch.int32, device=pred.device)
    es = em = None
    if want_errors:
        assert gt_3d is not None
        gt_3d = gt_3d.to(torch.float32).contiguous()
        es = torch.empty((B, K, Fr, J), dtype=torch.float32, device=pred.device)
        em = torch.empty_like(es)
    with torch.cuda.device(pred.device):
        _lib.check(lib.d3dp_jpma(pred.data_ptr(), traj.data_ptr(), cam.data_ptr(), gt_2d.data_ptr(), _lib.ptr(gt_3d),
                                 agg.data_ptr(), sel.data_ptr(), _lib.ptr(es), _lib.ptr(em), B, K, H, Fr, J,
                                 int(zero_root), _lib.current_stream()), "d3dp_jpma")
    return (agg, sel, es, em) if want_errors else (agg, sel)


def jpma_aggregate(pred: torch.Tensor, reproj: torch.Tensor, gt_2d: torch.Tensor) -> torch.Tensor:
    """Joint-wise reprojection-based multi-hypothesis aggregation: for every (clip, step, frame, joint) keep the
    hypothesis whose reprojection is closest to the 2D input -> (B,K,F,J,3)."""
    err2d = torch.norm(reproj - gt_2d[:, None, None], dim=-1)
    sel = err2d.min(dim=2, keepdim=True).indices
    return torch.gather(pred, 2, sel[..., None].expand(-1, -1, -1, -1, -1, 3))[:, :, 0]
