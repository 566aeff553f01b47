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


def mpjpe_diffusion(pred: torch.Tensor, gt: torch.Tensor) -> torch.Tensor:
    """loss.py:78-107 ``mpjpe_diffusion``: per step, the best single hypothesis' mean error -> (K,)."""
    K, H = pred.shape[1], pred.shape[2]
    err = torch.norm(pred - gt[:, None, None], dim=-1)
    return err.permute(1, 2, 0, 3, 4).reshape(K, H, -1).mean(-1).min(dim=1).values


def jpma_hip(pred: torch.Tensor, traj: torch.Tensor, cam: torch.Tensor, gt_2d: torch.Tensor,
             gt_3d: torch.Tensor = None, zero_root: bool = True, want_errors: bool = False):
    """The fused HIP kernel (include/d3dp_hip.h: d3dp_jpma) for GPU tensors: returns the aggregated poses (B,K,F,J,3),
    the selected hypothesis index (B,K,F,J) and, with ``want_errors``, the per-joint J_Agg / J_Best errors."""
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


# ---- reduced hypothesis exchange --------------------------------------------------------------------
def jpma_winners(pred: torch.Tensor, traj: torch.Tensor, cam: torch.Tensor, gt_2d: torch.Tensor, h_offset: int = 0,
                 zero_root: bool = True) -> torch.Tensor:
    """This rank's per-joint winner (B,K,F,J,5) = (2D reprojection error, x, y, z, int32 bits of the global hypothesis
    index).  GPU tensors go through the fused HIP kernel (d3dp_jpma_winners); host tensors (the gloo tests) through the
    torch statement of the same selection."""
    B, K, H, Fr, J, _ = pred.shape
    if pred.is_cuda:
        from . import _lib
        lib = _lib.load()
        pred = pred.to(torch.float32).contiguous()
        traj = traj.to(torch.float32).reshape(B, Fr, 3).contiguous()
        cam = cam.to(device=pred.device, dtype=torch.float32).reshape(-1)[:9].contiguous()
        gt_2d = gt_2d.to(torch.float32).contiguous()
        win = torch.empty((B, K, Fr, J, 5), dtype=torch.float32, device=pred.device)
        with torch.cuda.device(pred.device):
            _lib.check(lib.d3dp_jpma_winners(pred.data_ptr(), traj.data_ptr(), cam.data_ptr(), gt_2d.data_ptr(),
                                             win.data_ptr(), int(h_offset), B, K, H, Fr, J, int(zero_root),
                                             _lib.current_stream()), "d3dp_jpma_winners")
        return win
    p = pred.clone().float()
    if zero_root:
        p[:, :, :, :, 0] = 0
    err2d = torch.norm(reproject(p, traj.reshape(B, Fr, 1, 3).float(), cam.float()) - gt_2d[:, None, None].float(), dim=-1)
    e, sel = err2d.min(dim=2, keepdim=True)
    xyz = torch.gather(p, 2, sel[..., None].expand(-1, -1, -1, -1, -1, 3))[:, :, 0]
    hbits = (sel[:, :, 0] + h_offset).to(torch.int32).view(torch.float32)
    return torch.cat((e[:, :, 0, ..., None], xyz, hbits[..., None]), dim=-1).contiguous()


def jpma_combine(win_all: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """win_all (R,B,K,F,J,5), rank-major -> aggregated poses (B,K,F,J,3) and the winning global hypothesis index
    (B,K,F,J) int32: smallest 2D error per joint, lowest rank on ties (= lowest global h, what torch.min over the
    gathered hypothesis axis returns, loss.py:67)."""
    R = win_all.shape[0]
    shp = win_all.shape[1:-1]
    if win_all.is_cuda:
        from . import _lib
        lib = _lib.load()
        w = win_all.to(torch.float32).contiguous()
        n = w[0, ..., 0].numel()
        agg = torch.empty(shp + (3,), dtype=torch.float32, device=w.device)
        sel = torch.empty(shp, dtype=torch.int32, device=w.device)
        with torch.cuda.device(w.device):
            _lib.check(lib.d3dp_jpma_combine(w.data_ptr(), R, n, agg.data_ptr(), sel.data_ptr(), _lib.current_stream()),
                       "d3dp_jpma_combine")
        return agg, sel
    r = win_all[..., 0].min(dim=0, keepdim=True).indices                   # first minimum = lowest rank
    best = torch.gather(win_all, 0, r[..., None].expand(-1, *([-1] * len(shp)), 5))[0]
    return best[..., 1:4].contiguous(), best[..., 4].contiguous().view(torch.int32)


# ---- Protocol #2 (Procrustes-aligned) errors, SURVEY.md §8(f) row N4 --------------------------------------------------
def procrustes_errors(pred: torch.Tensor, gt: torch.Tensor, want_aligned: bool = False):
    """pred (B,K,H,F,J,3), gt (B,F,J,3) on the GPU -> per-joint error after the optimal similarity alignment of every
    predicted pose to its target (B,K,H,F,J) (common/loss.py:190-247 for one pose; batched 3x3 SVD in one HIP launch,
    d3dp_procrustes).  The reference runs this through numpy on the host for every batch (loss.py:263-264)."""
    from . import _lib
    if not pred.is_cuda:
        raise _lib.D3DPHipError("procrustes_errors runs on the GPU (tensor on %s); there is no CPU fallback" % pred.device)
    lib = _lib.load()
    B, K, H, Fr, J, _ = pred.shape
    p = pred.to(torch.float32).contiguous()
    g = gt.to(torch.float32).contiguous()
    err = torch.empty((B, K, H, Fr, J), dtype=torch.float32, device=p.device)
    al = torch.empty_like(p) if want_aligned else None
    with torch.cuda.device(p.device):
        _lib.check(lib.d3dp_procrustes(p.data_ptr(), g.data_ptr(), err.data_ptr(), _lib.ptr(al), B, K * H, Fr, J,
                                       _lib.current_stream()), "d3dp_procrustes")
    return (err, al) if want_aligned else err


def p_mpjpe_metrics(pred: torch.Tensor, gt: torch.Tensor, reproj: torch.Tensor, gt_2d: torch.Tensor) -> Dict[str, torch.Tensor]:
    """The four Protocol #2 errors main.py:726-729 logs per step (K,), in metres:
    J_Best loss.py:p_mpjpe_diffusion_all_min, P_Best p_mpjpe_diffusion, P_Agg ..._all_min(mean_pos=True),
    J_Agg p_mpjpe_diffusion_reproj."""
    K, H = pred.shape[1], pred.shape[2]
    err = procrustes_errors(pred, gt)                                                       # (B,K,H,F,J)
    j_best = err.min(dim=2).values.permute(1, 0, 2, 3).reshape(K, -1).mean(-1)
    p_best = err.permute(1, 2, 0, 3, 4).reshape(K, H, -1).mean(-1).min(dim=1).values
    p_agg = procrustes_errors(pred.mean(dim=2, keepdim=True), gt)[:, :, 0].permute(1, 0, 2, 3).reshape(K, -1).mean(-1)
    sel = torch.norm(reproj - gt_2d[:, None, None], dim=-1).min(dim=2, keepdim=True).indices
    j_agg = torch.gather(err, 2, sel).permute(1, 2, 0, 3, 4).reshape(K, -1).mean(-1)
    return {"J_Best": j_best, "P_Best": p_best, "P_Agg": p_agg, "J_Agg": j_agg}
