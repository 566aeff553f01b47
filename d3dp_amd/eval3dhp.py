"""MPI-INF-3DHP evaluation flow around the hot path (reference main_3dhp.py:659-912).

Differences from the Human3.6M flow (d3dp_amd/cli.py), all taken from the reference:
  * poses in millimetres (model: d3dp_amd.D3DP3DHP), root joint 14 instead of 0 (main_3dhp.py:773-777);
  * per-frame validity mask on the errors (common/loss.py:109-145 ``mpjpe_diffusion_3dhp``);
  * J-Agg reprojects with fixed pixel-unit intrinsics -- linear model for TS1-TS4, distortion model with zero
    coefficients for TS5/TS6 (main_3dhp.py:696-701, 806-814) -- against the 2D input mapped back to pixel
    coordinates (camera.py:14-18, main_3dhp.py:829);
  * the four aggregated POSES (P-Agg, P-Best, J-Best, J-Agg) are materialised, stitched back to sequence length with
    the final clip owning the last F frames (main_3dhp.py:327-331) and exported as MATLAB files (:903-912).

Device work: clip cutting/stitching (d3dp_clip_gather / d3dp_clip_scatter), the sampler, and one fused pass over the
hypotheses (d3dp_jpma_ex) that yields the J-Agg, J-Best and P-Agg poses; P-Best is an argmin over H of a mean that
spans the whole batch and stays a short chain of torch reductions on the device.
"""
from __future__ import annotations

import os
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib
from .clips import clip_gather, clip_scatter

ROOT_3DHP = 14
KPS_LEFT_3DHP, KPS_RIGHT_3DHP = [5, 6, 7, 11, 12, 13], [2, 3, 4, 8, 9, 10]       # main_3dhp.py:117-118
# (intrinsics in mm, [width, height, sensor_x, sensor_y]) -- main_3dhp.py:696-699
_CAM_MM = {1: ([7.32506, 7.32506, -0.0322884, 0.0929296, 0, 0, 0, 0, 0], [2048, 2048, 10, 10]),
           2: ([8.770747185, 8.770747185, -0.104908645, 0.104899704, 0, 0, 0, 0, 0], [1920, 1080, 10, 5.625])}


def cam_mm_to_pix(cam, cam_data) -> torch.Tensor:
    """main_3dhp.py:334-342."""
    cam = torch.as_tensor(cam, dtype=torch.float32).clone()
    mx, my = cam_data[0] / cam_data[2], cam_data[1] / cam_data[3]
    cam[0] = cam[0] * mx
    cam[1] = cam[1] * my
    cam[2] = cam[2] * mx + cam_data[0] / 2
    cam[3] = cam[3] * my + cam_data[1] / 2
    return cam


def camera_for(key: str):
    """(pixel-unit intrinsics (9,), [w,h,..], linear projection?) for a test sequence -- main_3dhp.py:806-814."""
    which = 2 if key in ("TS5", "TS6") else 1
    cam, data = _CAM_MM[which]
    return cam_mm_to_pix(cam, data), data, which == 1


def image_coordinates(x: torch.Tensor, w: float, h: float) -> torch.Tensor:
    """camera.py:14-18: undo the screen normalisation."""
    return (x + torch.tensor([1.0, h / w], dtype=x.dtype, device=x.device)) * w / 2


def mpjpe_diffusion_3dhp(pred: torch.Tensor, target: torch.Tensor, valid: torch.Tensor, mean_pos: bool = False):
    """loss.py:109-145.  pred (B,K,H,F,J,3), target (B,F,J,3), valid (B,F,1) bool -> (K,): over valid frames only, the
    best hypothesis' mean error (or the mean pose's error)."""
    v = valid.reshape(valid.shape[0], valid.shape[1]).bool()
    p = pred.permute(0, 3, 1, 2, 4, 5)[v]                                   # (Nv,K,H,J,3)
    t = target[v]                                                           # (Nv,J,3)
    K, H = p.shape[1], p.shape[2]
    if not mean_pos:
        err = torch.norm(p - t[:, None, None], dim=-1)                      # (Nv,K,H,J)
        return err.permute(1, 2, 0, 3).reshape(K, H, -1).mean(-1).min(dim=1).values
    err = torch.norm(p.mean(dim=2) - t[:, None], dim=-1)                    # (Nv,K,J)
    return err.permute(1, 0, 2).reshape(K, -1).mean(-1)


def aggregate_poses(pred: torch.Tensor, gt: torch.Tensor, traj: torch.Tensor, cam: torch.Tensor, target_2d: torch.Tensor,
                    linear: bool, root_joint: int = -1) -> Dict[str, torch.Tensor]:
    """The four aggregated poses of main_3dhp.py:779-835, each (B,K,F,J,3).  pred (B,K,H,F,J,3) with the root already
    zeroed (or pass root_joint), gt (B,F,J,3), traj (B,F,1,3), cam (9,) pixel units, target_2d (B,F,J,2) pixels."""
    if not pred.is_cuda:
        raise _lib.D3DPHipError("aggregate_poses runs on the GPU (tensor on %s); there is no CPU fallback" % pred.device)
    lib = _lib.load()
    B, K, H, Fr, J, _ = pred.shape
    p = pred.to(torch.float32).contiguous()
    g = gt.to(torch.float32).contiguous()
    tr = traj.to(torch.float32).reshape(B, Fr, 3).contiguous()
    c = cam.to(device=p.device, dtype=torch.float32).reshape(-1)[:9].contiguous()
    t2 = target_2d.to(torch.float32).contiguous()
    out = {k: torch.empty((B, K, Fr, J, 3), dtype=torch.float32, device=p.device) for k in ("J_Agg", "J_Best", "P_Agg")}
    with torch.cuda.device(p.device):
        _lib.check(lib.d3dp_jpma_ex(p.data_ptr(), tr.data_ptr(), c.data_ptr(), t2.data_ptr(), g.data_ptr(),
                                    out["J_Agg"].data_ptr(), 0, 0, 0, out["J_Best"].data_ptr(), out["P_Agg"].data_ptr(),
                                    B, K, H, Fr, J, int(root_joint), int(linear), _lib.current_stream()), "d3dp_jpma_ex")
    # P-Best (main_3dhp.py:782-792): per step the hypothesis with the smallest error averaged over the whole batch
    pz = p
    if root_joint >= 0:
        pz = p.clone()
        pz[:, :, :, :, root_joint] = 0
    err = torch.norm(pz - g[:, None, None], dim=-1)                                          # (B,K,H,F,J)
    h_min = err.permute(1, 2, 0, 3, 4).reshape(K, H, -1).mean(-1).min(dim=1).indices        # (K,)
    out["P_Best"] = pz[:, torch.arange(K, device=p.device), h_min]                           # (B,K,F,J,3)
    return out


@torch.no_grad()
def evaluate_sequence(model, seq3d_mm, seq2d, valid, key: str, receptive_field: int, batch_clips: int = 2,
                      generator: Optional[torch.Generator] = None, noise=None):
    """One test sequence through main_3dhp.py:711-874.  seq3d_mm (n,17,3) camera-space millimetres, seq2d (n,17,2)
    normalised screen coordinates, valid (n,) -> (per-step error sums weighted by clips*F for P_Best and P_Agg, N,
    stitched poses {name: (K,n,17,3)} plus 'all' (K,H,n,17,3))."""
    dev = next(model.parameters()).device
    s3 = torch.as_tensor(np.asarray(seq3d_mm), dtype=torch.float32, device=dev)
    s2 = torch.as_tensor(np.asarray(seq2d), dtype=torch.float32, device=dev)
    vf = torch.as_tensor(np.asarray(valid), dtype=torch.float32, device=dev).reshape(-1, 1, 1)
    n = s3.shape[0]
    x2, x2f = clip_gather(s2, receptive_field, KPS_LEFT_3DHP, KPS_RIGHT_3DHP)
    x3, _ = clip_gather(s3, receptive_field)
    vclip, _ = clip_gather(vf, receptive_field)                           # (nc,F,1,1)
    cam, cam_data, linear = camera_for(key)
    w, h = float(cam_data[0]), float(cam_data[1])
    K = model.sampling_timesteps
    sums = {"P_Best": torch.zeros(K, device=dev), "P_Agg": torch.zeros(K, device=dev)}
    N, parts = 0, {k: [] for k in ("all", "P_Agg", "P_Best", "J_Best", "J_Agg")}
    for bi, i in enumerate(range(0, x3.shape[0], batch_clips)):
        a2, a2f, a3 = x2[i:i + batch_clips].contiguous(), x2f[i:i + batch_clips].contiguous(), x3[i:i + batch_clips].clone()
        av = vclip[i:i + batch_clips, :, 0] > 0.5                          # (b,F,1)
        traj = a3[:, :, ROOT_3DHP:ROOT_3DHP + 1].clone()
        a3[:, :, ROOT_3DHP] = 0
        kw = {} if noise is None else {"noise": noise[bi]}
        pred = model(a2, a3, input_2d_flip=a2f, generator=generator, **kw)      # (b,K,H,F,17,3) mm
        pred[:, :, :, :, ROOT_3DHP] = 0
        poses = aggregate_poses(pred, a3, traj, cam, image_coordinates(a2, w, h), linear)
        parts["all"].append(pred)
        for k in ("P_Agg", "P_Best", "J_Best", "J_Agg"):
            parts[k].append(poses[k])
        wgt = a3.shape[0] * a3.shape[1]
        sums["P_Best"] += wgt * mpjpe_diffusion_3dhp(pred, a3, av)
        sums["P_Agg"] += wgt * mpjpe_diffusion_3dhp(pred, a3, av, mean_pos=True)
        N += wgt
    stitched = {"all": clip_scatter(torch.cat(parts["all"]), n, last_wins=True)}
    for k in ("P_Agg", "P_Best", "J_Best", "J_Agg"):
        stitched[k] = clip_scatter(torch.cat(parts[k])[:, :, None], n, last_wins=True)[:, 0]
    return sums, N, stitched


def export_mat(checkpoint_dir: str, per_sequence: Dict[str, Dict[str, torch.Tensor]]) -> Dict[str, str]:
    """main_3dhp.py:903-912: one .mat per aggregation with {sequence key: array (3,17,n,K)} (the reference's
    ``transpose(3,2,1,0)`` of (K,n,17,3))."""
    import scipy.io as scio
    os.makedirs(checkpoint_dir, exist_ok=True)
    paths = {}
    for name in ("P_Agg", "P_Best", "J_Best", "J_Agg"):
        data = {key: st[name].detach().cpu().numpy().astype(np.float64).transpose(3, 2, 1, 0) for key, st in per_sequence.items()}
        paths[name] = os.path.join(checkpoint_dir, "inference_data_%s.mat" % name)
        scio.savemat(paths[name], data)
    return paths
