"""Whole-video inference around the hot path (the serving flow of the reference's in_the_wild/ scripts, minus the 2D
detector and the renderer): pixel keypoints of any length in, per-frame multi-hypothesis 3D poses out.

    keypoints (n,17,2) pixels --normalize_screen_coordinates (camera.py:7-11)--> [-1,1]
      --flip copy + clip cutting (in_the_wild/utils.py:246-262, 199-240; ONE d3dp_clip_gather launch)-->
      (n_clips,F,17,2) x2 --D3DP.forward in batches of ``batch_clips`` (utils.py:268-283)--> (b,K,H,F,17,3)
      --root joint zeroed (utils.py:284)--> --d3dp_clip_scatter (videopose_diffusion.py:150-164)--> (K,H,n,17,3)

Everything after the upload of the keypoints stays on the GPU.  The reference moves every batch back to the host and
re-assembles the video with numpy (utils.py:292-296, videopose_diffusion.py:146-164).
"""
from __future__ import annotations

from typing import Optional, Sequence

import numpy as np
import torch

from . import _lib
from .clips import clip_gather, clip_scatter

# in_the_wild/utils.py:250-251 (COCO-ordered detector keypoints are flipped with these lists; the model keeps the
# Human3.6M joint lists)
WILD_KPS_LEFT, WILD_KPS_RIGHT = [1, 3, 5, 7, 9, 11, 13, 15], [2, 4, 6, 8, 10, 12, 14, 16]


def normalize_screen_coordinates(x: torch.Tensor, w: float, h: float) -> torch.Tensor:
    """camera.py:7-11: [0, w] -> [-1, 1], aspect ratio preserved."""
    return x / w * 2 - torch.tensor([1.0, h / w], dtype=x.dtype, device=x.device)


@torch.no_grad()
def predict_video(model, keypoints_px, width: int, height: int, batch_clips: int = 2,
                  kps_left: Sequence[int] = WILD_KPS_LEFT, kps_right: Sequence[int] = WILD_KPS_RIGHT,
                  generator: Optional[torch.Generator] = None, noise=None, zero_root: bool = True) -> torch.Tensor:
    """keypoints_px (n,17,2) in pixels (numpy or tensor) -> (K, H, n, 17, 3) fp32 on the model's device.
    ``noise``: optional list (one entry per batch of clips) of per-step noise lists, for reproducible runs."""
    dev = next(model.parameters()).device
    if dev.type != "cuda":
        raise _lib.D3DPHipError("predict_video runs on an MI355X (model on %s); there is no CPU fallback" % dev)
    if torch.is_tensor(keypoints_px):
        kp = normalize_screen_coordinates(keypoints_px[..., :2].to(device=dev, dtype=torch.float32), float(width), float(height))
    else:       # like the reference: float64 numpy arithmetic on the detector output, one rounding to fp32 (utils.py:253)
        kpn = np.asarray(keypoints_px, dtype=np.float64)[..., :2]
        kp = torch.from_numpy((kpn / width * 2 - [1, height / width]).astype(np.float32)).to(dev)
    n = kp.shape[0]
    x2, x2f = clip_gather(kp, model.frames, kps_left, kps_right)
    outs = []
    for bi, i in enumerate(range(0, x2.shape[0], batch_clips)):
        kw = {} if noise is None else {"noise": noise[bi]}
        pred = model(x2[i:i + batch_clips].contiguous(), None, input_2d_flip=x2f[i:i + batch_clips].contiguous(),
                     generator=generator, **kw)
        if zero_root:
            pred[:, :, :, :, 0] = 0
        outs.append(pred)
    return clip_scatter(torch.cat(outs), n)
