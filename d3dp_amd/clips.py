"""Clip chunking / de-chunking on the device.

The sampler consumes fixed-length clips of F frames; videos have any length.  The reference cuts and re-assembles
them with Python loops over CPU tensors (main.py:267-299 ``eval_data_prepare``; in_the_wild/utils.py:199-240;
in_the_wild/videopose_diffusion.py:150-164 for the way back).  Here both directions are one HIP launch each
(include/d3dp_hip.h: d3dp_clip_gather / d3dp_clip_scatter), and the flipped test-time-augmentation input
(main.py:646-648) comes out of the same pass as the clips.

Clip layout (identical to the reference):  n >= F: clips 0..n//F-1 are consecutive, a trailing partial clip is the
LAST F frames;  n < F: one clip, the sequence replicate-padded on the right.
De-chunking keeps the reference's behaviour for n < F as well: it takes the last n frames of the padded clip
(videopose_diffusion.py:156-159 with batch_num = 1).
"""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import torch

from . import _lib


def clip_count(n_frames: int, receptive_field: int) -> int:
    return 1 if n_frames <= receptive_field else n_frames // receptive_field + (1 if n_frames % receptive_field else 0)


def flip_perm(left: Sequence[int], right: Sequence[int], n_joints: int = 17) -> list:
    """perm[j] = source joint of joint j under ``x[:, left + right] = x[:, right + left]``."""
    perm = list(range(n_joints))
    for dst, src in zip(list(left) + list(right), list(right) + list(left)):
        perm[dst] = src
    return perm


def clip_gather(seq: torch.Tensor, receptive_field: int, kps_left: Optional[Sequence[int]] = None,
                kps_right: Optional[Sequence[int]] = None) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """seq (n,J,D) or (1,n,J,D) on the GPU -> clips (n_clips,F,J,D) and, when the left/right keypoint lists are
    given, the flipped clips (x negated, left/right swapped) the sampler takes as ``input_2d_flip``."""
    if not seq.is_cuda:
        raise _lib.D3DPHipError("clip_gather runs on the GPU (tensor on %s); there is no CPU fallback" % seq.device)
    lib = _lib.load()
    src = seq.reshape(-1, *seq.shape[-2:]).to(torch.float32).contiguous()
    n, J, D = src.shape
    nc = clip_count(n, receptive_field)
    dst = torch.empty((nc, receptive_field, J, D), dtype=torch.float32, device=src.device)
    flip = perm = None
    if kps_left is not None:
        perm = torch.tensor(flip_perm(kps_left, kps_right, J), dtype=torch.int32, device=src.device)
        flip = torch.empty_like(dst)
    with torch.cuda.device(src.device):
        _lib.check(lib.d3dp_clip_gather(src.data_ptr(), dst.data_ptr(), _lib.ptr(flip), _lib.ptr(perm), n,
                                        receptive_field, J, D, _lib.current_stream()), "d3dp_clip_gather")
    return dst, flip


def clip_scatter(pred: torch.Tensor, n_frames: int, last_wins: bool = False) -> torch.Tensor:
    """pred (n_clips,K,H,F,J,D) on the GPU -> (K,H,n_frames,J,D): the per-video prediction the reference assembles at
    videopose_diffusion.py:150-164; ``last_wins`` selects main_3dhp.py:327-330's variant, where the final clip
    overwrites all of the last F frames."""
    if not pred.is_cuda:
        raise _lib.D3DPHipError("clip_scatter runs on the GPU (tensor on %s); there is no CPU fallback" % pred.device)
    lib = _lib.load()
    nc, K, H, Fr, J, D = pred.shape
    assert nc == clip_count(n_frames, Fr), (nc, n_frames, Fr)
    src = pred.to(torch.float32).contiguous()
    out = torch.empty((K, H, n_frames, J, D), dtype=torch.float32, device=src.device)
    with torch.cuda.device(src.device):
        _lib.check(lib.d3dp_clip_scatter(src.data_ptr(), out.data_ptr(), n_frames, K, H, Fr, J, D, int(last_wins),
                                         _lib.current_stream()), "d3dp_clip_scatter")
    return out
