"""Multi-GPU layout of the hot path: one process per GPU, hypotheses sharded over ranks.

Hypotheses are independent given the 2D input (reference common/mixste.py:227-230 folds H into the batch axis
and nothing in the sampler mixes across h), so rank r of N samples hypotheses [r*H_local, (r+1)*H_local) of the
SAME clips with zero communication during the K steps; the one exchange is an all-gather of the per-rank
``x_start`` stacks along the hypothesis axis before JPMA aggregation.  This replaces (does not
port) the reference's nn.DataParallel batch split with its per-call weight broadcast (main.py:242-248).

Backend: ``nccl`` (= RCCL over xGMI on ROCm) on GPUs, ``gloo`` on CPU for the world_size-2 tests.
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None) -> tuple:
    """Initialise torch.distributed from RANK/WORLD_SIZE/MASTER_* (set by torch.distributed.run).
    Returns (rank, world_size, local_rank).  No-op for a single process."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local


def hypothesis_slice(h_total: int, rank: int, world: int) -> slice:
    """Contiguous slice of the global hypothesis axis owned by ``rank`` (h_total must divide evenly)."""
    if h_total % world:
        raise ValueError(f"H_total={h_total} is not divisible by world_size={world}")
    h_local = h_total // world
    return slice(rank * h_local, (rank + 1) * h_local)


def shard_noise(noise: Optional[Sequence[torch.Tensor]], rank: int, world: int) -> Optional[List[torch.Tensor]]:
    """Per-rank slice of GLOBAL noise draws (B, H_total, F, J, 3): with it an N-rank run reproduces the 1-rank
    run with H_total hypotheses bit-for-bit.  Production runs pass ``None`` and seed a per-rank generator."""
    if noise is None:
        return None
    sl = hypothesis_slice(noise[0].shape[1], rank, world)
    return [n[:, sl].contiguous() for n in noise]


def all_gather_hypotheses(preds_local: torch.Tensor, group=None) -> torch.Tensor:
    """(B, K, H_local, F, J, 3) on every rank -> (B, K, H_total, F, J, 3) on every rank, rank-major along H.
    One ncclAllGather (RCCL) per batch; a no-op without an initialised process group."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return preds_local
    world = dist.get_world_size(group)
    B, K, Hl = preds_local.shape[:3]
    src = preds_local.contiguous()
    gathered = torch.empty((world * B,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    dist.all_gather_into_tensor(gathered, src, group=group)      # rank-major concatenation along dim 0
    # (world, B, K, Hl, ...) -> (B, K, world*Hl, ...)
    return gathered.view(world, B, K, Hl, *src.shape[3:]).permute(1, 2, 0, 3, 4, 5, 6).reshape(
        B, K, world * Hl, *src.shape[3:])


def rank_generator(seed: int, rank: int, device) -> torch.Generator:
    g = torch.Generator(device=device)
    g.manual_seed(seed * 1000003 + rank)
    return g


def jpma_sharded(preds_local: torch.Tensor, traj: torch.Tensor, cam: torch.Tensor, gt_2d: torch.Tensor,
                 zero_root: bool = True, group=None):
    """JPMA over hypotheses sharded across ranks with the REDUCED exchange of SURVEY.md §8 E1: every rank selects its
    best hypothesis per (clip, step, frame, joint) locally, the ranks all-gather 5 floats per joint instead of
    H_local poses, and each rank combines the winners.  Returns the same (aggregated poses (B,K,F,J,3), global
    hypothesis index (B,K,F,J)) as JPMA over the all-gathered (B,K,H_total,F,J,3) tensor, with
    H_local/ (5/3) = 12x less traffic at H_local = 20."""
    from .jpma import jpma_combine, jpma_winners
    world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
    rank = dist.get_rank(group) if world > 1 else 0
    Hl = preds_local.shape[2]
    win = jpma_winners(preds_local, traj, cam, gt_2d, h_offset=rank * Hl, zero_root=zero_root)
    if world == 1:
        return jpma_combine(win[None])
    gathered = torch.empty((world * win.shape[0],) + tuple(win.shape[1:]), dtype=win.dtype, device=win.device)
    dist.all_gather_into_tensor(gathered, win.contiguous(), group=group)
    return jpma_combine(gathered.view(world, *win.shape))
