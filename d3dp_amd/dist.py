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

# ROCr reads this flag once, at hsa_init (the process's first HIP call): it has to be in the environment BEFORE any
# torch.cuda call, so it is exported when this module (or the package) is imported, not inside init_from_env.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch                          # noqa: E402
import torch.distributed as dist      # noqa: E402


RCCL_ENV = ("HSA_ENABLE_IPC_MODE_LEGACY", "NCCL_DEBUG", "NCCL_SOCKET_IFNAME", "NCCL_IB_DISABLE", "RCCL_MSCCL_ENABLE",
            "HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "MASTER_ADDR", "MASTER_PORT", "RANK", "LOCAL_RANK", "WORLD_SIZE")


# A process group of ONE rank normally takes none of the collectives below (there is nothing to exchange).  With this flag set
# (bench.py --force-exchange, init_from_env(force=True)) they are issued all the same: on a box with one GPU that is the only way
# the RCCL side of the N-rank path -- rendezvous, communicator, all_gather_into_tensor into the rank-major buffer, the library's
# kernels ordered behind it on torch's stream -- executes on hardware at all.  It proves the stack, not xGMI bandwidth.
FORCE_COLLECTIVES = False


def _exchanges(group=None) -> bool:
    """Whether the collectives of this module are issued: an initialised process group of more than one rank (or forced)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size(group) > 1 or FORCE_COLLECTIVES


def init_from_env(backend: Optional[str] = None, timeout_s: float = 180.0, force: bool = False) -> tuple:
    """Initialise torch.distributed from RANK/WORLD_SIZE/MASTER_* (set by torch.distributed.run).
    Returns (rank, world_size, local_rank).  No-op for a single process unless ``force`` (then a one-rank group is built and
    FORCE_COLLECTIVES set: see above).

    The rendezvous and the first collective are bounded by ``timeout_s`` and a failure names the environment RCCL reads:
    on this ROCm stack cross-process device memory needs dmabuf IPC (``HSA_ENABLE_IPC_MODE_LEGACY=0``, exported when this
    module is IMPORTED if unset -- ROCr reads it at the first HIP call; without it RCCL dies in hipIpcGetMemHandle), and a
    wrong MASTER_ADDR shows up as a silent hang otherwise."""
    import datetime
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if force:
        global FORCE_COLLECTIVES
        FORCE_COLLECTIVES = True
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = dict(rank=rank, world_size=world, timeout=datetime.timedelta(seconds=timeout_s))
        try:
            if backend == "nccl":
                if os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY") != "0":
                    # a deliberate choice on a stack whose legacy IPC works is the user's to make: warn, and let
                    # the probe collective below decide; its failure message names the variable
                    import warnings
                    warnings.warn("d3dp_amd.dist: HSA_ENABLE_IPC_MODE_LEGACY=%r; hosts whose driver only supports dmabuf IPC "
                                  "need 0 (set before the process's first HIP call) or RCCL fails in hipIpcGetMemHandle"
                                  % os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"))
                if torch.cuda.device_count() <= local:
                    raise RuntimeError(f"LOCAL_RANK={local} but only {torch.cuda.device_count()} device(s) visible")
                torch.cuda.set_device(local)
                dist.init_process_group(backend, device_id=torch.device("cuda", local), **kw)
                probe = torch.ones(1, device="cuda")
                dist.all_reduce(probe)                         # the first collective builds the RCCL communicator
                torch.cuda.synchronize()
                if int(probe.item()) != world:
                    raise RuntimeError(f"all_reduce over {world} ranks returned {probe.item()}")
            else:
                dist.init_process_group(backend, **kw)
        except Exception as e:
            env = ", ".join(f"{k}={os.environ[k]}" for k in RCCL_ENV if k in os.environ)
            raise RuntimeError(f"d3dp_amd.dist: rank {rank}/{world} could not join the {backend} process group within "
                               f"{timeout_s:.0f} s: {type(e).__name__}: {e}\n  environment: {env}\n  (RCCL over xGMI needs one "
                               f"visible GPU per rank and HSA_ENABLE_IPC_MODE_LEGACY=0; rendezvous on 127.0.0.1)") from e
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


def all_gather_raw(preds_local: torch.Tensor, group=None) -> torch.Tensor:
    """(B, K, H_local, F, J, 3) on every rank -> (R, B, K, H_local, F, J, 3) on every rank: exactly what ONE ncclAllGather
    (RCCL) leaves, rank-major, not a byte moved afterwards.  Hypothesis h = r H_local + hl.  The consumers below read this
    layout in place (d3dp_jpma_gathered); `gathered_view` presents it with the reference's axis order as a view."""
    src = preds_local.contiguous()
    if not _exchanges(group):
        return src[None]
    world = dist.get_world_size(group)
    gathered = torch.empty((world,) + tuple(src.shape), dtype=src.dtype, device=src.device)
    dist.all_gather_into_tensor(gathered.view(world * src.shape[0], *src.shape[1:]), src, group=group)
    return gathered


def gathered_view(gathered: torch.Tensor) -> torch.Tensor:
    """(R, B, K, H_local, F, J, 3) -> (B, K, R, H_local, F, J, 3): a VIEW with the hypothesis axis split as (rank, local).
    Reductions over hypotheses (min / mean over dims (2, 3)) need no copy; a flat H axis does (`all_gather_hypotheses`)."""
    return gathered.permute(1, 2, 0, 3, 4, 5, 6)


def all_gather_hypotheses(preds_local: torch.Tensor, group=None) -> torch.Tensor:
    """(B, K, H_local, F, J, 3) on every rank -> (B, K, H_total, F, J, 3) on every rank, rank-major along H: the tensor the
    reference's caller would hold (main.py:698).  One all-gather plus ONE COPY (a flat hypothesis axis cannot be a view of
    the rank-major result): 158.6 MB per rank at configs[3] -- use `jpma_allgather`, which consumes the all-gather result
    in place, when JPMA is what follows.  A no-op without an initialised process group."""
    if not _exchanges(group):
        return preds_local
    g = all_gather_raw(preds_local, group)
    R, B, K, Hl = g.shape[:4]
    return gathered_view(g).reshape(B, K, R * Hl, *g.shape[4:])


def jpma_allgather(preds_local: torch.Tensor, traj: torch.Tensor, cam: torch.Tensor, gt_2d: torch.Tensor,
                   zero_root: bool = True, group=None):
    """north_star's exchange: RCCL all-gather of every rank's hypotheses, then JPMA (reference main.py:700-718,
    common/loss.py:54-76) -- with the fused HIP kernel reading the all-gather result where RCCL left it
    (d3dp_jpma_gathered; no permute, no copy).  Returns (aggregated poses (B,K,F,J,3), global hypothesis index (B,K,F,J))
    on every rank, equal bit for bit to `jpma_sharded` and to JPMA over the flat (B,K,H_total,F,J,3) tensor."""
    g = all_gather_raw(preds_local, group)
    R, B, K, Hl, Fr, J, _ = g.shape
    if g.is_cuda:
        from . import _lib
        lib = _lib.load()
        g = g.to(torch.float32)
        traj = traj.to(torch.float32).reshape(B, Fr, 3).contiguous()
        cam = cam.to(device=g.device, dtype=torch.float32).reshape(-1)[:9].contiguous()
        gt_2d = gt_2d.to(torch.float32).contiguous()
        agg = torch.empty((B, K, Fr, J, 3), dtype=torch.float32, device=g.device)
        sel = torch.empty((B, K, Fr, J), dtype=torch.int32, device=g.device)
        with torch.cuda.device(g.device):
            _lib.check(lib.d3dp_jpma_gathered(g.data_ptr(), traj.data_ptr(), cam.data_ptr(), gt_2d.data_ptr(), 0,
                                              agg.data_ptr(), sel.data_ptr(), 0, 0, R, B, K, Hl, Fr, J, int(zero_root),
                                              _lib.current_stream()), "d3dp_jpma_gathered")
        return agg, sel
    # host tensors (the gloo tests and `bench.py --dist-dry-run`): the torch statement of the same selection on the flat
    # layout, one clip at a time (at the BASELINE configs[3] shape the flat tensor is 1.27 GB and the projection's temporaries
    # several times that: eight ranks of it do not fit one host)
    from .jpma import jpma_combine, jpma_winners
    view = gathered_view(g)
    aggs, sels = [], []
    for b in range(B):
        flat = view[b:b + 1].reshape(1, K, R * Hl, Fr, J, 3)
        a, s_ = jpma_combine(jpma_winners(flat, traj[b:b + 1], cam, gt_2d[b:b + 1], h_offset=0, zero_root=zero_root)[None])
        aggs.append(a)
        sels.append(s_)
    return torch.cat(aggs), torch.cat(sels)


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
    exchanges = _exchanges(group)
    world = dist.get_world_size(group) if exchanges else 1
    rank = dist.get_rank(group) if exchanges else 0
    Hl = preds_local.shape[2]
    win = jpma_winners(preds_local, traj, cam, gt_2d, h_offset=rank * Hl, zero_root=zero_root)
    if not exchanges:
        return jpma_combine(win[None])
    gathered = torch.empty((world * win.shape[0],) + tuple(win.shape[1:]), dtype=win.dtype, device=win.device)
    dist.all_gather_into_tensor(gathered, win.contiguous(), group=group)
    return jpma_combine(gathered.view(world, *win.shape))
