"""N>1 path on CPU (gloo, world_size 2): hypothesis sharding + the one all-gather before JPMA.
The sampler itself needs the GPU; here a deterministic stand-in produces each rank's (B,K,H_local,F,J,3) stack so
that the layout contract (rank-major along H, identical on every rank, consumer result == single-process result)
is checked end to end through torch.distributed."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from d3dp_amd import jpma
from d3dp_amd.dist import all_gather_hypotheses, hypothesis_slice, shard_noise


def fake_sampler(x2d, noise):
    """Stand-in with the sampler's independence structure: hypothesis h depends only on (x2d, noise[:, h])."""
    K = len(noise)
    return torch.stack([torch.tanh(n * 0.3 + x2d.mean(dim=-1, keepdim=True)[:, None] * (k + 1)) for k, n in enumerate(noise)], dim=1)


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(3)
        B, K, H, Fr = 2, 3, 6, 5
        x2d = torch.randn(B, Fr, 17, 2, generator=g)
        noise = [torch.randn(B, H, Fr, 17, 3, generator=g) for _ in range(K)]
        full = fake_sampler(x2d, noise)                                   # single-process reference (B,K,H,...)
        local = fake_sampler(x2d, shard_noise(noise, rank, world))        # this rank's hypotheses
        assert local.shape[2] == H // world
        gathered = all_gather_hypotheses(local)
        assert gathered.shape == full.shape
        ok = torch.equal(gathered, full)
        # consumer: JPMA aggregation on the gathered stack == on the single-process stack
        gt2 = x2d
        rp = gathered[..., :2] * 0.5
        agg = jpma.jpma_aggregate(gathered, rp, gt2)
        agg_ref = jpma.jpma_aggregate(full, full[..., :2] * 0.5, gt2)
        ok = ok and torch.equal(agg, agg_ref)
        out[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def test_hypothesis_slice():
    assert hypothesis_slice(160, 3, 8) == slice(60, 80)
    with pytest.raises(ValueError):
        hypothesis_slice(10, 0, 3)


def test_all_gather_layout_world2_gloo():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    assert dict(out) == {0: True, 1: True}


def test_all_gather_is_identity_without_process_group():
    x = torch.randn(1, 2, 3, 4, 17, 3)
    assert all_gather_hypotheses(x) is x
