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
from d3dp_amd.dist import (all_gather_hypotheses, all_gather_raw, gathered_view, hypothesis_slice, jpma_allgather, jpma_sharded,
                           shard_noise)


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
        # reduced exchange: local winners -> all-gather of 5 floats per joint -> combine must pick
        # the same hypothesis and pose as JPMA over the gathered stack (real projection, root zeroing, ties at joint 0)
        cam = torch.tensor([2.29, 2.287, 0.0254, 0.0289, -0.2070, 0.2477, -0.0030, -0.0009, -0.0014])
        traj = torch.randn(B, Fr, 1, 3, generator=torch.Generator().manual_seed(5)) * 0.1 + torch.tensor([0.0, 0.0, 4.0])
        agg_r, sel_r = jpma_sharded(local, traj, cam, gt2)
        z = full.clone()
        z[:, :, :, :, 0] = 0
        rp_full = jpma.reproject(z, traj, cam)
        sel_ref = torch.norm(rp_full - gt2[:, None, None], dim=-1).min(dim=2).indices
        ok = ok and torch.equal(agg_r, jpma.jpma_aggregate(z, rp_full, gt2)) and torch.equal(sel_r.long(), sel_ref)
        ok = ok and bool((sel_r[..., 0] == 0).all())          # zeroed root: all hypotheses tie -> global h = 0
        # the all-gather result as RCCL leaves it, (R,B,K,H_local,...): the axis-ordered VIEW of it equals the flat tensor,
        # and JPMA on it (north_star's "all-gather before JPMA", no permute copy) == the reduced exchange == the reference order
        raw = all_gather_raw(local)
        ok = ok and raw.shape == (world, B, K, H // world, Fr, 17, 3) and raw.is_contiguous()
        ok = ok and torch.equal(gathered_view(raw).reshape(full.shape), full)
        agg_g, sel_g = jpma_allgather(local, traj, cam, gt2)
        ok = ok and torch.equal(agg_g, agg_r) and torch.equal(sel_g, sel_r)
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


def test_bench_dist_dry_run_8_ranks():
    """`python bench.py --dist-dry-run 8`: the N-rank control flow of bench.py (self-launch under torch.distributed.run,
    WORLD_SIZE check, shard check, timed loop with the all-gather inside, MAX-reduce of the time, multi_gpu block) at the
    world size of BASELINE configs[3], over gloo with a CPU stand-in for the sampler.  What stays untested without an
    8-GPU node is the RCCL transport itself."""
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--dist-dry-run", "8", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=repo)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["dry_run"] is True and d["n_gpus"] == 8 and d["value"] is None
    mg = d["multi_gpu"]
    assert mg["world_size"] == 8 and mg["sharded_equals_single_rank"] is True and mg["all_gather_bytes_per_rank"] > 0
    # the consumer is inside the N-rank timed region, and both exchange forms are reported side by side
    assert "JPMA" in mg["timed_region"] and mg["both_select_the_same_poses"] is True
    full, red = mg["allgather_then_jpma"], mg["winners_exchange"]
    assert full["ms"] > 0 and red["ms"] > 0 and full["bytes_per_rank"] == mg["all_gather_bytes_per_rank"]
    assert abs(red["traffic_ratio"] - 2 * 3 / 5) < 1e-9          # H_local = 2 in the dry run: 2 poses of 3 floats vs 5 floats
    assert d["config"]["parallelism"] == "hshard8"
    # a job whose world size disagrees with --gpus must refuse to run
    env2 = dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    r2 = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--dist-dry-run", "4"], capture_output=True,
                        text=True, timeout=120, env=env2, cwd=repo)
    assert r2.returncode != 0 and "WORLD_SIZE=2" in (r2.stderr + r2.stdout)


def _forced_worker(rank, port, out):
    """One rank, collectives forced (d3dp_amd.dist.FORCE_COLLECTIVES): every exchange of the module is issued on a group of one."""
    import d3dp_amd.dist as dd
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    try:
        assert dd.init_from_env("gloo", force=True) == (0, 1, 0) and dist.is_initialized() and dd.FORCE_COLLECTIVES
        B, K, Hl, Fr = 2, 2, 3, 5
        g = torch.Generator().manual_seed(5)
        local = torch.randn(B, K, Hl, Fr, 17, 3, generator=g)
        traj = torch.randn(B, Fr, 1, 3, generator=g) * 0.1 + torch.tensor([0.0, 0.0, 4.0])
        cam = torch.tensor([1.1, 1.1, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0])
        gt2 = torch.randn(B, Fr, 17, 2, generator=g) * 0.2
        raw = all_gather_raw(local)
        ok = raw.shape == (1,) + tuple(local.shape) and torch.equal(raw[0], local) and raw.data_ptr() != local.data_ptr()
        ok = ok and torch.equal(all_gather_hypotheses(local), local)
        agg_r, sel_r = jpma_sharded(local, traj, cam, gt2)            # (winners all-gathered over the one-rank group)
        agg_g, sel_g = jpma_allgather(local, traj, cam, gt2)
        ok = ok and torch.equal(agg_g, agg_r) and torch.equal(sel_g, sel_r)
        out[0] = bool(ok)
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_forced_collectives_on_a_group_of_one_rank():
    """bench.py --force-exchange (the only way the RCCL side of the N-rank path executes on a one-GPU box): with
    FORCE_COLLECTIVES set a group of ONE rank still issues the all-gathers -- here over gloo -- and both exchange forms select
    the same poses; without the flag a group-less process takes none (test_all_gather_is_identity_without_process_group)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_forced_worker, args=(port, out), nprocs=1, join=True)
    assert dict(out) == {0: True}
