"""The training step (q_sample + MixSTE2 forward / backward + MPJPE loss, no optimizer) against the batch size and the clip length:
usage: train_shapes.py "B,F B,F ..."      (cs = 512, dep = 8; 10 steps after 2 warm-ups each)"""
import os
import sys
import time
from types import SimpleNamespace

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from d3dp_amd import D3DP  # noqa: E402
from d3dp_amd.weights import H36M_JOINTS_LEFT as KL, H36M_JOINTS_RIGHT as KR, make_state_dict  # noqa: E402
import bench  # noqa: E402

for spec in (sys.argv[1] if len(sys.argv) > 1 else "1,243 2,243 4,243 8,243 4,351 2,513").split():
    B, F = (int(v) for v in spec.split(","))
    args = SimpleNamespace(number_of_frames=F, test_time_augmentation=True, timestep=1000, scale=1.0, cs=512, dep=8)
    m = D3DP(args, KL, KR, is_train=True)
    m.load_state_dict(make_state_dict(7, 512, 8, F), strict=False)
    m = m.cuda().train()
    x2 = torch.rand(B, F, 17, 2, device="cuda") * 2 - 1
    x3 = torch.randn(B, F, 17, 3, device="cuda") * 0.3

    def step():
        m.zero_grad(set_to_none=True)
        pr = m(x2, x3)
        loss = torch.mean(torch.norm(pr - x3, dim=-1))
        loss.backward(loss.clone().detach())

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 10 * 1e3
    tf = 3 * B * bench.flops_per_denoiser_call(frames=F) / 1e12
    print(f"B = {B} F = {F}: {ms:.2f} ms per step = {tf / ms * 1e3:.1f} TFLOP/s algorithmic, {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB peak")
    del m
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats()
