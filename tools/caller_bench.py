"""Times the caller-side kernels at the sizes BASELINE configs[2] / configs[4] produce and
prints one JSON object: per kernel the launch time, the algorithmic HBM bytes and the achieved GB/s (peak ~8000)."""
import json
import sys
from types import SimpleNamespace

import numpy as np
import torch

sys.path.insert(0, ".")
from d3dp_amd import D3DP, eval3dhp, jpma  # noqa: E402
from d3dp_amd.clips import clip_gather, clip_scatter  # noqa: E402
from d3dp_amd.data import ChunkedBatcher  # noqa: E402
from d3dp_amd.optim import HipAdamW  # noqa: E402
from d3dp_amd.weights import H36M_JOINTS_LEFT as KL, H36M_JOINTS_RIGHT as KR, make_state_dict  # noqa: E402


def timed(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    dev = torch.device("cuda")
    out = {}
    B, K, H, F, J = 16, 10, 20, 243, 17
    pred = torch.randn(B, K, H, F, J, 3, device=dev) * 0.3
    gt = torch.randn(B, F, J, 3, device=dev) * 0.3
    traj = torch.randn(B, F, 1, 3, device=dev) * 0.1 + torch.tensor([0.0, 0.0, 4.0], device=dev)
    cam = torch.tensor([2.29, 2.287, 0.0254, 0.0289, -0.2070, 0.2477, -0.0030, -0.0009, -0.0014], device=dev)
    gt2 = torch.rand(B, F, J, 2, device=dev) * 2 - 1
    nb = pred.numel() * 4

    def rec(name, ms, nbytes):
        out[name] = {"ms": round(ms, 4), "MB": round(nbytes / 1e6, 1), "GBps": round(nbytes / ms / 1e6, 1)}

    rec("jpma (agg+sel+errors)", timed(lambda: jpma.jpma_hip(pred, traj, cam, gt2, gt, want_errors=True)), nb + B * K * F * J * 24)
    rec("jpma_winners", timed(lambda: jpma.jpma_winners(pred, traj, cam, gt2)), nb + B * K * F * J * 20)
    win = torch.stack([jpma.jpma_winners(pred, traj, cam, gt2, h_offset=r * H) for r in range(8)])
    rec("jpma_combine (8 ranks)", timed(lambda: jpma.jpma_combine(win)), win.numel() * 4 + B * K * F * J * 16)
    rec("jpma_ex (3DHP poses)", timed(lambda: eval3dhp.aggregate_poses(pred, gt, traj, cam, gt2, True, 14)), None or nb * 2 + B * K * F * J * 36)
    rec("procrustes (777.6k poses)", timed(lambda: jpma.procrustes_errors(pred, gt)), nb + pred.numel() // 3 * 4)
    seq = torch.randn(100000, J, 2, device=dev)
    rec("clip_gather+flip (100k frames)", timed(lambda: clip_gather(seq, F, KL, KR)), seq.numel() * 4 * 3)
    pc = torch.randn(412, 1, 1, F, J, 3, device=dev)
    rec("clip_scatter (100k frames)", timed(lambda: clip_scatter(pc, 100000)), 100000 * J * 3 * 4 * 2)
    rng = np.random.default_rng(0)
    lens = [3000] * 40
    bt = ChunkedBatcher(4, None, [rng.standard_normal((n, J, 3)).astype(np.float32) for n in lens],
                        [rng.standard_normal((n, J, 2)).astype(np.float32) for n in lens], F, augment=True, kps_left=KL,
                        kps_right=KR, joints_left=KL, joints_right=KR, device=dev)
    _, pairs = bt.next_pairs()
    tab = bt._tables(pairs)[:4].contiguous()
    rec("batch_gather (B=4)", timed(lambda: bt.gather(tab)), 4 * F * J * 5 * 4 * 2)
    # AdamW over the 34.7 M parameters of the F=243 model
    args = SimpleNamespace(number_of_frames=F, test_time_augmentation=True, timestep=1000, scale=1.0, cs=512, dep=8)
    m = D3DP(args, KL, KR, is_train=True)
    m.load_state_dict(make_state_dict(7, 512, 8, F), strict=False)
    m = m.cuda().train()
    ps = list(m.parameters())
    for p in ps:
        p.grad = torch.randn_like(p) * 1e-3
    opt = HipAdamW(ps, lr=6e-5, weight_decay=0.1)
    n_par = sum(p.numel() for p in ps)
    rec("adamw (34.7M params, 1 launch)", timed(lambda: opt.step()), n_par * 28)
    ref = torch.optim.AdamW(ps, lr=6e-5, weight_decay=0.1)
    rec("torch.optim.AdamW (same, for scale)", timed(lambda: ref.step(), reps=5, warm=2), n_par * 28)
    # one full training step of BASELINE configs[4] (B=4, F=243): batch gather + q_sample + fwd + bwd + AdamW
    x2 = torch.rand(4, F, J, 2, device=dev) * 2 - 1
    x3 = torch.randn(4, F, J, 3, device=dev) * 0.3

    def step():
        opt.zero_grad()
        pr = m(x2, x3)
        loss = torch.mean(torch.norm(pr - x3, dim=-1))
        loss.backward(loss.clone().detach())
        opt.step()
    ms = timed(step, reps=5, warm=2)
    out["train step c5 (B=4,F=243) incl. AdamW"] = {"ms": round(ms, 2), "TFLOPs": round(3 * 4 * 294.86e9 / ms / 1e9, 1)}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
