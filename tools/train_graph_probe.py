"""What capturing the configs[4] training step into ONE hipGraph would buy: the step of bench.py's c5 leg (static inputs) replayed
from a torch.cuda.CUDAGraph (on ROCm: hipGraph; the library's launches, its second stream and its events are captured with the rest)
against the same step launched eagerly.  A measurement tool, not the product path.
usage: train_graph_probe.py [steps]"""
import os
import sys
import time
from types import SimpleNamespace

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from d3dp_amd import D3DP, _lib  # noqa: E402
if os.environ.get("D3DP_LIB"):
    _lib.LIB_PATH = os.environ["D3DP_LIB"]
from d3dp_amd.weights import H36M_JOINTS_LEFT as KL, H36M_JOINTS_RIGHT as KR, make_state_dict  # noqa: E402

F, J, B = 243, 17, 4
args = SimpleNamespace(number_of_frames=F, test_time_augmentation=True, timestep=1000, scale=1.0, cs=512, dep=8)
m = D3DP(args, KL, KR, is_train=True)
m.load_state_dict(make_state_dict(7, 512, 8, F), strict=False)
m = m.cuda().train()
x2 = torch.rand(B, F, J, 2, device="cuda") * 2 - 1
x3 = torch.randn(B, F, J, 3, device="cuda") * 0.3
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20


def step():
    m.zero_grad(set_to_none=True)
    pr = m(x2, x3)
    loss = torch.mean(torch.norm(pr - x3, dim=-1))
    loss.backward(loss.clone().detach())
    return loss


def timed(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        step()
torch.cuda.current_stream().wait_stream(side)
eager = [timed(step) for _ in range(2)]
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    loss = step()
replay = [timed(g.replay) for _ in range(3)]
eager2 = timed(step)
print(f"configs[4] step, eager: {eager[0]:.3f} / {eager[1]:.3f} / {eager2:.3f} ms;  one hipGraph replayed: " + " / ".join(f"{v:.3f}" for v in replay) + " ms")
