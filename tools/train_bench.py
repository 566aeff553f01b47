"""One training step of BASELINE configs[4] (B=4, F=243, cs=512, dep=8) in a loop, for rocprofv3 --kernel-trace --stats.
usage: train_bench.py [steps [batch]]"""
import sys
from types import SimpleNamespace

import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from d3dp_amd import D3DP, _lib  # noqa: E402
if os.environ.get("D3DP_LIB"):          # A/B of differently compiled libraries
    _lib.LIB_PATH = os.environ["D3DP_LIB"]
from d3dp_amd.optim import HipAdamW  # noqa: E402
from d3dp_amd.weights import H36M_JOINTS_LEFT as KL, H36M_JOINTS_RIGHT as KR, make_state_dict  # noqa: E402

F, J = 243, 17
args = SimpleNamespace(number_of_frames=F, test_time_augmentation=True, timestep=1000, scale=1.0, cs=512, dep=8)
m = D3DP(args, KL, KR, is_train=True)
m.load_state_dict(make_state_dict(7, 512, 8, F), strict=False)
m = m.cuda().train()
opt = HipAdamW(m.parameters(), lr=6e-5, weight_decay=0.1)
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
x2 = torch.rand(B, F, J, 2, device="cuda") * 2 - 1
x3 = torch.randn(B, F, J, 3, device="cuda") * 0.3
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
for i in range(n + 2):
    if i == 2:
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
    opt.zero_grad()
    pr = m(x2, x3)
    loss = torch.mean(torch.norm(pr - x3, dim=-1))
    loss.backward(loss.clone().detach())
    opt.step()
b.record()
torch.cuda.synchronize()
ms = a.elapsed_time(b) / n
print(f"train step (B={B}): {ms:.2f} ms = {3 * B * 294.86e9 / ms / 1e9:.1f} TFLOP/s, {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB peak")
