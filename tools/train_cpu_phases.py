"""Where the HOST spends a configs[4] training step, and whether the GPU ever waits for it: the step of bench.py's c5 leg
(no optimizer) in a loop WITHOUT synchronising, host timestamps around each phase, device events at the same points.
If the host's time per step is below the device's the launches run ahead and the device never idles between steps.
usage: train_cpu_phases.py [steps]"""
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
names = ["zero_grad", "forward (q_sample, masks, d3dp_train_forward)", "loss", "backward (autograd + d3dp_train_backward)"]
host = [0.0] * len(names)
evs = []
per_step = []


def step(record):
    ts = [time.perf_counter()]
    es = [torch.cuda.Event(enable_timing=True) for _ in range(5)] if record else None
    if record: es[0].record()
    m.zero_grad(set_to_none=True); ts.append(time.perf_counter())
    if record: es[1].record()
    pr = m(x2, x3); ts.append(time.perf_counter())
    if record: es[2].record()
    loss = torch.mean(torch.norm(pr - x3, dim=-1)); ts.append(time.perf_counter())
    if record: es[3].record()
    loss.backward(loss.clone().detach()); ts.append(time.perf_counter())
    if record: es[4].record()
    if record:
        for i in range(len(names)):
            host[i] += ts[i + 1] - ts[i]
        evs.append(es)
        per_step.append([ts[i + 1] - ts[i] for i in range(len(names))])


if os.environ.get("D3DP_SIDE_STREAM"):          # the same loop on a non-default stream (the default one is the legacy NULL stream)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    torch.cuda.set_stream(side)
for _ in range(3):
    step(False)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    step(True)
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"{n} steps: host done enqueueing after {t_host / n * 1e3:.2f} ms per step, device done after {t_all / n * 1e3:.2f} ms per step")
dev = [sum(e[i].elapsed_time(e[i + 1]) for e in evs[2:]) / (n - 2) for i in range(4)]
for i, nm in enumerate(names):
    print(f"  {nm:55s} host {host[i] / n * 1e3:7.3f} ms   device (event to event) {dev[i]:7.3f} ms")
print("  host ms per phase, step by step after the synchronisation (an un-throttled host shows in the first steps):")
for i, ps in enumerate(per_step[:8]):
    print(f"    step {i}: " + "  ".join(f"{v * 1e3:6.3f}" for v in ps))
print(f"  step to step on the device: {sum(a[0].elapsed_time(b[0]) for a, b in zip(evs[2:-1], evs[3:])) / (n - 3):.3f} ms")

if os.environ.get("D3DP_HOST_PROFILE"):
    # the host's own profile of the same loop (cProfile; the ctypes calls into the library show up as built-in calls)
    import cProfile
    import pstats
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(n):
        step(False)
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("cumulative").print_stats(45)
