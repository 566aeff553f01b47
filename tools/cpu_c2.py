#!/usr/bin/env python3
"""The CPU oracle on BASELINE configs[1] at FULL size, un-extrapolated: ddim_sample_flip with F=243, J=17,
H=5, K=5, B=4 (58.97 TFLOP) on the host's cores at the thread count bench.py's sweep found best.  Prints one JSON line.
usage: python tools/cpu_c2.py [threads]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from d3dp_amd.weights import (H36M_JOINTS_LEFT, H36M_JOINTS_RIGHT, flip_2d, make_state_dict, synthetic_inputs_2d,  # noqa: E402
                              synthetic_noise)
from oracle import d3dp_oracle as orc  # noqa: E402

threads = int(sys.argv[1]) if len(sys.argv) > 1 else 16
torch.set_num_threads(threads)
Fr, B, H, K = 243, 4, 5, 5
p = orc.strip_prefix(make_state_dict(7, 512, 8, Fr))
x2d = synthetic_inputs_2d(1301, B, Fr)
nz = [torch.from_numpy(synthetic_noise(1400 + k, (B, H, Fr, 17, 3))) for k in range(K)]
t0 = time.perf_counter()
with torch.no_grad():
    out = orc.ddim_sample_flip(p, orc.cosine_schedule(1000), torch.from_numpy(x2d), torch.from_numpy(flip_2d(x2d)), H, K, 8,
                               H36M_JOINTS_LEFT, H36M_JOINTS_RIGHT, nz)
dt = time.perf_counter() - t0
flop = 294860054528 * 2 * K * B * H
print(json.dumps({"workload": "BASELINE configs[1] full size: F=243 J=17 H=5 K=5 B=4, CPU oracle (port of the reference path)",
                  "threads": threads, "host_cpus": os.cpu_count(), "seconds": dt, "hypothesis_clips_per_s": B * H / dt,
                  "tflop": flop / 1e12, "gflops": flop / dt / 1e9, "finite": bool(torch.isfinite(out).all())}))
