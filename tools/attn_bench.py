#!/usr/bin/env python3
"""Micro-benchmark of the attention kernels through the C ABI (d3dp_op_attention) at the denoiser's shapes."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from d3dp_amd import _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seqs", type=int, default=15)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--which", default="temporal,spatial")
    ap.add_argument("--joints", type=int, default=17, help="J (temporal sequences gather rows at a stride of J token rows)")
    ap.add_argument("--exact", action="store_true", help="the EXACT-mode (split-fp16) kernels on fp32 rows (impl 2) instead of the bf16 ones")
    a = ap.parse_args()
    lib = _lib.load()
    F, J, C, heads = 243, a.joints, 512, 8
    T = a.seqs * F * J
    qkv = torch.randn(T, 3 * C, device="cuda")
    out = torch.empty(T, C, device="cuda")
    if not a.exact:
        qkv, out = qkv.to(torch.bfloat16), out.to(torch.bfloat16)
    act, impl = (0, 2) if a.exact else (1, 1)
    st = torch.cuda.current_stream().cuda_stream
    for name in a.which.split(","):
        axis = 1 if name == "temporal" else 0
        for _ in range(2):
            _lib.check(lib.d3dp_op_attention(act, impl, axis, qkv.data_ptr(), out.data_ptr(), a.seqs, F, J, C, heads, st))
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.iters)]
        for e0, e1 in evs:
            e0.record()
            _lib.check(lib.d3dp_op_attention(act, impl, axis, qkv.data_ptr(), out.data_ptr(), a.seqs, F, J, C, heads, st))
            e1.record()
        torch.cuda.synchronize()
        ts = sorted(e0.elapsed_time(e1) for e0, e1 in evs)
        n = F if axis else J
        flops = 4.0 * n * C * T
        print(f"{name:9s} T={T}: median {ts[len(ts)//2]*1e3:7.1f} us  min {ts[0]*1e3:7.1f} us  {flops/ts[len(ts)//2]/1e9:7.1f} TFLOP/s  "
              f"{T*4*C*2/ts[len(ts)//2]/1e9*1e-3:6.2f} TB/s")


if __name__ == "__main__":
    main()
