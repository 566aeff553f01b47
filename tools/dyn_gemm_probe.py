#!/usr/bin/env python3
"""The training step's Linear alone (test hook d3dp_debug_train_linear) at one shape, N times -- run under
`rocprofv3 --kernel-trace --stats` to read gemm_f16x2_dyn_kernel's duration at that shape (tools/dyn_gemm_probe.sh sweeps the
shapes that separate the per-launch, per-round and per-k-step costs).  usage: dyn_gemm_probe.py M N K [iters]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from d3dp_amd import _lib  # noqa: E402

M, N, K = (int(v) for v in sys.argv[1:4])
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 12
lib = _lib.load()
fn = lib.d3dp_debug_train_linear
fn.restype, fn.argtypes = C.c_int, [C.c_void_p] * 4 + [C.c_int32] * 4 + [C.c_void_p, C.c_int32, C.c_void_p]
A = torch.randn(M, K, device="cuda")
W = torch.randn(N, K, device="cuda") / K ** 0.5
b = torch.randn(N, device="cuda")
out = torch.empty(M, N, device="cuda")
amax = torch.zeros(1, dtype=torch.int32, device="cuda")
for _ in range(iters):
    _lib.check(fn(A.data_ptr(), W.data_ptr(), b.data_ptr(), out.data_ptr(), M, N, K, 1, amax.data_ptr(), 0,
                  torch.cuda.current_stream().cuda_stream), "d3dp_debug_train_linear")
torch.cuda.synchronize()
