#!/usr/bin/env python3
"""Does v_mfma_f32_16x16x32_f16 keep fp16 SUBNORMAL inputs?  (Run on the GPU box.)  Builds operand planes by hand --
d3dp_op_split2 never emits a subnormal hi -- and multiplies them through the EXACT-mode Linear (d3dp_op_linear mode 3)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from d3dp_amd import _lib  # noqa: E402

lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
M, N, K = 128, 128, 32
A2 = torch.zeros(2, M, K, dtype=torch.float16, device="cuda")
W2 = torch.zeros(2, N, K, dtype=torch.float16, device="cuda")
A2[0] = 3.0e-5            # fp16 subnormal (min normal 6.1e-5)
W2[0] = 1.0
b = torch.zeros(N, device="cuda")
out = torch.empty(M, N, device="cuda")
_lib.check(lib.d3dp_op_linear_x2(0, A2.data_ptr(), W2.data_ptr(), b.data_ptr(), 1.0, out.data_ptr(), M, N, K, st))
torch.cuda.synchronize()
want = float(A2[0, 0, 0].float()) * K / 16.0          # the op divides by the activation scale
print(f"subnormal A.hi x 1.0: got {out[0, 0].item():.6e}, exact {want:.6e} -> fp16 subnormal inputs are "
      f"{'PRESERVED' if abs(out[0, 0].item() - want) < 1e-9 else 'FLUSHED'} by the fp16 MFMA")
