"""Dump op-level outputs of the library selected by D3DP_LIB for bitwise comparison between builds (debug aid)."""
import os, sys, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from d3dp_amd import _lib
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
g = torch.Generator().manual_seed(3)
M, K, C = 4131, 512, 512
A = (torch.randn(M, K, generator=g) * 2).cuda()
out = {}
A2 = torch.empty(2, M, K, dtype=torch.float16, device="cuda")
_lib.check(lib.d3dp_op_split2(A.data_ptr(), A2.data_ptr(), M * K, 16.0, st))
out["A2"] = A2
for name, epi, N in (("bias", 0, 512), ("gelu", 1, 1024), ("resid", 2, 512), ("qkv", 4, 1536)):
    W = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    b = torch.randn(N, generator=g).cuda()
    W2 = torch.empty(2, N, K, dtype=torch.float16, device="cuda")
    ws = 2.0 ** (13 - int(np.floor(np.log2(W.abs().max().item()))))
    _lib.check(lib.d3dp_op_split2(W.data_ptr(), W2.data_ptr(), N * K, ws, st))
    o = torch.randn(M, N, generator=g).cuda() if epi == 2 else torch.zeros(M * N, device="cuda")
    _lib.check(lib.d3dp_op_linear_x2(epi, A2.data_ptr(), W2.data_ptr(), b.data_ptr(), ws, o.data_ptr(), M, N, K, st))
    out[name] = o
F, J, nbh = 243, 17, 1
qkv = torch.randn(nbh * F * J, 3 * C, generator=g).cuda()
for axis in (0, 1):
    o = torch.zeros(nbh * F * J, C, device="cuda")
    _lib.check(lib.d3dp_op_attention(0, 2, axis, qkv.data_ptr(), o.data_ptr(), nbh, F, J, C, 8, st))
    out[f"attn{axis}"] = o
x = torch.randn(1000, C, generator=g).cuda()
w = torch.randn(C, generator=g).cuda(); bb = torch.randn(C, generator=g).cuda()
o = torch.zeros(2, 1000, C, dtype=torch.float16, device="cuda")
try:
    _lib.check(lib.d3dp_op_layernorm(3, x.data_ptr(), w.data_ptr(), bb.data_ptr(), 1e-6, o.data_ptr(), 1000, C, st))
    out["ln_h2"] = o
except Exception as e:
    print("ln:", e)
torch.cuda.synchronize()
for k, v in out.items():
    print(k, hashlib.sha256(v.cpu().numpy().tobytes()).hexdigest()[:16])
