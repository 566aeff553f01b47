"""CPU study (authoring container or GPU-box host; needs no GPU): which operand splits keep the EXACT mode
within the 1e-3 mm parity tolerance?

Every big nn.Linear of one denoiser call is replaced by an emulation of a multi-pass MFMA GEMM on split operands:

  bf16x3  : x = x0 + x1 + x2 (bf16 planes), six leading plane pairs            (round-1 EXACT mode)
  f16x2   : x = hi + lo * 2^-11 (fp16 planes, lo pre-scaled by 2^11), passes hi.hi + (hi.lo + lo.hi) * 2^-11
            (first round-2 form: cross terms in their own accumulator)
  f16x2+  : as f16x2 plus the lo.lo pass (four passes)
  f16x2u  : THE SHIPPED FORM (gemm_x2.hip): activations x 16 and weights x 2^s (max |w| 2^s in [2^13, 2^14)) split into
            hi = fp16(y), lo = fp16(y - hi) (unscaled), three passes hi.hi + hi.lo + lo.hi into ONE accumulator

Accumulation is emulated two ways: `acc32` multiplies the planes with torch's fp32 matmul (products of two 11-bit
significands are exact in fp32, accumulation rounds like an fp32 kernel), `acc64` accumulates in fp64 and rounds once
(the representation error alone).  Printed: MPJPE (mm) of the emulated path against the fp32 oracle (the parity
target) and against the fp64 oracle (the truth), for a single denoiser call.

    python tools/err_budget_split.py [F] [t]
"""
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, '.')
from oracle import d3dp_oracle as orc                                    # noqa: E402
from d3dp_amd.weights import make_state_dict, synthetic_inputs_2d, synthetic_noise   # noqa: E402

torch.set_num_threads(8)
Fr = int(sys.argv[1]) if len(sys.argv) > 1 else 27
tval = int(sys.argv[2]) if len(sys.argv) > 2 else 499
cs, dep, B, H = 512, 8, 1, 2
sd = make_state_dict(7, cs, dep, Fr)
p32 = orc.strip_prefix(sd)
p64 = orc.strip_prefix(sd, dtype=torch.float64)
x2 = torch.from_numpy(synthetic_inputs_2d(1, B, Fr))
x3 = torch.from_numpy(synthetic_noise(2, (B, H, Fr, 17, 3)))
t = torch.tensor([tval])
ref32 = orc.mixste_forward(p32, x2, x3, t, dep)
ref64 = orc.mixste_forward(p64, x2.double(), x3.double(), t, dep)
print(f'F={Fr} t={tval}:  fp32 oracle vs fp64 oracle {orc.mpjpe_mm(ref32, ref64):.3e} mm')

orig_linear = F.linear
S = 2048.0


def split_bf16x3(x):
    a = x.to(torch.bfloat16).float()
    r = x - a
    b = r.to(torch.bfloat16).float()
    c = (r - b).to(torch.bfloat16).float()
    return a, b, c


def split_f16x2(x):
    hi = x.to(torch.float16).float()
    lo = ((x - hi) * S).to(torch.float16).float()
    return hi, lo


def split_f16x2u(y):
    hi = y.to(torch.float16).float()
    lo = (y - hi).to(torch.float16).float()
    return hi, lo


def mm(a, w, acc64):
    if acc64:
        return a.double() @ w.double().t()
    return a @ w.t()


def make_linear(kind, acc64):
    def lin(x, w, b=None):
        if not (w.shape[0] >= 64 and w.shape[1] >= 64 and x.dtype == torch.float32 and x.numel() // x.shape[-1] > 64):
            return orig_linear(x, w, b)
        shp = x.shape
        x = x.reshape(-1, shp[-1])
        if kind == 'bf16x3':
            a0, a1, a2 = split_bf16x3(x)
            w0, w1, w2 = split_bf16x3(w)
            y = mm(a2, w0, acc64) + mm(a1, w1, acc64) + mm(a0, w2, acc64) + mm(a1, w0, acc64) + mm(a0, w1, acc64) + mm(a0, w0, acc64)
        elif kind == 'f16x2u':
            ws = 2.0 ** (13 - int(torch.floor(torch.log2(w.abs().max())).item()))
            ah, al = split_f16x2u(x * 16.0)
            wh, wl = split_f16x2u(w * ws)
            y = (mm(ah, wl, acc64) + mm(al, wh, acc64) + mm(ah, wh, acc64)) / (16.0 * ws)
        else:
            ah, al = split_f16x2(x)
            wh, wl = split_f16x2(w)
            cross = mm(ah, wl, acc64) + mm(al, wh, acc64)
            y = mm(ah, wh, acc64) + cross / S
            if kind == 'f16x2+':
                y = y + mm(al, wl, acc64) / (S * S)
        y = y.float() if acc64 else y
        if b is not None:
            y = y + b
        return y.reshape(*shp[:-1], w.shape[0])
    return lin


for kind in ('bf16x3', 'f16x2', 'f16x2+', 'f16x2u'):
    for acc64 in (False, True):
        F.linear = make_linear(kind, acc64)
        try:
            out = orc.mixste_forward(p32, x2, x3, t, dep)
        finally:
            F.linear = orig_linear
        print(f'{kind:8s} {"acc64" if acc64 else "acc32"}: vs fp32 oracle {orc.mpjpe_mm(out, ref32):.3e} mm   '
              f'vs fp64 {orc.mpjpe_mm(out, ref64):.3e} mm')
