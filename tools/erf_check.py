"""CPU check of the rational erf used by the EXACT fc1 epilogue (common.h gelu_erf_rational) against fp64:
error of erf on [-6, 6] and of the GELU on N(0,1) inputs, beside torch's own fp32 GELU.   python tools/erf_check.py"""
import numpy as np
import torch
from scipy.special import erf

f = np.float32
A = [f(-2.72614225801306e-10), f(2.77068142495902e-08), f(-2.10102402082508e-06), f(-5.69250639462346e-05),
     f(-7.34990630326855e-04), f(-2.95459980854025e-03), f(-1.60960333262415e-02)]
B = [f(-1.45660718464996e-05), f(-2.13374055278905e-04), f(-1.68282697438203e-03), f(-7.37332916720468e-03),
     f(-1.42647390514189e-02)]


def fma(x, y, z):
    return (x.astype(np.float64) * y.astype(np.float64) + z.astype(np.float64)).astype(np.float32)


def erf_rational(x):
    x = np.clip(x, f(-4), f(4))
    x2 = x * x
    p = np.full_like(x, A[0])
    for c in A[1:]:
        p = fma(p, x2, np.full_like(x, c))
    p = p * x
    q = np.full_like(x, B[0])
    for c in B[1:]:
        q = fma(q, x2, np.full_like(x, c))
    return (p / q).astype(np.float32)


x = np.linspace(-6, 6, 4000001).astype(np.float32)
err = np.abs(erf_rational(x) - erf(x.astype(np.float64)))
print(f"erf on [-6, 6]: max |err| {err.max():.3e} at x = {x[err.argmax()]:.3f}, mean {err.mean():.3e}")
xs = np.random.default_rng(0).standard_normal(2000000).astype(np.float32)
g = 0.5 * xs.astype(np.float64) * (1 + erf(xs.astype(np.float64) / np.sqrt(2)))
ge = (f(0.5) * xs * (f(1) + erf_rational((xs * f(0.70710678)).astype(np.float32)))).astype(np.float64)
tg = torch.nn.functional.gelu(torch.from_numpy(xs)).numpy().astype(np.float64)
print(f"GELU on N(0,1): rational mean |err| {np.abs(ge - g).mean():.3e} max {np.abs(ge - g).max():.3e};  "
      f"torch fp32 mean {np.abs(tg - g).mean():.3e} max {np.abs(tg - g).max():.3e}")
