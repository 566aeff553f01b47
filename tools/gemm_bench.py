#!/usr/bin/env python3
"""Micro-benchmark of the Linear kernels through the C ABI (d3dp_op_linear) at the denoiser's shapes.
Usage: python tools/gemm_bench.py [--m 61965] [--iters 20] [--tile]   (run on the GPU box; wrap with rocprofv3 for PMC)"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from d3dp_amd import _lib  # noqa: E402

if os.environ.get("D3DP_LIB"):          # A/B of differently compiled libraries
    _lib.LIB_PATH = os.environ["D3DP_LIB"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=15 * 4131)
    ap.add_argument("--iters", type=int, default=20)
    
    ap.add_argument("--shapes", default="qkv,proj,fc1,fc2")
    ap.add_argument("--x3", action="store_true", help="six-pass split-bf16 kernel (three planes per operand)")
    ap.add_argument("--x2", action="store_true", help="EXACT-mode split-fp16 kernel (two planes per operand, three passes)")
    ap.add_argument("--real-epi", action="store_true", help="--x2: the epilogues the denoiser launches (qkv: packed rows, proj / fc2: x += ..., fc1: GELU planes)")
    ap.add_argument("--pp", action="store_true", help="--x2: the ping-pong form of the kernel (epi | 2048)")
    ap.add_argument("--wide", action="store_true", help="--x2: the 256 x 256 tile form of the kernel (epi | 4096)")
    ap.add_argument("--skew", type=int, default=0, help="--x2 --real-epi: D of the skewed schedule for qkv and fc1 (0 = plain)")
    ap.add_argument("--check", action="store_true", help="compare each result with torch (fp32 matmul of the bf16 operands)")
    ap.add_argument("--cache", default="hot", choices=["hot", "cold", "produced"],
                    help="state of A before each timed launch: hot = same buffers back to back; cold = 1 GB written in "
                         "between (evicts L2 + the 256 MB memory-side cache); produced = A rewritten front to back by a "
                         "copy kernel (+ a 2x-sized unrelated read/write, like the row kernels in the denoiser)")
    a = ap.parse_args()
    lib = _lib.load()
    M = a.m
    shapes = {"qkv": (1536, 512, 0), "proj": (512, 512, 0), "fc1": (1024, 512, 1), "fc2": (512, 1024, 0),   # bf16 outputs, as the denoiser launches them
              "cal": (128, 512, 0)}   # ONE column tile: every A element is fetched exactly once (calibrates FETCH_SIZE)
    st = torch.cuda.current_stream().cuda_stream
    if a.x2:
        for name in a.shapes.split(","):
            N, K, epi = shapes[name]
            if a.real_epi:
                epi = {"qkv": 4 | (a.skew << 8), "proj": 2, "fc1": 1 | (a.skew << 8), "fc2": 2}.get(name, epi)
            if a.pp:
                epi |= 2048
            if a.wide:
                epi |= 4096
            A = torch.randn(M, K, device="cuda")
            W = torch.randn(N, K, device="cuda") / K ** 0.5
            A2 = torch.empty(2, M, K, dtype=torch.float16, device="cuda")
            W2 = torch.empty(2, N, K, dtype=torch.float16, device="cuda")
            _lib.check(lib.d3dp_op_split2(A.data_ptr(), A2.data_ptr(), M * K, 16.0, st))
            _lib.check(lib.d3dp_op_split2(W.data_ptr(), W2.data_ptr(), N * K, 65536.0, st))
            b = torch.randn(N, device="cuda")
            out = torch.empty(M * N, device="cuda", dtype=torch.float32)
            ts = []
            for i in range(a.iters + 3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                _lib.check(lib.d3dp_op_linear_x2(epi, A2.data_ptr(), W2.data_ptr(), b.data_ptr(), 65536.0, out.data_ptr(), M, N, K, st))
                e1.record()
                torch.cuda.synchronize()
                if i >= 3:
                    ts.append(e0.elapsed_time(e1))
            ts.sort()
            med = ts[len(ts) // 2]
            if a.check and not a.real_epi:
                ref = A.double() @ W.double().t() + b.double()
                if epi:
                    got = out.view(torch.float16)[:2 * M * N].view(M, N // 32, 2, 32)   # h2i layout: [hi 32 | lo 32] blocks
                    got = (got[:, :, 0].double() + got[:, :, 1].double()).reshape(M, N) / 16.0
                    ref = torch.nn.functional.gelu(ref)
                else:
                    got = out.view(M, N).double()
                print(f"   check {name}: mean |err| {(got - ref).abs().mean().item():.2e} (torch fp32 matmul: "
                      f"{((A @ W.t() + b).double() - (A.double() @ W.double().t() + b.double())).abs().mean().item():.2e})")
            print(f"x2 {name:5s} M={M} N={N} K={K}: median {med * 1e3:8.1f} us  min {ts[0] * 1e3:8.1f} us  "
                  f"{2 * M * N * K / med / 1e9:7.1f} TFLOP/s effective ({3 * 2 * M * N * K / med / 1e9:7.1f} TFLOP/s of MFMA work)")
        return
    if a.x3:
        for name in a.shapes.split(","):
            N, K, epi = shapes[name]
            epi = 1 if name == "fc1" else 0
            A = torch.randn(M, K, device="cuda")
            W = torch.randn(N, K, device="cuda") / K ** 0.5
            A3 = torch.empty(3, M, K, dtype=torch.bfloat16, device="cuda")
            W3 = torch.empty(3, N, K, dtype=torch.bfloat16, device="cuda")
            _lib.check(lib.d3dp_op_split3(A.data_ptr(), A3.data_ptr(), M * K, st))
            _lib.check(lib.d3dp_op_split3(W.data_ptr(), W3.data_ptr(), N * K, st))
            b = torch.randn(N, device="cuda")
            out = torch.empty(3 * M * N if epi else M * N * 2, device="cuda", dtype=torch.bfloat16)
            ts = []
            for i in range(a.iters + 3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                _lib.check(lib.d3dp_op_linear(2, epi, A3.data_ptr(), W3.data_ptr(), b.data_ptr(), out.data_ptr(), M, N, K, st))
                e1.record()
                torch.cuda.synchronize()
                if i >= 3:
                    ts.append(e0.elapsed_time(e1))
            ts.sort()
            med = ts[len(ts) // 2]
            print(f"x3 {name:5s} M={M} N={N} K={K}: median {med * 1e3:8.1f} us  {2 * M * N * K / med / 1e9:7.1f} TFLOP/s effective "
                  f"({6 * 2 * M * N * K / med / 1e9:7.1f} TFLOP/s of MFMA work)")
        return
    for name in a.shapes.split(","):
        N, K, epi = shapes[name]
        A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        W = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
        b = torch.randn(N, device="cuda")
        out = torch.empty(M, N, device="cuda", dtype=torch.float32 if epi & 16 else torch.bfloat16)
        for _ in range(3):
            _lib.check(lib.d3dp_op_linear(1, epi, A.data_ptr(), W.data_ptr(), b.data_ptr(), out.data_ptr(), M, N, K, st))
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.iters)]
        junk = torch.empty(256 * 1024 * 1024, dtype=torch.float32, device="cuda") if a.cache == "cold" else None
        src = A.clone() if a.cache == "produced" else None
        side = torch.empty(M, K, dtype=torch.float32, device="cuda") if a.cache == "produced" else None
        for e0, e1 in evs:
            if junk is not None:
                junk.fill_(1.0)
            if src is not None:
                side.add_(1.0)
                A.copy_(src)
            e0.record()
            _lib.check(lib.d3dp_op_linear(1, epi, A.data_ptr(), W.data_ptr(), b.data_ptr(), out.data_ptr(), M, N, K, st))
            e1.record()
        torch.cuda.synchronize()
        if a.check:
            ref = A.float() @ W.float().t() + b
            if epi & 1:
                ref = torch.nn.functional.gelu(ref)
            err = (out.float() - ref).abs().max().item() / ref.abs().max().item()
            print(f"   check {name}: max |err| / max |ref| = {err:.2e}  ({'OK' if err < 1e-2 else 'MISMATCH'})")
        ts = sorted(e0.elapsed_time(e1) for e0, e1 in evs)
        med = ts[len(ts) // 2]
        print(f"{name:5s} M={M} N={N} K={K} epi={epi:2d}: median {med * 1e3:8.1f} us  min {ts[0] * 1e3:8.1f} us  "
              f"{2 * M * N * K / med / 1e9:7.1f} TFLOP/s")


if __name__ == "__main__":
    main()
