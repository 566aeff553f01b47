"""GPU parity tests (run with ``-m gpu`` on an MI355X): libd3dp_hip.so, called through its C ABI, against
 (a) the CPU oracle on identical seeded inputs and (b) the committed reference-generated golden fixtures.

Tolerances
  exact mode : <= 1e-3 mm mean per-joint error (BASELINE.json north_star) for poses; fp32-class for single ops.
  fast mode  : bf16 MFMA inputs with fp32 accumulation.  Single ops are gated against the same op evaluated on
               bf16-ROUNDED operands in fp32 (tight: accumulation-order noise only).  End to end the deviation
               from the fp32 oracle is REPORTED and gated at FAST_TOL_MM; it cannot meet 1e-3 mm by construction
               (SURVEY.md §7 hard part 1: bf16 operand rounding alone gives mm-scale error on random weights).
"""
import ctypes as C
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from d3dp_amd import D3DP, _lib
from d3dp_amd.weights import (H36M_JOINTS_LEFT, H36M_JOINTS_RIGHT, flip_2d, make_state_dict, synthetic_inputs_2d,
                              synthetic_noise)
from oracle import d3dp_oracle as orc

pytestmark = pytest.mark.gpu
EXACT_TOL_MM = 1e-3
FAST_TOL_MM = 8.0           # vs the fp32 oracle: reported; above the largest deviation measured in four rounds (2.1 ... 6.5 mm)


@pytest.fixture(scope="module")
def lib():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return _lib.load()


def stream():
    return _lib.current_stream()


def bf16_round(t):
    return t.to(torch.bfloat16).to(torch.float32)


def make_model(frames, cs, dep, H, K, numerics, seed, scale=1.0, chunk_seqs=0):
    args = SimpleNamespace(number_of_frames=frames, test_time_augmentation=True, timestep=1000, scale=scale, cs=cs,
                           dep=dep, chunk_seqs=chunk_seqs)
    m = D3DP(args, H36M_JOINTS_LEFT, H36M_JOINTS_RIGHT, is_train=False, num_proposals=H, sampling_timesteps=K,
             numerics=numerics)
    m.load_state_dict(make_state_dict(seed, cs, dep, frames), strict=False)
    return m.cuda().eval()


def test_native_library_is_loaded(lib):
    maps = open("/proc/self/maps").read()
    assert "libd3dp_hip.so" in maps


def h2i_planes(buf, R, K):
    """(hi, lo) [R, K] views of a split-fp16 matrix in the library's line-interleaved "h2i" layout (common.h): a row is K/32
    blocks of 64 fp16, block kb = [hi of columns 32 kb .. 32 kb + 31 | lo of the same columns]."""
    v = buf.reshape(-1)[:2 * R * K].view(R, K // 32, 2, 32)
    return v[:, :, 0].reshape(R, K), v[:, :, 1].reshape(R, K)


# ------------------------------------------------------------------------------------------------ single ops
@pytest.mark.parametrize("mode", ["exact", "fast"])
@pytest.mark.parametrize("M,N,K", [(4131, 1536, 512), (300, 512, 1024), (17, 1024, 512), (1000, 64, 128),
                                   (70000, 512, 512)])
def test_linear_all_epilogues(lib, mode, M, N, K):
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) / K ** 0.5
    bias = torch.randn(N, generator=g)
    R = torch.randn(M, N, generator=g)
    fast = mode == "fast"
    Ar, Wr = (bf16_round(A), bf16_round(W)) if fast else (A, W)
    lin = (Ar.double() @ Wr.double().t() + bias.double())
    Ad = (A.to(torch.bfloat16) if fast else A).cuda().contiguous()
    Wd = (W.to(torch.bfloat16) if fast else W).cuda().contiguous()
    bd = bias.cuda()
    cases = [(_lib.EPI_BIAS, lin), (_lib.EPI_GELU, torch.nn.functional.gelu(lin))]
    cases += [(_lib.EPI_BIAS | 16, lin)] if fast else [(_lib.EPI_RESID, R.double() + lin)]   # fast: fp32-output form
    for epi, want in cases:
        f32_out = (not fast) or epi == _lib.EPI_RESID or bool(epi & 16)
        if epi == _lib.EPI_RESID:
            out = R.clone().cuda()
        else:
            out = torch.full((M, N), float("nan"), dtype=torch.float32 if f32_out else torch.bfloat16, device="cuda")
        _lib.check(lib.d3dp_op_linear(_lib.MODE_FAST if fast else _lib.MODE_EXACT, epi, Ad.data_ptr(), Wd.data_ptr(),
                                      bd.data_ptr(), out.data_ptr(), M, N, K, stream()))
        torch.cuda.synchronize()
        got = out.float().cpu().double()
        # the streaming kernel keeps finished tiles as packed bf16 before storing: bf16 precision even for fp32 output
        out_is_bf16 = (not f32_out) or fast
        atol = 2e-2 if out_is_bf16 else 2e-5 * K ** 0.5
        rtol = 1e-2 if out_is_bf16 else 1e-5
        assert torch.allclose(got, want, atol=atol, rtol=rtol), (mode, epi, (got - want).abs().max().item())


@pytest.mark.parametrize("M,N,K", [(4131, 1536, 512), (1000, 512, 1024), (300, 64, 128), (66000, 512, 512)])
def test_linear_split_bf16_is_fp32_class(lib, M, N, K):
    """EXACT-mode Linear: three bf16 planes per operand, six MFMA passes.  Must be at least as accurate as an fp32
    GEMM: mean error vs fp64 within 3x of torch's own (blocked, vectorised) fp32 matmul and far below one bf16 pass."""
    g = torch.Generator().manual_seed(M * 3 + N + K)
    A = torch.randn(M, K, generator=g) * 2
    W = torch.randn(N, K, generator=g) / K ** 0.5
    bias = torch.randn(N, generator=g)
    want = A.double() @ W.double().t() + bias.double()
    f32_err = ((A @ W.t() + bias).double() - want).abs().mean().item()
    Ad, Wd, bd = A.cuda(), W.cuda(), bias.cuda()
    A3 = torch.empty(3, M, K, dtype=torch.bfloat16, device="cuda")
    W3 = torch.empty(3, N, K, dtype=torch.bfloat16, device="cuda")
    _lib.check(lib.d3dp_op_split3(Ad.data_ptr(), A3.data_ptr(), M * K, stream()))
    _lib.check(lib.d3dp_op_split3(Wd.data_ptr(), W3.data_ptr(), N * K, stream()))
    assert torch.equal(A3.float().sum(0).cpu(), A)            # the split is exact
    out = torch.full((M, N), float("nan"), device="cuda")
    _lib.check(lib.d3dp_op_linear(_lib.MODE_SPLIT3, _lib.EPI_BIAS, A3.data_ptr(), W3.data_ptr(), bd.data_ptr(),
                                  out.data_ptr(), M, N, K, stream()))
    err = (out.cpu().double() - want).abs().mean().item()
    print(f"split-bf16 linear M={M} N={N} K={K}: mean |err| {err:.3e} (torch fp32 matmul {f32_err:.3e})")
    assert err <= 3.0 * f32_err and err < 2e-6
    # GELU epilogue re-split into planes
    out3 = torch.empty(3, M, N, dtype=torch.bfloat16, device="cuda")
    _lib.check(lib.d3dp_op_linear(_lib.MODE_SPLIT3, _lib.EPI_GELU, A3.data_ptr(), W3.data_ptr(), bd.data_ptr(),
                                  out3.data_ptr(), M, N, K, stream()))
    got = out3.float().sum(0).cpu().double()
    assert torch.allclose(got, torch.nn.functional.gelu(want), atol=2e-5, rtol=1e-5)


@pytest.mark.parametrize("M,N,K", [(4131, 1536, 512), (1000, 512, 1024), (300, 64, 128), (66000, 512, 512),
                                   (129, 192, 64), (17, 1024, 512)])
def test_linear_split_f16_is_fp32_class(lib, M, N, K):
    """The EXACT-mode Linear: two fp16 planes per operand (hi, lo * 2^11), three fp16-MFMA passes, separate fp32
    accumulator for the cross terms.  Gate: mean error vs fp64 within 3x of torch's own fp32 matmul (it is usually
    BELOW it), also with the power-of-two weight pre-scale the denoiser applies, and the GELU -> planes epilogue."""
    g = torch.Generator().manual_seed(M * 3 + N + K)
    A = torch.randn(M, K, generator=g) * 2
    A[0, :8] = torch.tensor([3e-6, -4e-6, 1e-8, 0.0, 3.9e-6, -1e-9, 1000.0, -3000.0])   # fp16-subnormal (x 16) and large inputs
    W = torch.randn(N, K, generator=g) / K ** 0.5
    bias = torch.randn(N, generator=g)
    want = A.double() @ W.double().t() + bias.double()
    f32_err = ((A @ W.t() + bias).double() - want).abs().mean().item()
    Ad, Wd, bd = A.cuda(), W.cuda(), bias.cuda()
    A2 = torch.empty(2, M, K, dtype=torch.float16, device="cuda")
    W2 = torch.empty(2, N, K, dtype=torch.float16, device="cuda")
    _lib.check(lib.d3dp_op_split2(Ad.data_ptr(), A2.data_ptr(), M * K, 16.0, stream()))    # activation scale
    w_scale = 2.0 ** (13 - int(np.floor(np.log2(W.abs().max().item()))))    # max |w| w_scale in [2^13, 2^14), as capi.hip
    _lib.check(lib.d3dp_op_split2(Wd.data_ptr(), W2.data_ptr(), N * K, w_scale, stream()))
    a_hi, a_lo = h2i_planes(A2, M, K)
    assert torch.equal(a_hi, (Ad * 16.0).half()) and torch.equal(a_lo, (Ad * 16.0 - a_hi.float()).half())   # layout + split
    rec = (a_hi.double() + a_lo.double()) / 16.0
    rel = ((rec.cpu() - A.double()).abs() / A.double().abs().clamp_min(1e-30))
    assert rel[A.abs() > 2.0 ** -7].max().item() <= 2.0 ** -21     # 22-bit representation wherever lo is a normal fp16
    assert (rec.cpu() - A.double()).abs()[A.abs() <= 2.0 ** -7].max().item() <= 2.0 ** -28   # absolute below that
    assert torch.isfinite(A2.float()).all()
    out = torch.full((M, N), float("nan"), device="cuda")
    _lib.check(lib.d3dp_op_linear_x2(_lib.EPI_BIAS, A2.data_ptr(), W2.data_ptr(), bd.data_ptr(), w_scale,
                                     out.data_ptr(), M, N, K, stream()))
    err = (out.cpu().double() - want).abs().mean().item()
    print(f"split-fp16 linear M={M} N={N} K={K}: mean |err| {err:.3e} (torch fp32 matmul {f32_err:.3e})")
    assert err <= 3.0 * f32_err
    # GELU epilogue re-split into planes
    out2 = torch.empty(2, M, N, dtype=torch.float16, device="cuda")
    _lib.check(lib.d3dp_op_linear_x2(_lib.EPI_GELU, A2.data_ptr(), W2.data_ptr(), bd.data_ptr(), w_scale,
                                     out2.data_ptr(), M, N, K, stream()))
    g_hi, g_lo = h2i_planes(out2, M, N)
    assert torch.equal(g_lo, ((g_hi.float() + g_lo.float()) - g_hi.float()).half())   # a proper (hi, lo) pair
    got = ((g_hi.double() + g_lo.double()) / 16.0).cpu()
    assert torch.allclose(got, torch.nn.functional.gelu(want), atol=2e-5, rtol=2e-5)
    # the tail of the range relies on the fp16 matrix cores NOT flushing subnormal inputs: hand-made planes
    if (M, N, K) == (129, 192, 64):
        Az = torch.zeros(2, M, K, dtype=torch.float16, device="cuda")
        Wz = torch.zeros(2, N, K, dtype=torch.float16, device="cuda")
        h2i_planes(Az, M, K)[0].fill_(3.0e-5)              # hi values: fp16 subnormal (smallest normal 6.1e-5)
        h2i_planes(Wz, N, K)[0].fill_(1.0)
        _lib.check(lib.d3dp_op_linear_x2(_lib.EPI_BIAS, Az.data_ptr(), Wz.data_ptr(), torch.zeros(N, device="cuda").data_ptr(),
                                         1.0, out.data_ptr(), M, N, K, stream()))
        assert abs(out[0, 0].item() - float(Az.reshape(-1)[0].float()) * K / 16.0) < 1e-9




def unpack_qkv_rows(buf, M, C):
    """Packed rows of the EXACT qkv Linear (epi 4): q fp32 [C] | k hi | k lo | v hi | v lo (fp16 [C] each, x 16) -> (M, 3C) fp64."""
    raw = buf.view(torch.uint8).reshape(M, 12 * C)
    q = raw[:, :4 * C].contiguous().view(torch.float32).double()
    h = raw[:, 4 * C:].contiguous().view(torch.float16).reshape(M, 4, C).double()
    return torch.cat((q, (h[:, 0] + h[:, 1]) / 16.0, (h[:, 2] + h[:, 3]) / 16.0), dim=1)


@pytest.mark.variants
@pytest.mark.parametrize("D", [4, 2, 1])
@pytest.mark.parametrize("M", [70000, 4131 + 29, 300])
def test_linear_split_f16_skewed_schedule(lib, M, D):
    """The row-class skewed schedule of the EXACT qkv and fc1 Linears (gemm_x2.hip: a tile's epilogue leaves under the next
    tile's k-loop).  Against fp64 like the plain kernel; and structurally: 16-row blocks of class 0 ((m / 16) % 4 == 0) sum
    their k-steps in the natural order, so they are BIT-IDENTICAL to the plain kernel's, the other classes differ by summation
    order only; and the result of a row depends on (m % 64) and its inputs alone -- the same rows 64 further down the
    matrix come out bit-identical (the batch-invariance the padded sequence pitch builds on)."""
    K, C = 512, 512
    g = torch.Generator().manual_seed(M + D)
    A = torch.randn(M, K, generator=g) * 2
    bd = torch.randn(3 * C, generator=g).cuda()
    Ad = A.cuda()
    A2 = torch.empty(2, M, K, dtype=torch.float16, device="cuda")
    _lib.check(lib.d3dp_op_split2(Ad.data_ptr(), A2.data_ptr(), M * K, 16.0, stream()))
    cls = (torch.arange(M) // 16) % 4
    for epi, N in ((_lib.EPI_QKV_PACK, 3 * C), (_lib.EPI_GELU, 2 * C)):
        W = torch.randn(N, K, generator=g) / K ** 0.5
        W2 = torch.empty(2, N, K, dtype=torch.float16, device="cuda")
        w_scale = 2.0 ** (13 - int(np.floor(np.log2(W.abs().max().item()))))
        _lib.check(lib.d3dp_op_split2(W.cuda().data_ptr(), W2.data_ptr(), N * K, w_scale, stream()))
        want = A.double() @ W.double().t() + bd[:N].cpu().double()
        if epi == _lib.EPI_GELU:
            want = torch.nn.functional.gelu(want)

        def run(e, a2, m):
            nbytes = m * (12 * C if epi == _lib.EPI_QKV_PACK else 4 * N)
            out = torch.full((nbytes,), 0xFF, dtype=torch.uint8, device="cuda")
            _lib.check(lib.d3dp_op_linear_x2(e, a2.data_ptr(), W2.data_ptr(), bd.data_ptr(), w_scale, out.data_ptr(), m, N, K,
                                             stream()), "d3dp_op_linear_x2")
            torch.cuda.synchronize()
            return out.cpu()

        def value(raw, m):
            if epi == _lib.EPI_QKV_PACK:
                return unpack_qkv_rows(raw, m, C)
            hi, lo = h2i_planes(raw.view(torch.float16), m, N)
            return (hi.double() + lo.double()) / 16.0

        plain, skew = run(epi, A2, M), run(epi | (D << 8), A2, M)
        e_plain, e_skew = (value(plain, M) - want).abs().mean().item(), (value(skew, M) - want).abs().mean().item()
        print(f"skewed schedule D={D} epi={epi} M={M}: mean |err| {e_skew:.3e} (plain schedule {e_plain:.3e})")
        assert e_skew <= 1.5 * e_plain + 1e-9 and torch.isfinite(value(skew, M)).all()
        rb = (12 * C) if epi == _lib.EPI_QKV_PACK else 4 * N
        p2, s2 = plain.reshape(M, rb), skew.reshape(M, rb)
        assert torch.equal(p2[cls == 0], s2[cls == 0])          # class 0: the natural k order
        assert not torch.equal(p2[cls == 1], s2[cls == 1])      # the others: rotated (different roundings somewhere)
        # shift the matrix down by 64 rows: every original row keeps its class and must keep its bits
        A2s = torch.zeros(2 * (M + 64) * K, dtype=torch.float16, device="cuda")
        A2s[2 * 64 * K:] = A2.reshape(-1)
        shifted = run(epi | (D << 8), A2s, M + 64).reshape(M + 64, rb)
        assert torch.equal(shifted[64:], s2)


@pytest.mark.variants
@pytest.mark.parametrize("M,N,K,epi", [(70000, 1536, 512, 4), (4131, 1024, 512, 1), (66000, 512, 1024, 2), (129, 192, 64, 0),
                                       (1000, 512, 512, 2), (300, 1536, 512, 4), (25000, 1024, 512, 1)])
def test_linear_split_f16_pingpong_is_bit_identical(lib, M, N, K, epi):
    """The ping-pong form of the EXACT Linear (gemm_x2.hip: the two compute waves of a SIMD half a k-step apart -- one reads
    its fragments while the other multiplies) runs the same MFMAs in the same order on the same operands: every epilogue's
    output equals the lock-step kernel's bit for bit."""
    g = torch.Generator().manual_seed(M + N + K + epi)
    A = (torch.randn(M, K, generator=g) * 2).cuda()
    W = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    bias = torch.randn(N, generator=g).cuda()
    x0 = torch.randn(M, N, generator=g).cuda()
    A2 = torch.empty(2, M, K, dtype=torch.float16, device="cuda")
    W2 = torch.empty(2, N, K, dtype=torch.float16, device="cuda")
    w_scale = 2.0 ** (13 - int(np.floor(np.log2(W.abs().max().item()))))
    _lib.check(lib.d3dp_op_split2(A.data_ptr(), A2.data_ptr(), M * K, 16.0, stream()))
    _lib.check(lib.d3dp_op_split2(W.data_ptr(), W2.data_ptr(), N * K, w_scale, stream()))
    outs = []
    for flag in (0, 2048):
        out = x0.clone() if epi == 2 else torch.full((M, N), float("nan"), device="cuda")
        _lib.check(lib.d3dp_op_linear_x2(epi | flag, A2.data_ptr(), W2.data_ptr(), bias.data_ptr(), w_scale, out.data_ptr(), M, N,
                                         K, stream()), "d3dp_op_linear_x2")
        torch.cuda.synchronize()
        outs.append(out.view(torch.int32).cpu())
    assert torch.equal(outs[0], outs[1])


@pytest.mark.variants
@pytest.mark.parametrize("M,N,K,epi", [(70000, 1536, 512, 4), (4131, 1024, 512, 1), (66000, 512, 1024, 2), (129, 256, 64, 0),
                                       (1000, 512, 512, 2), (300, 1536, 512, 4), (25000, 1024, 512, 1), (256, 512, 128, 0),
                                       (128960, 1536, 512, 4)])
def test_linear_split_f16_wide_is_bit_identical(lib, M, N, K, epi):
    """The wide form of the EXACT Linear (gemm_x2.hip: 256 x 256 tiles, every wave loads its own share by LDS-DMA and counts
    its epilogue stores into the same vmcnt) meets every accumulator with the products of the plain kernel in its order:
    every epilogue's output equals the plain kernel's bit for bit -- at the full size too, where each workgroup walks eleven
    or twelve tiles and every counted wait is exercised."""
    g = torch.Generator().manual_seed(M + N + K + epi)
    A = (torch.randn(M, K, generator=g) * 2).cuda()
    W = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    bias = torch.randn(N, generator=g).cuda()
    x0 = torch.randn(M, N, generator=g).cuda()
    A2 = torch.empty(2, M, K, dtype=torch.float16, device="cuda")
    W2 = torch.empty(2, N, K, dtype=torch.float16, device="cuda")
    w_scale = 2.0 ** (13 - int(np.floor(np.log2(W.abs().max().item()))))
    _lib.check(lib.d3dp_op_split2(A.data_ptr(), A2.data_ptr(), M * K, 16.0, stream()))
    _lib.check(lib.d3dp_op_split2(W.data_ptr(), W2.data_ptr(), N * K, w_scale, stream()))
    outs = []
    for flag in (0, 4096, 4096):
        out = x0.clone() if epi == 2 else torch.full((M, N), float("nan"), device="cuda")
        _lib.check(lib.d3dp_op_linear_x2(epi | flag, A2.data_ptr(), W2.data_ptr(), bias.data_ptr(), w_scale, out.data_ptr(), M, N,
                                         K, stream()), "d3dp_op_linear_x2")
        torch.cuda.synchronize()
        outs.append(out.view(torch.int32).cpu())
    assert torch.equal(outs[0], outs[1])
    assert torch.equal(outs[0], outs[2])


@pytest.mark.variants
def test_skewed_schedule_end_to_end_keeps_parity_and_batch_invariance(monkeypatch):
    """D3DP_X2_SKEW=4 (the experiment of gemm_x2.hip kept behind a switch: measured slower, off by default) through the whole
    denoiser: every sequence is padded to a multiple of 64 rows, so a token's summation order depends on its index in its
    sequence alone -- the sampler output meets the exact tolerance and a clip's result does not depend on what else is in
    the batch, bit for bit."""
    monkeypatch.setenv("D3DP_X2_SKEW", "4")
    frames, H, K = 27, 3, 2
    m = make_model(frames, 512, 8, H, K, "exact", seed=7)
    x2d = torch.from_numpy(synthetic_inputs_2d(31, 3, frames)).cuda()
    x2f = flip_2d(x2d)
    noises = [torch.from_numpy(synthetic_noise(40 + k, (3, H, frames, 17, 3))).cuda() for k in range(K)]
    full = m(x2d, None, input_2d_flip=x2f, noise=noises)
    one = m(x2d[1:2], None, input_2d_flip=x2f[1:2], noise=[n[1:2] for n in noises])
    assert torch.equal(one, full[1:2])
    sd = make_state_dict(7, 512, 8, frames)
    want = orc.ddim_sample_flip(orc.strip_prefix(sd), orc.cosine_schedule(1000), x2d.cpu(), x2f.cpu(), H, K, 8,
                                H36M_JOINTS_LEFT, H36M_JOINTS_RIGHT, [n.cpu() for n in noises])
    e = orc.mpjpe_mm(full.cpu(), want)
    print(f"skewed schedule, F=27 B=3 H=3 K=2: {e:.3e} mm vs the fp32 oracle")
    assert e <= EXACT_TOL_MM


def test_linear_split_f16_rejects_k_not_multiple_of_64(lib):
    """The k-loop runs two k-steps of 32 per iteration: K = 96 must be refused, not mis-computed."""
    M, N, K = 300, 128, 96
    A2 = torch.zeros(2, M, K, dtype=torch.float16, device="cuda")
    W2 = torch.zeros(2, N, K, dtype=torch.float16, device="cuda")
    out = torch.zeros(M, N, device="cuda")
    rc = lib.d3dp_op_linear_x2(_lib.EPI_BIAS, A2.data_ptr(), W2.data_ptr(), torch.zeros(N, device="cuda").data_ptr(), 1.0,
                               out.data_ptr(), M, N, K, stream())
    assert rc != 0 and b"d3dp_launch_linear_f16x2" in lib.d3dp_last_error()


@pytest.mark.parametrize("M,N,K", [(4131, 512, 512), (1000, 512, 1024), (129, 192, 64)])
def test_linear_split_f16_residual_epilogue_adds_in_place(lib, M, N, K):
    """EPI_RESID (proj and fc2 of the EXACT denoiser): x += A W^T + b on the fp32 residual stream, bit for bit the plain
    epilogue's result added to x in fp32 (what the row kernels did before the add moved into the Linear)."""
    g = torch.Generator().manual_seed(M + N + K)
    A = (torch.randn(M, K, generator=g) * 2).cuda()
    W = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    bias = torch.randn(N, generator=g).cuda()
    x = torch.randn(M, N, generator=g).cuda()
    A2 = torch.empty(2, M, K, dtype=torch.float16, device="cuda")
    W2 = torch.empty(2, N, K, dtype=torch.float16, device="cuda")
    w_scale = 2.0 ** (13 - int(np.floor(np.log2(W.abs().max().item()))))
    _lib.check(lib.d3dp_op_split2(A.data_ptr(), A2.data_ptr(), M * K, 16.0, stream()))
    _lib.check(lib.d3dp_op_split2(W.data_ptr(), W2.data_ptr(), N * K, w_scale, stream()))
    plain = torch.empty(M, N, device="cuda")
    _lib.check(lib.d3dp_op_linear_x2(_lib.EPI_BIAS, A2.data_ptr(), W2.data_ptr(), bias.data_ptr(), w_scale,
                                     plain.data_ptr(), M, N, K, stream()))
    acc = x.clone()
    _lib.check(lib.d3dp_op_linear_x2(_lib.EPI_RESID, A2.data_ptr(), W2.data_ptr(), bias.data_ptr(), w_scale,
                                     acc.data_ptr(), M, N, K, stream()))
    torch.cuda.synchronize()
    assert torch.equal(acc, x + plain)


@pytest.mark.parametrize("M,C", [(4131, 512), (700, 128), (129, 64)])
def test_qkv_linear_packed_epilogue_is_the_split_of_the_plain_one(lib, M, C):
    """EPI_QKV_PACK (what the EXACT denoiser's qkv Linear writes): rows of 12 C bytes, q fp32 | k hi | k lo | v hi | v lo
    with hi = fp16(16 x), lo = fp16(16 x - hi) -- bit for bit the split of the fp32 result of the plain epilogue."""
    N, K = 3 * C, C
    g = torch.Generator().manual_seed(M + C)
    A = (torch.randn(M, K, generator=g) * 2).cuda()
    W = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    bias = torch.randn(N, generator=g).cuda()
    A2 = torch.empty(2, M, K, dtype=torch.float16, device="cuda")
    W2 = torch.empty(2, N, K, dtype=torch.float16, device="cuda")
    w_scale = 2.0 ** (13 - int(np.floor(np.log2(W.abs().max().item()))))
    _lib.check(lib.d3dp_op_split2(A.data_ptr(), A2.data_ptr(), M * K, 16.0, stream()))
    _lib.check(lib.d3dp_op_split2(W.data_ptr(), W2.data_ptr(), N * K, w_scale, stream()))
    plain = torch.empty(M, N, device="cuda")
    _lib.check(lib.d3dp_op_linear_x2(_lib.EPI_BIAS, A2.data_ptr(), W2.data_ptr(), bias.data_ptr(), w_scale,
                                     plain.data_ptr(), M, N, K, stream()))
    packed = torch.full((M, N), float("nan"), device="cuda")
    _lib.check(lib.d3dp_op_linear_x2(_lib.EPI_QKV_PACK, A2.data_ptr(), W2.data_ptr(), bias.data_ptr(), w_scale,
                                     packed.data_ptr(), M, N, K, stream()))
    torch.cuda.synchronize()
    assert torch.equal(packed[:, :C], plain[:, :C])                      # q: fp32, untouched
    planes = packed[:, C:].contiguous().view(torch.float16).view(M, 4, C)   # k hi | k lo | v hi | v lo
    for i, col0 in ((0, C), (2, 2 * C)):
        y = plain[:, col0:col0 + C] * 16.0
        hi = y.half()
        lo = (y - hi.float()).half()
        assert torch.equal(planes[:, i], hi) and torch.equal(planes[:, i + 1], lo)


def ref_attention(qkv, n_bh, F, J, C, heads, axis):
    """fp64 reference on the (n_bh, F, J, 3C) layout."""
    hd = C // heads
    x = qkv.double().reshape(n_bh, F, J, 3, heads, hd)
    q, k, v = x[:, :, :, 0], x[:, :, :, 1], x[:, :, :, 2]            # (bh, F, J, h, d)
    if axis == 0:   # sequences over joints
        q, k, v = (t.permute(0, 1, 3, 2, 4) for t in (q, k, v))       # (bh, F, h, J, d)
    else:           # sequences over frames
        q, k, v = (t.permute(0, 2, 3, 1, 4) for t in (q, k, v))       # (bh, J, h, F, d)
    a = torch.softmax(q @ k.transpose(-1, -2) * hd ** -0.5, dim=-1) @ v
    a = a.permute(0, 1, 3, 2, 4) if axis == 0 else a.permute(0, 3, 1, 2, 4)   # -> (bh, F, J, h, d)
    return a.reshape(n_bh * F * J, C)


@pytest.mark.parametrize("act,impl,axis,F,C", [
    ("f32", 0, 0, 27, 512), ("f32", 0, 1, 27, 512), ("f32", 0, 1, 243, 512), ("f32", 1, 1, 243, 512), ("f32", 1, 1, 27, 512),
    ("f32", 1, 1, 100, 512), ("f32", 0, 0, 9, 64), ("f32", 0, 1, 9, 64),
    ("f32", 2, 0, 27, 512), ("f32", 2, 0, 243, 512), ("f32", 2, 1, 27, 512), ("f32", 2, 1, 243, 512), ("f32", 2, 1, 100, 512),
    # the persistent LDS-DMA kernel's masking paths: keys end inside the last tile (49, 243), whole tiles masked (9, 33,
    # 130), nothing masked (256)
    ("f32", 2, 1, 9, 512), ("f32", 2, 1, 33, 512), ("f32", 2, 1, 49, 512), ("f32", 2, 1, 130, 512), ("f32", 2, 1, 256, 512),
    ("bf16", 0, 0, 27, 512), ("bf16", 1, 0, 27, 512), ("bf16", 1, 0, 243, 512), ("bf16", 1, 1, 27, 512), ("bf16", 1, 1, 243, 512), ("bf16", 1, 1, 100, 512),
    ("bf16", 0, 1, 243, 512),
    # clips longer than the MFMA attention kernels' LDS images (`-f 351`, reference common/arguments.py:58): the row kernel
    # passes K / V through LDS in chunks of 256 keys under its online softmax, 256 query rows at a time
    ("f32", 0, 1, 300, 512), ("f32", 0, 1, 351, 512), ("f32", 0, 1, 513, 512), ("bf16", 0, 1, 351, 512),
    # ... and EXACT mode's long-clip kernel (round 6: split-fp16 operands, keys in chunks of 128 under an online softmax): the last
    # chunk ends inside a key tile (257, 351), on a pair boundary (288), on a chunk boundary (384); three and eight chunks;
    # the last query group holds one tile (257), five (351) or all eight (384)
    ("f32", 2, 1, 257, 512), ("f32", 2, 1, 288, 512), ("f32", 2, 1, 351, 512), ("f32", 2, 1, 384, 512), ("f32", 2, 1, 513, 512),
    ("f32", 2, 1, 1000, 512)])
def test_attention(lib, act, impl, axis, F, C):
    n_bh, J, heads = 2, 17, 8
    g = torch.Generator().manual_seed(F * 7 + C + axis)
    qkv = torch.randn(n_bh * F * J, 3 * C, generator=g)
    qkv[:, :C] *= 2.0        # sharpen the softmax a little
    bf = act == "bf16"
    src = bf16_round(qkv) if bf else qkv
    want = ref_attention(src, n_bh, F, J, C, heads, axis)
    qd = (qkv.to(torch.bfloat16) if bf else qkv).cuda().contiguous()
    out = torch.full((n_bh * F * J, C), float("nan"), dtype=qd.dtype, device="cuda")
    _lib.check(lib.d3dp_op_attention(int(bf), impl, axis, qd.data_ptr(), out.data_ptr(), n_bh, F, J, C, heads, stream()))
    torch.cuda.synchronize()
    got = out.float().cpu().double()
    assert torch.isfinite(got).all()
    atol = 2e-2 if bf else 2e-5
    assert torch.allclose(got, want, atol=atol, rtol=1e-2 if bf else 1e-4), (got - want).abs().max().item()
    if not bf:
        err = (got - want).abs().mean().item()
        print(f"attention act={act} impl={impl} axis={axis} F={F}: mean |err| vs fp64 {err:.2e}")
        if impl == 2:      # the EXACT-mode kernel (split-fp16 operands) must be fp32-class: compare with the fp32 kernels
            assert err < 5e-7


def test_attention_split_f16_many_problems_per_workgroup(lib):
    """The EXACT temporal kernel is persistent: with more (sequence, head) problems than resident workgroups every
    workgroup loops, K(p+1) and V(p+1) streaming into the LDS images problem p is still being read from."""
    n_bh, F, J, C, heads = 24, 27, 17, 512, 8          # 3264 problems
    g = torch.Generator().manual_seed(99)
    qkv = torch.randn(n_bh * F * J, 3 * C, generator=g)
    want = ref_attention(qkv, n_bh, F, J, C, heads, 1)
    qd = qkv.cuda().contiguous()
    out = torch.full((n_bh * F * J, C), float("nan"), device="cuda")
    _lib.check(lib.d3dp_op_attention(0, 2, 1, qd.data_ptr(), out.data_ptr(), n_bh, F, J, C, heads, stream()))
    torch.cuda.synchronize()
    got = out.cpu().double()
    assert torch.allclose(got, want, atol=2e-5, rtol=1e-4), (got - want).abs().max().item()
    assert (got - want).abs().mean().item() < 5e-7


def test_attention_softmax_spike(lib):
    """One key dominating one query row (a large raw score) must not overflow or lose the row (max-subtraction)."""
    n_bh, F, J, C, heads = 1, 243, 17, 512, 8
    g = torch.Generator().manual_seed(5)
    qkv = torch.randn(n_bh * F * J, 3 * C, generator=g)
    qkv[100 * J + 3, :C] *= 30.0
    qkv[7 * J + 3, C:2 * C] = qkv[100 * J + 3, :C] / 30.0 * 4.0
    for bf, impl in ((False, 0), (False, 1), (False, 2), (True, 1)):
        src = bf16_round(qkv) if bf else qkv
        want = ref_attention(src, n_bh, F, J, C, heads, 1)
        qd = (qkv.to(torch.bfloat16) if bf else qkv).cuda().contiguous()
        out = torch.empty((n_bh * F * J, C), dtype=qd.dtype, device="cuda")
        _lib.check(lib.d3dp_op_attention(int(bf), impl, 1, qd.data_ptr(), out.data_ptr(), n_bh, F, J, C, heads, stream()))
        got = out.float().cpu().double()
        assert torch.isfinite(got).all()
        assert torch.allclose(got, want, atol=3e-2 if bf else 5e-5, rtol=2e-2 if bf else 1e-4)


@pytest.mark.parametrize("C_", [64, 128, 512])
def test_layernorm(lib, C_):
    T = 1001
    g = torch.Generator().manual_seed(C_)
    x = torch.randn(T, C_, generator=g) * 3 + 0.5
    w, b = torch.randn(C_, generator=g), torch.randn(C_, generator=g)
    want = torch.nn.functional.layer_norm(x.double(), (C_,), w.double(), b.double(), 1e-6)
    xd, wd, bd = x.cuda(), w.cuda(), b.cuda()
    for bf in (0, 1):
        out = torch.empty((T, C_), dtype=torch.bfloat16 if bf else torch.float32, device="cuda")
        _lib.check(lib.d3dp_op_layernorm(bf, xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), 1e-6, out.data_ptr(), T, C_,
                                         stream()))
        got = out.float().cpu().double()
        assert torch.allclose(got, want, atol=3e-2 if bf else 2e-5, rtol=1e-2 if bf else 1e-5)
    # out type 3: the split-fp16 operand of the EXACT Linears, h2i layout -- bit for bit the split of the fp32 result
    out32 = torch.empty((T, C_), device="cuda")
    _lib.check(lib.d3dp_op_layernorm(0, xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), 1e-6, out32.data_ptr(), T, C_, stream()))
    out2 = torch.full((2 * T * C_,), float("nan"), dtype=torch.float16, device="cuda")
    _lib.check(lib.d3dp_op_layernorm(3, xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), 1e-6, out2.data_ptr(), T, C_, stream()))
    hi, lo = h2i_planes(out2, T, C_)
    y = out32 * 16.0
    assert torch.equal(hi, y.half()) and torch.equal(lo, (y - y.half().float()).half())


# ------------------------------------------------------------------------------------------------ denoiser
def load_g(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


def test_g2_tiny_denoiser_exact(golden_dir):
    g = load_g(golden_dir, "g2_tiny_denoiser")
    m = make_model(int(g["frames"]), int(g["cs"]), int(g["dep"]), 3, 1, "exact", int(g["seed"]))
    out = m.pose_estimator(torch.from_numpy(g["x2d"]).cuda(), torch.from_numpy(g["x3d"]).cuda(),
                           torch.from_numpy(g["t"]).cuda())
    assert orc.mpjpe_mm(out.cpu(), torch.from_numpy(g["out"])) <= EXACT_TOL_MM


@pytest.mark.parametrize("frames", [27, 243])
def test_g3_full_width_denoiser(golden_dir, frames):
    g = load_g(golden_dir, f"g3_denoiser_F{frames}")
    x2d = torch.from_numpy(synthetic_inputs_2d(int(g["x2d_seed"]), 1, frames)).cuda()
    x3d = torch.from_numpy(synthetic_noise(int(g["x3d_seed"]), (1, 1, frames, 17, 3))).cuda()
    exact = make_model(frames, 512, 8, 1, 1, "exact", int(g["seed"]))
    fast = make_model(frames, 512, 8, 1, 1, "fast", int(g["seed"]))
    for tt in (999, 499, 99):
        t = torch.tensor([tt], device="cuda")
        want = torch.from_numpy(g[f"out_t{tt}"])
        e = orc.mpjpe_mm(exact.pose_estimator(x2d, x3d, t).cpu(), want)
        f = orc.mpjpe_mm(fast.pose_estimator(x2d, x3d, t).cpu(), want)
        print(f"[F={frames} t={tt}] MPJPE vs reference golden: exact {e:.3e} mm, fast(bf16) {f:.3e} mm")
        assert e <= EXACT_TOL_MM
        assert f <= FAST_TOL_MM


def test_train_branch_forward_matches_golden(golden_dir):
    g = load_g(golden_dir, "g6_train_step")
    cs, dep, Fr = int(g["cs"]), int(g["dep"]), int(g["frames"])
    args = SimpleNamespace(number_of_frames=Fr, test_time_augmentation=True, timestep=1000, scale=1.0, cs=cs, dep=dep)
    m = D3DP(args, H36M_JOINTS_LEFT, H36M_JOINTS_RIGHT, is_train=True, numerics="exact")
    m.load_state_dict(make_state_dict(int(g["seed"]), cs, dep, Fr), strict=False)
    m = m.cuda()
    with torch.no_grad():
        pred = m(torch.from_numpy(g["x2d"]).cuda(), torch.from_numpy(g["gt"]).cuda(),
                 t=torch.from_numpy(g["t"]).reshape(-1, 1), noise=torch.from_numpy(g["noise"]))
    assert orc.mpjpe_mm(pred.cpu(), torch.from_numpy(g["pred_nodrop"])) <= EXACT_TOL_MM


@pytest.mark.parametrize("tag", ["nodrop", "drop"])
def test_training_step_matches_reference(golden_dir, tag):
    """BASELINE config 5 in miniature (fixture g6: F=27, B=4, cs=64, dep=2): q_sample + MixSTE2 train-branch forward,
    MPJPE loss, backward seeded with the loss value (main.py:393) -- loss, prediction and the gradient norms of four
    parameters against the reference's own autograd, without and with (recorded) DropPath masks."""
    g = load_g(golden_dir, "g6_train_step")
    cs, dep, Fr = int(g["cs"]), int(g["dep"]), int(g["frames"])
    args = SimpleNamespace(number_of_frames=Fr, test_time_augmentation=True, timestep=1000, scale=1.0, cs=cs, dep=dep)
    m = D3DP(args, H36M_JOINTS_LEFT, H36M_JOINTS_RIGHT, is_train=True)
    assert m.pose_estimator.numerics == "train"
    m.load_state_dict(make_state_dict(int(g["seed"]), cs, dep, Fr), strict=False)
    m = m.cuda().train()
    if tag == "nodrop":
        m.eval()                       # DropPath off (nn.Module.training False); D3DP.is_train keeps the train branch
    dpd = None
    if tag == "drop":
        masks = [torch.from_numpy(g[f"drop_mask{k}"]) for k in range(int(g["drop_masks_n"]))]
        it, dpd = iter(masks), {}
        for i in range(1, dep):        # rate linspace(0, 0.1, dep)[0] == 0 -> Identity
            dpd[f"STEblocks.{i}"] = (next(it), next(it))
            dpd[f"TTEblocks.{i}"] = (next(it), next(it))
    gt = torch.from_numpy(g["gt"]).cuda()
    pred = m(torch.from_numpy(g["x2d"]).cuda(), gt, t=torch.from_numpy(g["t"]).reshape(-1, 1),
             noise=torch.from_numpy(g["noise"]), droppath=dpd)
    assert pred.requires_grad
    assert orc.mpjpe_mm(pred.detach().cpu(), torch.from_numpy(g[f"pred_{tag}"])) <= EXACT_TOL_MM
    loss = torch.mean(torch.norm(pred - gt, dim=-1))            # loss.py:13
    assert abs(loss.item() - float(g[f"loss_{tag}"])) < 2e-6
    loss.backward(loss.clone().detach())                         # main.py:393
    params = dict(m.named_parameters())
    checked = 0
    for key in g.files:
        if key.startswith(f"gradnorm_{tag}::"):
            name = key.split("::", 1)[1]
            got = params[name].grad.double().norm().item()
            want = float(g[key])
            print(f"[{tag}] |grad {name}| = {got:.6e} (reference {want:.6e})")
            assert got == pytest.approx(want, rel=2e-4), name
            checked += 1
    assert checked == 4
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())
    # every parameter's gradient against torch autograd through the CPU oracle (same inputs, masks and seeding)
    sd = make_state_dict(int(g["seed"]), cs, dep, Fr)
    po = {k: v.clone().requires_grad_(True) for k, v in orc.strip_prefix(sd).items()}
    xp = orc.prepare_targets(orc.cosine_schedule(1000), torch.from_numpy(g["gt"]), torch.from_numpy(g["t"]),
                             torch.from_numpy(g["noise"]))
    pred_o = orc.mixste_forward(po, torch.from_numpy(g["x2d"]), xp, torch.from_numpy(g["t"]), dep, droppath=dpd)
    loss_o = torch.mean(torch.norm(pred_o - torch.from_numpy(g["gt"]), dim=-1))
    loss_o.backward(loss_o.clone().detach())
    worst = 0.0
    for name, p in m.pose_estimator.named_parameters():
        ref = po[name].grad
        err = (p.grad.cpu().double() - ref.double()).norm().item() / max(ref.double().norm().item(), 1e-12)
        worst = max(worst, err)
        assert err < 2e-3, (name, err)
    print(f"[{tag}] worst relative gradient error over {len(po)} parameters: {worst:.2e}")


# ------------------------------------------------------------------------------------------------ sampler
@pytest.mark.parametrize("name", ["g4_sampler_c1", "g4_sampler_H3K5", "g4_sampler_tiny_K10"])
def test_g4_sampler_exact(golden_dir, name):
    g = load_g(golden_dir, name)
    cs, dep, Fr, B, H, K = (int(g[k]) for k in ("cs", "dep", "frames", "B", "H", "K"))
    x2d = synthetic_inputs_2d(int(g["x2d_seed"]), B, Fr)
    noises = [torch.from_numpy(synthetic_noise(int(g["noise_seed"]) + k, (B, H, Fr, 17, 3))) for k in range(K)]
    m = make_model(Fr, cs, dep, H, K, "exact", int(g["seed"]))
    out = m(torch.from_numpy(x2d).cuda(), None, input_2d_flip=torch.from_numpy(flip_2d(x2d)).cuda(), noise=noises)
    assert out.shape == (B, K, H, Fr, 17, 3) and out.dtype == torch.float32
    per_step = [orc.mpjpe_mm(out[:, k].cpu(), torch.from_numpy(g["out"][:, k])) for k in range(K)]
    print(f"[{name}] exact MPJPE per step (mm): {['%.2e' % v for v in per_step]}")
    assert max(per_step) <= EXACT_TOL_MM
    out[:, :, :, :, 0] = 0      # callers write into the result in place (main.py:700)


@pytest.mark.parametrize("numerics", ["exact", "fast"])
def test_sampler_on_a_clip_longer_than_256_frames(numerics, monkeypatch):
    """VERDICT r4 missing 3: the reference takes any `-f` (common/arguments.py:58, mixste.py:172); 351 frames used to be
    refused with ENOTSUP.  EXACT mode keeps its split-fp16 Linears; its temporal attention runs the chunked-key flash kernel on
    the same split-fp16 operands (round 6; D3DP_LONG_ATTN=rows: round 5's fp32 row kernel for both attentions, kept as the
    cross-check -- both within the 1e-3 mm tolerance against the oracle; cs = 512, dep = 2, H = 2, K = 2)."""
    frames, cs, dep, B, H, K = 351, 512, 2, 1, 2, 2
    sd = make_state_dict(13, cs, dep, frames)
    x2d = synthetic_inputs_2d(131, B, frames)
    noises = [torch.from_numpy(synthetic_noise(140 + k, (B, H, frames, 17, 3))) for k in range(K)]
    want = orc.ddim_sample_flip(orc.strip_prefix(sd), orc.cosine_schedule(1000), torch.from_numpy(x2d),
                                torch.from_numpy(flip_2d(x2d)), H, K, dep, H36M_JOINTS_LEFT, H36M_JOINTS_RIGHT, noises)
    m = make_model(frames, cs, dep, H, K, numerics, 13)
    out = m(torch.from_numpy(x2d).cuda(), None, input_2d_flip=torch.from_numpy(flip_2d(x2d)).cuda(), noise=noises)
    assert out.shape == (B, K, H, frames, 17, 3) and torch.isfinite(out).all()
    err = orc.mpjpe_mm(out.cpu(), want)
    print(f"F=351 {numerics}: MPJPE vs the fp32 oracle {err:.3e} mm")
    assert err <= (EXACT_TOL_MM if numerics == "exact" else FAST_TOL_MM)
    if numerics == "exact":
        assert m.pose_estimator.exact_scales()[2] == "f16x2"      # the Linears stay on the split-fp16 kernels
        monkeypatch.setenv("D3DP_LONG_ATTN", "rows")
        m2 = make_model(frames, cs, dep, H, K, numerics, 13)
        out2 = m2(torch.from_numpy(x2d).cuda(), None, input_2d_flip=torch.from_numpy(flip_2d(x2d)).cuda(), noise=noises)
        err2 = orc.mpjpe_mm(out2.cpu(), want)
        print(f"F=351 exact, D3DP_LONG_ATTN=rows: {err2:.3e} mm; the two implementations apart: {orc.mpjpe_mm(out.cpu(), out2.cpu()):.3e} mm")
        assert err2 <= EXACT_TOL_MM and not torch.equal(out, out2)    # (different kernels did run)


@pytest.mark.parametrize("cs", [256, 128])
@pytest.mark.parametrize("numerics", ["exact", "fast"])
def test_sampler_at_the_reference_small_width(numerics, cs):
    """cs = 256 (the reference's smaller `-cs`; 8 heads of 32 channels, hidden 512) and cs = 128: the sampler against the oracle
    at the same tolerances as cs = 512.  The 64-wide-head attention kernels do not apply at these widths -- the library must
    route around them, not fail."""
    frames, dep, B, H, K = 27, 2, 2, 2, 2
    sd = make_state_dict(29, cs, dep, frames)
    x2d = synthetic_inputs_2d(291, B, frames)
    noises = [torch.from_numpy(synthetic_noise(292 + k, (B, H, frames, 17, 3))) for k in range(K)]
    want = orc.ddim_sample_flip(orc.strip_prefix(sd), orc.cosine_schedule(1000), torch.from_numpy(x2d),
                                torch.from_numpy(flip_2d(x2d)), H, K, dep, H36M_JOINTS_LEFT, H36M_JOINTS_RIGHT, noises)
    m = make_model(frames, cs, dep, H, K, numerics, 29)
    out = m(torch.from_numpy(x2d).cuda(), None, input_2d_flip=torch.from_numpy(flip_2d(x2d)).cuda(), noise=noises)
    assert out.shape == (B, K, H, frames, 17, 3) and torch.isfinite(out).all()
    err = orc.mpjpe_mm(out.cpu(), want)
    print(f"cs={cs} {numerics}: MPJPE vs the fp32 oracle {err:.3e} mm")
    assert err <= (EXACT_TOL_MM if numerics == "exact" else FAST_TOL_MM)


def test_fast_mode_on_fp16_operands_is_closer_to_the_reference_than_on_bf16():
    """VERDICT r4 item 7 / r5 item 7 (optional, no parity claim): lib/variants/libd3dp_fastf16.so is the library with FAST mode's
    2-byte operand type swapped from bf16 to IEEE fp16 (common.h D3DP_FAST_F16; `make fastf16`).  A child process samples the smoke
    problem on it: three more significand bits in every operand must show as a several times smaller distance to the fp32 oracle
    than the bf16 library's (measured: 0.4 mm against 3.4), and the EXACT mode of that build is unchanged."""
    import json
    import subprocess
    import sys
    lib = os.path.join(os.path.dirname(_lib.LIB_PATH), "variants", "libd3dp_fastf16.so")
    if not os.path.exists(lib):
        pytest.skip("make -C d3dp_amd/csrc fastf16 was not run")
    code = ("import json, bench; p = bench.quick_parity(); print(json.dumps({k: p[k] for k in ('exact_mpjpe_mm', 'fast_mpjpe_mm')}))")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for name, env in (("bf16", {}), ("fp16", {"D3DP_LIB": lib})):
        e = dict(os.environ, **env)
        e.pop("D3DP_LIB", None) if not env else None
        out = subprocess.run([sys.executable, "-c", code], cwd=root, env=e, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        res[name] = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    print(f"FAST mode vs the fp32 oracle: bf16 operands {res['bf16']['fast_mpjpe_mm']:.3f} mm, fp16 operands {res['fp16']['fast_mpjpe_mm']:.3f} mm")
    assert res["fp16"]["fast_mpjpe_mm"] < 0.4 * res["bf16"]["fast_mpjpe_mm"]
    assert res["fp16"]["exact_mpjpe_mm"] == res["bf16"]["exact_mpjpe_mm"] <= EXACT_TOL_MM


def test_sampler_fast_mode_reported(golden_dir):
    g = load_g(golden_dir, "g4_sampler_H3K5")
    cs, dep, Fr, B, H, K = (int(g[k]) for k in ("cs", "dep", "frames", "B", "H", "K"))
    x2d = synthetic_inputs_2d(int(g["x2d_seed"]), B, Fr)
    noises = [torch.from_numpy(synthetic_noise(int(g["noise_seed"]) + k, (B, H, Fr, 17, 3))) for k in range(K)]
    m = make_model(Fr, cs, dep, H, K, "fast", int(g["seed"]))
    out = m(torch.from_numpy(x2d).cuda(), None, input_2d_flip=torch.from_numpy(flip_2d(x2d)).cuda(), noise=noises)
    per_step = [orc.mpjpe_mm(out[:, k].cpu(), torch.from_numpy(g["out"][:, k])) for k in range(K)]
    print(f"fast(bf16) MPJPE vs reference per DDIM step (mm): {['%.3f' % v for v in per_step]}")
    assert max(per_step) <= FAST_TOL_MM            # does not compound over steps


@pytest.mark.parametrize("frames,B,H,K", [(27, 2, 2, 2), (243, 1, 1, 1)])
def test_sampler_fast_mode_vs_bf16_emulating_oracle(frames, B, H, K):
    """FAST mode end to end against the oracle run with a bf16 rounding wherever the FAST kernels store or consume
    bf16 (oracle.emulate_bf16).  A TIGHT gate against that emulation does not exist: on these randomly initialised
    weights the bf16 pipeline is chaotic in its roundings -- two runs of the SAME emulation that differ only in the fp32
    accumulation order of their matmuls (K split in halves) end up 0.8 mm apart after 2 blocks and 3.2 mm after 16
    (measured), as far from each other as each is from the fp32 oracle.  What is gated instead, adaptively
    on the same inputs: the kernels' distance to the fp32 oracle must be bf16-CLASS (<= 1.5x the emulation's own
    distance), and their distance to the emulation must not exceed 1.5x the distance between two accumulation orders of
    the emulation.  A glue bug worth several millimetres fails both."""
    import torch.nn.functional as F
    sd = make_state_dict(7, 512, 8, frames)
    x2d = synthetic_inputs_2d(81, B, frames)
    noises = [torch.from_numpy(synthetic_noise(90 + k, (B, H, frames, 17, 3))) for k in range(K)]
    p = orc.strip_prefix(sd)
    run = lambda q: orc.ddim_sample_flip(q, orc.cosine_schedule(1000), torch.from_numpy(x2d), torch.from_numpy(flip_2d(x2d)),
                                         H, K, 8, H36M_JOINTS_LEFT, H36M_JOINTS_RIGHT, noises)
    want32, want16 = run(p), run(orc.emulate_bf16(p))
    orig = F.linear

    def halves(x, w, b=None):            # the same products, accumulated as two half-K sums
        k = w.shape[1]
        if k < 64:
            return orig(x, w, b)
        y = orig(x[..., :k // 2], w[:, :k // 2]) + orig(x[..., k // 2:], w[:, k // 2:])
        return y if b is None else y + b

    F.linear = halves
    try:
        want16b = run(orc.emulate_bf16(p))
    finally:
        F.linear = orig
    m = make_model(frames, 512, 8, H, K, "fast", 7)
    out = m(torch.from_numpy(x2d).cuda(), None, input_2d_flip=torch.from_numpy(flip_2d(x2d)).cuda(), noise=noises).cpu()
    e16, e32 = orc.mpjpe_mm(out, want16), orc.mpjpe_mm(out, want32)
    emu32, emu_self = orc.mpjpe_mm(want16, want32), orc.mpjpe_mm(want16, want16b)
    print(f"fast F={frames}: kernels vs fp32 oracle {e32:.3f} mm (bf16 emulation vs fp32 oracle {emu32:.3f} mm); kernels vs "
          f"emulation {e16:.3f} mm (two accumulation orders of the emulation {emu_self:.3f} mm apart)")
    assert e32 <= 1.5 * emu32 and e32 <= FAST_TOL_MM
    assert e16 <= 1.5 * max(emu_self, 0.5 * emu32)


# ------------------------------------------------------------------------------------------------ BASELINE configs at full size
def test_c2_full_size_vs_reference_fixture(golden_dir):
    """BASELINE configs[1]: F=243, J=17, H=5, K=5, B=4, exact mode, against the REFERENCE run at full size
    (fixture g13: every 10th frame in full + fp64 checksums of every (clip, step, hypothesis) over all frames)."""
    g = load_g(golden_dir, "g13_sampler_c2")
    cs, dep, Fr, B, H, K = (int(g[k]) for k in ("cs", "dep", "frames", "B", "H", "K"))
    assert (Fr, B, H, K) == (243, 4, 5, 5)
    x2d = synthetic_inputs_2d(int(g["x2d_seed"]), B, Fr)
    noises = [torch.from_numpy(synthetic_noise(int(g["noise_seed"]) + k, (B, H, Fr, 17, 3))) for k in range(K)]
    m = make_model(Fr, cs, dep, H, K, "exact", int(g["seed"]))
    out = m(torch.from_numpy(x2d).cuda(), None, input_2d_flip=torch.from_numpy(flip_2d(x2d)).cuda(), noise=noises).cpu()
    assert out.shape == (B, K, H, Fr, 17, 3)
    kept = torch.from_numpy(g["kept_frames"]).long()
    per_step = [orc.mpjpe_mm(out[:, k][:, :, kept], torch.from_numpy(g["out_kept"][:, k])) for k in range(K)]
    print(f"[c2 full size] exact MPJPE per step on {len(kept)} of {Fr} frames (mm): {['%.2e' % v for v in per_step]}")
    assert max(per_step) <= EXACT_TOL_MM
    # all frames through the checksums: mean signed / weighted deviation per coordinate far below the tolerance
    o = out.double().reshape(B, K, H, -1)
    n = o.shape[-1]
    w = torch.cos(torch.arange(n, dtype=torch.float64) * 0.37) + 1.5
    d_sum = (o.sum(-1) - torch.from_numpy(g["sum"])).abs().max().item() / n
    d_w = ((o * w).sum(-1) - torch.from_numpy(g["wsum"])).abs().max().item() / n
    d_sq = ((o * o).sum(-1) - torch.from_numpy(g["sumsq"])).abs().max().item() / n
    print(f"[c2 full size] checksum deviations per coordinate (m): sum {d_sum:.2e} weighted {d_w:.2e} squares {d_sq:.2e}")
    assert d_sum < 2e-7 and d_w < 3e-7 and d_sq < 2e-7        # 1e-3 mm = 1e-6 m per joint; these are means of signed errors


def test_c3_full_size_slices_vs_oracle():
    """BASELINE configs[2]: F=243, H=20, K=10, B=16 -- the benchmarked workload -- in exact mode.  The full run is
    checked for its size-independent properties and two (clip, hypothesis) slices are compared with the CPU oracle run
    on that slice alone (K=10 steps x 2 flip passes each): hypotheses and clips are independent given the 2D input,
    which test_full_size_properties proves bit-exactly for this library."""
    Fr, B, H, K = 243, 16, 20, 10
    x2d = synthetic_inputs_2d(1234, B, Fr)
    x2f = flip_2d(x2d)
    noises = [torch.from_numpy(synthetic_noise(2000 + k, (B, H, Fr, 17, 3))) for k in range(K)]
    m = make_model(Fr, 512, 8, H, K, "exact", 7)
    out = m(torch.from_numpy(x2d).cuda(), None, input_2d_flip=torch.from_numpy(x2f).cuda(), noise=noises).cpu()
    assert out.shape == (B, K, H, Fr, 17, 3) and torch.isfinite(out).all() and out.abs().max().item() <= 1.1 * (1 + 1e-6)
    p = orc.strip_prefix(make_state_dict(7, 512, 8, Fr))
    sched = orc.cosine_schedule(1000)
    for b, h in ((3, 7), (15, 19)):
        want = orc.ddim_sample_flip(p, sched, torch.from_numpy(x2d[b:b + 1]), torch.from_numpy(x2f[b:b + 1]), 1, K, 8,
                                    H36M_JOINTS_LEFT, H36M_JOINTS_RIGHT, [n[b:b + 1, h:h + 1] for n in noises])
        per_step = [orc.mpjpe_mm(out[b:b + 1, k, h:h + 1], want[:, k]) for k in range(K)]
        print(f"[c3 full size] slice (clip {b}, hypothesis {h}) exact MPJPE per step (mm): {['%.2e' % v for v in per_step]}")
        assert max(per_step) <= EXACT_TOL_MM


def test_c3_full_size_all_slices_vs_reference_fixture(golden_dir):
    """BASELINE configs[2] -- the benchmarked workload, F=243 H=20 K=10 B=16 -- against the REFERENCE run at full size,
    ALL 3200 (clip, step, hypothesis) slices (the test above compares 2 of 320 trajectories).  Fixture g14
    (tools/make_goldens.py: the reference, clip by clip, ~2.5 h of host CPU) keeps per slice the fp64 sum, sum of squares
    and four Gaussian random projections w_p . x of the 12,393 output coordinates.  For an error vector e of a slice,
    E[(w_p . e)^2] = |e|^2, so the projections measure RMS error, and by Jensen's inequality
        MPJPE = mean_j |e_j|  <=  sqrt(mean_j |e_j|^2) = sqrt(3) * rms_coordinate(e),
    i.e. sqrt(3) * rms is an UPPER bound of the slice's MPJPE.  Gates: the bound pooled over the 320 slices of every DDIM
    step (1280 projections each: a tight estimate) <= 1e-3 mm; every single slice's own 4-projection estimate <= 4e-3 mm
    (chi-square with 4 degrees of freedom: a slice at the pooled level exceeds 2.2x it with probability 1e-4)."""
    if not os.path.exists(os.path.join(golden_dir, "g14_sampler_c3.npz")):
        pytest.skip("fixture g14 (2.5 h of reference CPU time, tools/make_goldens.py --only g14) not generated")
    g = load_g(golden_dir, "g14_sampler_c3")
    cs, dep, Fr, B, H, K = (int(g[k]) for k in ("cs", "dep", "frames", "B", "H", "K"))
    assert (Fr, B, H, K) == (243, 16, 20, 10)
    x2d = synthetic_inputs_2d(int(g["x2d_seed"]), B, Fr)
    noises = [torch.from_numpy(synthetic_noise(int(g["noise_seed"]) + k, (B, H, Fr, 17, 3))) for k in range(K)]
    m = make_model(Fr, cs, dep, H, K, "exact", int(g["seed"]))
    out = m(torch.from_numpy(x2d).cuda(), None, input_2d_flip=torch.from_numpy(flip_2d(x2d)).cuda(), noise=noises)
    assert out.shape == (B, K, H, Fr, 17, 3) and torch.isfinite(out).all()
    n = Fr * 17 * 3
    o = out.double().reshape(B, K, H, n)                                      # on the GPU: 40 M doubles
    w = torch.from_numpy(np.random.Generator(np.random.PCG64(int(g["proj_seed"]))).standard_normal(size=(4, n))).cuda()
    d_proj = (o @ w.t()).cpu() - torch.from_numpy(g["proj"])                  # (B, K, H, 4) = w_p . e
    d_sum = (o.sum(-1).cpu() - torch.from_numpy(g["sum"])).abs().max().item() / n
    d_sq = ((o * o).sum(-1).cpu() - torch.from_numpy(g["sumsq"])).abs().max().item() / n
    rms_slice = (d_proj ** 2).mean(-1).div(n).sqrt()                          # per-slice RMS coordinate error (m), 4 samples
    rms_step = (d_proj ** 2).mean(dim=(0, 2, 3)).div(n).sqrt()                # pooled per DDIM step, 1280 samples
    bound_step_mm = (3 ** 0.5) * rms_step * 1e3
    worst_slice_mm = (3 ** 0.5) * rms_slice.max().item() * 1e3
    print(f"[c3 all slices] MPJPE upper bound per step (mm): {['%.2e' % v for v in bound_step_mm.tolist()]}; worst single "
          f"slice estimate {worst_slice_mm:.2e} mm; checksum deviations per coordinate (m): sum {d_sum:.2e} squares {d_sq:.2e}")
    assert bound_step_mm.max().item() <= EXACT_TOL_MM
    assert worst_slice_mm <= 4 * EXACT_TOL_MM
    assert d_sum < 2e-7 and d_sq < 2e-7


@pytest.mark.parametrize("numerics", ["exact", "fast"])
def test_sampler_scale_and_live_oracle(numerics):
    """scale != 1 exercises the clamp/scale arithmetic; oracle computed live on the host CPU."""
    Fr, B, H, K, cs, dep, scale = 27, 2, 2, 3, 512, 2, 2.5
    sd = make_state_dict(21, cs, dep, Fr)
    x2d = synthetic_inputs_2d(31, B, Fr)
    noises = [torch.from_numpy(synthetic_noise(40 + k, (B, H, Fr, 17, 3))) * 1.5 for k in range(K)]
    want = orc.ddim_sample_flip(orc.strip_prefix(sd), orc.cosine_schedule(1000), torch.from_numpy(x2d),
                                torch.from_numpy(flip_2d(x2d)), H, K, dep, H36M_JOINTS_LEFT, H36M_JOINTS_RIGHT, noises,
                                scale=scale)
    m = make_model(Fr, cs, dep, H, K, numerics, 21, scale=scale)
    out = m(torch.from_numpy(x2d).cuda(), None, input_2d_flip=torch.from_numpy(flip_2d(x2d)).cuda(), noise=noises)
    err = orc.mpjpe_mm(out.cpu(), want)
    print(f"[{numerics}] scale={scale}: MPJPE vs oracle {err:.3e} mm")
    assert err <= (EXACT_TOL_MM if numerics == "exact" else FAST_TOL_MM)
    assert out.abs().max().item() <= 1.1 * scale * (1 + 1e-6)


@pytest.mark.parametrize("numerics", ["exact", "fast"])
def test_full_size_properties(numerics):
    """Size-independent properties at F=243 (BASELINE config-2 shape, reduced K): finite, clamped, bit-exact
    flip equivariance (the TTA average is symmetric in its two branches), bit-exact invariance to the internal
    chunking, and determinism."""
    Fr, B, H, K = 243, 2, 3, 2
    x2d = synthetic_inputs_2d(51, B, Fr)
    x2f = flip_2d(x2d)
    noises = [torch.from_numpy(synthetic_noise(60 + k, (B, H, Fr, 17, 3))) for k in range(K)]
    m = make_model(Fr, 512, 8, H, K, numerics, 7, chunk_seqs=5)
    a = m(torch.from_numpy(x2d).cuda(), None, input_2d_flip=torch.from_numpy(x2f).cuda(), noise=noises)
    assert torch.isfinite(a).all() and a.abs().max().item() <= 1.1
    a2 = m(torch.from_numpy(x2d).cuda(), None, input_2d_flip=torch.from_numpy(x2f).cuda(), noise=noises)
    assert torch.equal(a, a2)
    # flipped problem: swap the roles of the two 2D inputs and flip every noise draw
    fn = [orc.flip_pose(n, H36M_JOINTS_LEFT, H36M_JOINTS_RIGHT) for n in noises]
    b = m(torch.from_numpy(x2f).cuda(), None, input_2d_flip=torch.from_numpy(x2d).cuda(), noise=fn)
    assert torch.equal(orc.flip_pose(b.cpu(), H36M_JOINTS_LEFT, H36M_JOINTS_RIGHT), a.cpu())
    m.pose_estimator.set_numerics(numerics, chunk_seqs=1)
    c = m(torch.from_numpy(x2d).cuda(), None, input_2d_flip=torch.from_numpy(x2f).cuda(), noise=noises)
    assert torch.equal(a, c)
    # hypotheses are independent given the 2D input: sampling h=1 alone reproduces slice h=1 (the H-sharding contract)
    m1 = make_model(Fr, 512, 8, 1, K, numerics, 7)
    d = m1(torch.from_numpy(x2d).cuda(), None, input_2d_flip=torch.from_numpy(x2f).cuda(),
           noise=[n[:, 1:2].contiguous() for n in noises])
    assert torch.equal(d[:, :, 0], a[:, :, 1])


@pytest.mark.parametrize("impl", ["f32", "bf16x3"])
def test_exact_mode_cross_check_implementations(golden_dir, monkeypatch, impl):
    """env D3DP_EXACT_IMPL keeps the plain fp32-MFMA Linears and the six-pass split-bf16 Linears alive as cross-checks
    of the split-fp16 default, at the same tolerance."""
    monkeypatch.setenv("D3DP_EXACT_IMPL", impl)
    g = load_g(golden_dir, "g3_denoiser_F27")
    x2d = torch.from_numpy(synthetic_inputs_2d(int(g["x2d_seed"]), 1, 27)).cuda()
    x3d = torch.from_numpy(synthetic_noise(int(g["x3d_seed"]), (1, 1, 27, 17, 3))).cuda()
    m = make_model(27, 512, 8, 1, 1, "exact", int(g["seed"]))
    e = orc.mpjpe_mm(m.pose_estimator(x2d, x3d, torch.tensor([999], device="cuda")).cpu(), torch.from_numpy(g["out_t999"]))
    assert e <= EXACT_TOL_MM


def test_exact_residual_adds_inside_the_linears_change_no_bit(golden_dir, monkeypatch):
    """EXACT mode adds proj / fc2 into the residual stream inside the Linear epilogues; env D3DP_NO_FOLD=1 keeps the earlier
    dataflow (separate y1 / y buffers, added by the row kernels).  Same fp32 additions in the same order: bit-identical."""
    g = load_g(golden_dir, "g3_denoiser_F27")
    x2d = torch.from_numpy(synthetic_inputs_2d(int(g["x2d_seed"]), 1, 27)).cuda()
    x3d = torch.from_numpy(synthetic_noise(int(g["x3d_seed"]), (1, 1, 27, 17, 3))).cuda()
    t = torch.tensor([499], device="cuda")
    folded = make_model(27, 512, 8, 1, 1, "exact", int(g["seed"])).pose_estimator(x2d, x3d, t).clone()
    monkeypatch.setenv("D3DP_NO_FOLD", "1")
    plain = make_model(27, 512, 8, 1, 1, "exact", int(g["seed"])).pose_estimator(x2d, x3d, t)
    assert torch.equal(folded, plain)
    assert orc.mpjpe_mm(folded.cpu(), torch.from_numpy(g["out_t499"])) <= EXACT_TOL_MM


@pytest.mark.variants
@pytest.mark.parametrize("frames", [27, 243])
def test_exact_norm2_folded_into_the_linears(golden_dir, monkeypatch, frames):
    """D3DP_FOLD_LN=1 (built, measured and left off: no faster, capi.hip fold_ln): norm2 (mixste.py:115) without
    a kernel of its own -- proj's epilogue leaves x + proj(...) as fc1's UN-normalised split-fp16 operand with the row
    statistics in 64-column pieces, and fc1 = LN folded into the Linear (W diag(gamma), rstd (. - mean c1) + c2 in the
    epilogue).  Against the reference goldens at three timesteps, and against the default (norm2 as a row kernel): the two
    forms differ by rounding only and neither is systematically closer to the reference."""
    g = load_g(golden_dir, f"g3_denoiser_F{frames}")
    x2d = torch.from_numpy(synthetic_inputs_2d(int(g["x2d_seed"]), 1, frames)).cuda()
    x3d = torch.from_numpy(synthetic_noise(int(g["x3d_seed"]), (1, 1, frames, 17, 3))).cuda()
    outs = {}
    for tag in ("kernel", "folded"):
        if tag == "folded":
            monkeypatch.setenv("D3DP_FOLD_LN", "1")
        m = make_model(frames, 512, 8, 1, 1, "exact", int(g["seed"]))
        outs[tag] = {tt: m.pose_estimator(x2d, x3d, torch.tensor([tt], device="cuda")).cpu() for tt in (999, 499, 99)}
        assert not m.pose_estimator.nonfinite_seen()
    for tt in (999, 499, 99):
        ref = torch.from_numpy(g[f"out_t{tt}"])
        ef, ek = orc.mpjpe_mm(outs["folded"][tt], ref), orc.mpjpe_mm(outs["kernel"][tt], ref)
        d = orc.mpjpe_mm(outs["folded"][tt], outs["kernel"][tt])
        print(f"F={frames} t={tt}: folded norm2 vs reference {ef:.3e} mm, norm2 kernel vs reference {ek:.3e} mm, apart {d:.3e} mm")
        assert ef <= EXACT_TOL_MM and ek <= EXACT_TOL_MM and d <= 0.5 * EXACT_TOL_MM


def test_deferred_backward_recomputes_its_own_forward(golden_dir):
    """Two forwards before the first backward share one activation workspace: the first backward must differentiate
    through ITS forward (it re-runs it), as plain autograd does in the reference."""
    g = load_g(golden_dir, "g6_train_step")
    cs, dep, Fr = int(g["cs"]), int(g["dep"]), int(g["frames"])
    args = SimpleNamespace(number_of_frames=Fr, test_time_augmentation=True, timestep=1000, scale=1.0, cs=cs, dep=dep,
                           numerics="fast")                       # the train model ignores the inference numerics
    x2d, gt, t = (torch.from_numpy(g[k]).cuda() for k in ("x2d", "gt", "t"))
    noise = torch.from_numpy(g["noise"]).cuda()

    def grads(defer):
        m = D3DP(args, H36M_JOINTS_LEFT, H36M_JOINTS_RIGHT, is_train=True)
        assert m.pose_estimator.numerics == "train"
        m.load_state_dict(make_state_dict(int(g["seed"]), cs, dep, Fr), strict=False)
        m = m.cuda().eval()                                        # eval(): DropPath off, still differentiable
        l1 = torch.mean(torch.norm(m(x2d, gt, t=t[:, None], noise=noise) - gt, dim=-1))
        if defer:                                                  # a second forward overwrites the workspace
            l2 = torch.mean(torch.norm(m(x2d.flip(0), gt.flip(0), t=t.flip(0)[:, None], noise=noise.flip(0)) - gt.flip(0), dim=-1))
        l1.backward()
        return [p.grad.clone() for p in m.parameters()]

    a, b = grads(False), grads(True)
    # (gradients taken through the WRONG forward's activations would differ by O(1); the re-run forward reproduces the first
    #  one's activations exactly and no gradient is summed with float atomics any more: bit-equal)
    for x, y in zip(a, b):
        assert torch.equal(x, y)


def test_backward_after_optimizer_step_raises(golden_dir):
    """The training kernels read the parameters' own storage (borrowed weights): a backward() issued after the weights
    changed would differentiate at the NEW weights.  PyTorch raises in the equivalent situation; so do we."""
    g = load_g(golden_dir, "g6_train_step")
    cs, dep, Fr = int(g["cs"]), int(g["dep"]), int(g["frames"])
    args = SimpleNamespace(number_of_frames=Fr, test_time_augmentation=True, timestep=1000, scale=1.0, cs=cs, dep=dep)
    x2d, gt, t = (torch.from_numpy(g[k]).cuda() for k in ("x2d", "gt", "t"))
    noise = torch.from_numpy(g["noise"]).cuda()
    m = D3DP(args, H36M_JOINTS_LEFT, H36M_JOINTS_RIGHT, is_train=True)
    m.load_state_dict(make_state_dict(int(g["seed"]), cs, dep, Fr), strict=False)
    m = m.cuda().eval()
    loss = torch.mean(torch.norm(m(x2d, gt, t=t[:, None], noise=noise) - gt, dim=-1))
    with torch.no_grad():
        next(m.pose_estimator.parameters()).add_(1e-3)             # what optimizer.step() does
    with pytest.raises(RuntimeError, match="modified"):
        loss.backward()


def test_exact_mode_range_guard(monkeypatch):
    """EXACT mode must not answer NaN where the fp32 reference is finite.  The split-fp16 operands hold
    |x| 2^s < 65504; the library proves the range of every data-dependent operand from the weights (d3dp_exact_range_bound)
    and lowers the scale s of the blocks that need it (d3dp_exact_scales) -- no warning, no environment variable.
    (1) ordinary weights: every scale 2^4; (2) one fc1 matrix x 50: the proof exceeds 4094, that block's hidden scale drops,
    the result stays within the exact tolerance of the fp32 oracle; (3) x 5000: activations of several thousand, still finite
    and within the oracle's own fp32 noise for such magnitudes; (4) a proj bias of +5000 only moves the fp32 residual
    stream: nothing to scale; (5) a LayerNorm gain of 300 pushes LayerNorm's OWN output bound out of range: the context falls
    back to the six-pass split-bf16 implementation by itself; (6) the run-time check of the D3DP_FOLD_LN measurement switch
    (the one operand outside the proof) still reports through d3dp_status."""
    Fr, B, H, K, cs, dep = 9, 2, 2, 1, 512, 2
    x2d = synthetic_inputs_2d(5, B, Fr)
    x2f = flip_2d(x2d)
    noises = [torch.from_numpy(synthetic_noise(70, (B, H, Fr, 17, 3)))]

    def run(edit):
        sd = make_state_dict(7, cs, dep, Fr)
        edit(sd)
        args = SimpleNamespace(number_of_frames=Fr, test_time_augmentation=True, timestep=1000, scale=1.0, cs=cs, dep=dep)
        m = D3DP(args, H36M_JOINTS_LEFT, H36M_JOINTS_RIGHT, is_train=False, num_proposals=H, sampling_timesteps=K,
                 numerics="exact")
        m.load_state_dict(sd, strict=False)
        m = m.cuda().eval()
        out = m(torch.from_numpy(x2d).cuda(), None, input_2d_flip=torch.from_numpy(x2f).cuda(), noise=noises)
        want = orc.ddim_sample_flip(orc.strip_prefix(sd), orc.cosine_schedule(1000), torch.from_numpy(x2d),
                                    torch.from_numpy(x2f), H, K, dep, H36M_JOINTS_LEFT, H36M_JOINTS_RIGHT, noises)
        return m.pose_estimator, out, orc.mpjpe_mm(out.cpu(), want)

    def scaled(suffix, factor=1.0, add=0.0):
        def edit(sd):
            key = [k for k in sd if k.endswith(suffix)][0]
            sd[key] = sd[key] * factor + add
        return edit

    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("error")                             # no warning in any of the cases
        net, out, e = run(lambda sd: None)                         # (1)
        b = net.exact_range_bound()
        kv, hd, impl = net.exact_scales()
        print(f"provable operand bound with seed-generated weights: {b:.1f} (2^4 holds {net.SPLIT_RANGE}); error {e:.2e} mm")
        assert 1.0 < b < net.SPLIT_RANGE and impl == "f16x2" and set(kv) == {16.0} and set(hd) == {16.0}
        assert torch.isfinite(out).all() and not net.nonfinite_seen() and e <= EXACT_TOL_MM
        for factor, tol in ((50.0, EXACT_TOL_MM), (5000.0, 0.05)):   # (2), (3); block STE 1 is index 1 of the scale lists
            net, out, e = run(scaled("STEblocks.1.mlp.fc1.weight", factor))
            kv, hd, impl = net.exact_scales()
            print(f"fc1 x {factor:g}: bound {net.exact_range_bound():.4g}, hidden scale of that block {hd[1]:g}, error {e:.3e} mm")
            assert net.exact_range_bound() >= net.SPLIT_RANGE and impl == "f16x2"
            assert hd[1] < 16.0 and hd[1] * net.exact_range_bound() < 65504.0 and [h for i, h in enumerate(hd) if i != 1] == [16.0] * 3
            assert set(kv) == {16.0}
            assert torch.isfinite(out).all() and not net.nonfinite_seen() and e <= tol
        net, out, e = run(scaled("STEblocks.1.attn.qkv.weight", 200.0))   # q, k, v of several hundred: s_kv drops
        kv, hd, impl = net.exact_scales()
        print(f"qkv x 200: bound {net.exact_range_bound():.4g}, q/k/v scale of that block {kv[1]:g}, error {e:.3e} mm")
        assert kv[1] < 16.0 and set(hd) == {16.0} and impl == "f16x2"
        assert torch.isfinite(out).all() and not net.nonfinite_seen() and e <= 0.05
        net, out, e = run(scaled("STEblocks.1.attn.proj.bias", add=5000.0))   # (4)
        assert net.exact_range_bound() < net.SPLIT_RANGE and torch.isfinite(out).all() and not net.nonfinite_seen()
        print(f"proj bias + 5000: error {e:.3e} mm")
        assert e <= 0.5         # (x ~ 5000 in front of norm2: the fp32 oracle's own LayerNorm cancels 5000 - 5000 at an ulp of 5e-4)
        net, out, e = run(scaled("TTEblocks.0.norm2.weight", 300.0))         # (5)
        kv, hd, impl = net.exact_scales()
        print(f"norm2 gain x 300: implementation {impl}, error {e:.3e} mm")
        assert impl == "bf16x3" and torch.isfinite(out).all() and not net.nonfinite_seen() and e <= 0.05
    # (6) the un-normalised residual operand proj hands to fc1 under D3DP_FOLD_LN=1 is outside the static proof: checked at run
    # time.  The switch selects an experiment epilogue: a library built without the variants REFUSES it (never ignores it).
    monkeypatch.setenv("D3DP_FOLD_LN", "1")
    if _lib.load().d3dp_debug_x2_variants() == 1:
        net, out, _ = run(scaled("STEblocks.1.attn.proj.bias", add=5000.0))
        assert net.exact_range_bound() < net.SPLIT_RANGE and net.nonfinite_seen()
        assert not net.nonfinite_seen()                            # the query resets the flag
    else:
        with pytest.raises(_lib.D3DPHipError, match="built without"):
            run(scaled("STEblocks.1.attn.proj.bias", add=5000.0))
    monkeypatch.delenv("D3DP_FOLD_LN")


@pytest.mark.parametrize("cs,dep", [(512, 2), (128, 4)])
def test_sampler_parity_on_weights_that_went_through_training(cs, dep):
    """VERDICT r5 missing 2: no trained checkpoint exists here (README.md:33-39: h36m_best_epoch.bin), so every tolerance is known on
    seed-generated weights.  The closest stand-in this box can produce: the SAME model trained for 60 AdamW steps at a large learning
    rate through the library's own training step on a fixed synthetic regression target (the weights move by tens of per cent:
    LayerNorm gains drift from 1, matrices lose their uniform distribution, the position embeddings grow), then the flip-TTA
    sampler on those weights against the oracle on the same weights at EXACT mode's tolerance -- with the proven operand range and
    the operand scales d3dp_set_weights derives from the NEW weights reported."""
    from d3dp_amd.optim import HipAdamW
    Fr, B, H, K = 27, 2, 2, 2
    args = SimpleNamespace(number_of_frames=Fr, test_time_augmentation=True, timestep=1000, scale=1.0, cs=cs, dep=dep)
    sd0 = make_state_dict(53, cs, dep, Fr)
    mt = D3DP(args, H36M_JOINTS_LEFT, H36M_JOINTS_RIGHT, is_train=True)
    mt.load_state_dict(sd0, strict=False)
    mt = mt.cuda().train()
    opt = HipAdamW(mt.parameters(), lr=2e-3, weight_decay=0.1)
    torch.manual_seed(78)                                    # (the training branch draws its timesteps, noise and DropPath masks from the global generators)
    g = torch.Generator().manual_seed(77)
    x2 = (torch.rand(8, Fr, 17, 2, generator=g) * 2 - 1).cuda()
    gt = (torch.randn(8, Fr, 17, 3, generator=g) * 0.4).cuda()
    gt[:, :, 0] = 0
    first = last = None
    for it in range(60):
        idx = torch.randint(0, 8, (4,), generator=g)
        opt.zero_grad(set_to_none=True)
        pred = mt(x2[idx], gt[idx])
        loss = torch.mean(torch.norm(pred - gt[idx], dim=-1))
        loss.backward(loss.clone().detach())
        opt.step()
        first = loss.item() if first is None else first
        last = loss.item()
    assert last < 0.8 * first, (first, last)               # it did train
    sd = {k: v.detach().cpu().clone() for k, v in mt.state_dict().items()}
    moved = max(((sd[k] - sd0[k]).norm() / sd0[k].norm().clamp_min(1e-12)).item() for k in sd0 if k.endswith("qkv.weight"))
    gain = max((sd[k] - 1).abs().max().item() for k in sd if k.endswith("norm1.weight"))
    x2d = synthetic_inputs_2d(531, B, Fr)
    noises = [torch.from_numpy(synthetic_noise(532 + k, (B, H, Fr, 17, 3))) for k in range(K)]
    want = orc.ddim_sample_flip(orc.strip_prefix(sd), orc.cosine_schedule(1000), torch.from_numpy(x2d), torch.from_numpy(flip_2d(x2d)),
                                H, K, dep, H36M_JOINTS_LEFT, H36M_JOINTS_RIGHT, noises)
    m = D3DP(args, H36M_JOINTS_LEFT, H36M_JOINTS_RIGHT, is_train=False, num_proposals=H, sampling_timesteps=K, numerics="exact")
    m.load_state_dict(sd, strict=False)
    m = m.cuda().eval()
    out = m(torch.from_numpy(x2d).cuda(), None, input_2d_flip=torch.from_numpy(flip_2d(x2d)).cuda(), noise=noises)
    err = orc.mpjpe_mm(out.cpu(), want)
    kv, hid, impl = m.pose_estimator.exact_scales()
    print(f"cs={cs} dep={dep} after 60 AdamW steps (loss {first:.3f} -> {last:.3f}; qkv weights moved by up to {100 * moved:.0f} %, norm1 gains by up to "
          f"{gain:.2f}): exact MPJPE vs the oracle on the trained weights {err:.3e} mm; proven operand range {m.pose_estimator.exact_range_bound():.1f}, "
          f"scales {min(kv):g} .. {max(kv):g} / {min(hid):g} .. {max(hid):g}, implementation {impl}")
    assert err <= EXACT_TOL_MM and torch.isfinite(out).all() and not m.pose_estimator.nonfinite_seen()


def test_ddim_sample_no_flip_runs():
    m = make_model(27, 512, 2, 2, 2, "exact", 21)
    m.flip = False
    x2d = torch.from_numpy(synthetic_inputs_2d(71, 2, 27)).cuda()
    out = m(x2d, None)
    assert isinstance(out, list) and len(out) == 2 and out[0].shape == (2, 2, 27, 17, 3)


def test_profile_counters():
    m = make_model(27, 512, 2, 1, 1, "fast", 21)
    x2d = torch.from_numpy(synthetic_inputs_2d(71, 1, 27)).cuda()
    m(x2d, None, input_2d_flip=x2d)
    m.pose_estimator.profile_enable(True)
    m(x2d, None, input_2d_flip=x2d)
    prof = m.pose_estimator.profile_read()
    m.pose_estimator.profile_enable(False)
    assert prof["gemm_qkv"][0] == 4 and prof["gemm_fc2"][0] == 4 and prof["head"][0] == 1
    assert all(ms >= 0 for _, ms in prof.values()) and prof["gemm_qkv"][1] > 0


def test_jpma_kernel_matches_reference_metrics(golden_dir):
    """d3dp_jpma (fused root zeroing + trajectory + projection + per-joint argmin + gather) against fixture g5, i.e.
    the reference's own camera.project_to_2d + loss.mpjpe_diffusion_reproj / mpjpe_diffusion_all_min."""
    from d3dp_amd import jpma
    g = load_g(golden_dir, "g5_caller")
    Fr = int(g["frames"])
    x2, x3 = jpma.eval_data_prepare(Fr, torch.from_numpy(g["seq2d"])[None], torch.from_numpy(g["seq3d"])[None])
    traj = x3[:, :, :1].clone()
    x3[:, :, 0] = 0
    pred = torch.from_numpy(g["pred"])                       # root joint NOT yet zeroed: the kernel does it
    agg, sel, es, em = jpma.jpma_hip(pred.cuda(), traj.cuda(), torch.from_numpy(g["cam"]).cuda(), x2.cuda(), x3.cuda(),
                                     zero_root=True, want_errors=True)
    K = pred.shape[1]
    j_agg = es.permute(1, 0, 2, 3).reshape(K, -1).mean(-1).cpu()
    j_best = em.permute(1, 0, 2, 3).reshape(K, -1).mean(-1).cpu()
    assert torch.allclose(j_agg, torch.from_numpy(g["e_jagg"]), atol=1e-6, rtol=1e-5)
    assert torch.allclose(j_best, torch.from_numpy(g["e_jbest"]), atol=1e-6, rtol=1e-5)
    pz = pred.clone()
    pz[:, :, :, :, 0] = 0
    want = jpma.jpma_aggregate(pz, torch.from_numpy(g["reproj"]), x2)
    assert torch.equal(agg.cpu(), want)


def test_training_step_config5_vs_oracle_autograd():
    """BASELINE config 5 at full size: F=243, J=17, H=1, batch=4, cs=512, dep=8, q_sample + MixSTE2 fwd/bwd, MPJPE loss,
    DropPath active (recorded masks).  Loss and EVERY parameter gradient against torch autograd through the CPU oracle."""
    import time
    Fr, B, cs, dep = 243, 4, 512, 8
    args = SimpleNamespace(number_of_frames=Fr, test_time_augmentation=True, timestep=1000, scale=1.0, cs=cs, dep=dep)
    sd = make_state_dict(7, cs, dep, Fr)
    m = D3DP(args, H36M_JOINTS_LEFT, H36M_JOINTS_RIGHT, is_train=True)
    m.load_state_dict(sd, strict=False)
    m = m.cuda().train()
    x2d = torch.from_numpy(synthetic_inputs_2d(901, B, Fr))
    gt = torch.from_numpy(synthetic_noise(902, (B, Fr, 17, 3))) * 0.3
    gt[:, :, 0] = 0
    t = torch.tensor([[3], [250], [640], [999]], dtype=torch.long)
    noise = torch.from_numpy(synthetic_noise(903, (B, Fr, 17, 3)))
    rates = [x.item() for x in torch.linspace(0, 0.1, dep)]
    gmask = torch.Generator().manual_seed(5)
    dpd = {}
    for i in range(1, dep):
        keep = 1 - rates[i]
        mk = lambda S: (torch.rand(S, 1, 1, generator=gmask) < keep).float() / keep
        dpd[f"STEblocks.{i}"] = (mk(B * Fr), mk(B * Fr))
        dpd[f"TTEblocks.{i}"] = (mk(B * 17), mk(B * 17))
    def step():
        m.zero_grad(set_to_none=True)
        pred = m(x2d.cuda(), gt.cuda(), t=t, noise=noise, droppath=dpd)
        loss = torch.mean(torch.norm(pred - gt.cuda(), dim=-1))
        loss.backward(loss.clone().detach())
        return pred, loss
    pred, loss = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pred, loss = step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"config-5 training step (fwd+bwd, {m.pose_estimator.train_arithmetic()}): {dt * 1e3:.1f} ms = {3 * 4 * 294.86e9 / dt / 1e12:.1f} TFLOP/s")
    po = {k: v.clone().requires_grad_(True) for k, v in orc.strip_prefix(sd).items()}
    xp = orc.prepare_targets(orc.cosine_schedule(1000), gt, t[:, 0], noise)
    pred_o = orc.mixste_forward(po, x2d, xp, t[:, 0], dep, droppath=dpd)
    loss_o = torch.mean(torch.norm(pred_o - gt, dim=-1))
    loss_o.backward(loss_o.clone().detach())
    assert orc.mpjpe_mm(pred.detach().cpu(), pred_o.detach()) <= EXACT_TOL_MM
    assert abs(loss.item() - loss_o.item()) < 2e-6
    worst = ("", 0.0)
    for name, p in m.pose_estimator.named_parameters():
        ref = po[name].grad.double()
        err = (p.grad.cpu().double() - ref).norm().item() / max(ref.norm().item(), 1e-12)
        if err > worst[1]:
            worst = (name, err)
        assert err < 5e-3, (name, err)
    print(f"config-5: worst relative gradient error over {len(po)} parameters: {worst[1]:.2e} ({worst[0]})")


@pytest.mark.parametrize("Fr", [9, 40, 81, 243])
def test_attention_backward_on_matrix_cores_matches_the_valu_kernels(monkeypatch, Fr):
    """Head-dim-64 attention of the training step, three implementations of the same arithmetic:
      x2   (default)             -- BOTH axes forward and backward on split-fp16 operands (train_attn.hip: three fp16-MFMA passes
                                    per product, device-side operand scales, running power-of-two scale for dS);
      x2t  (D3DP_TRAIN_ATTN=x2t) -- the temporal axis on those kernels, the spatial axis as below;
      mfma (D3DP_TRAIN_ATTN=f32) -- the round-4 kernels: fp32-MFMA temporal forward and backward (train.hip attn_bwd_{q,kv}_mfma_kernel:
                                    2, 4, 8 or 16 key tiles -- the spatial axis' 17 joints and F = 9 frames use 2);
      valu (+ D3DP_TRAIN_ATTN_BWD=valu) -- the two-threads-per-row VALU backward kernels.
    All fp32-class in a different summation order: every parameter gradient of a training step agrees to fp32 noise."""
    B, cs, dep = 2, 512, 2
    args = SimpleNamespace(number_of_frames=Fr, test_time_augmentation=True, timestep=1000, scale=1.0, cs=cs, dep=dep)
    sd = make_state_dict(11, cs, dep, Fr)
    x2d = torch.from_numpy(synthetic_inputs_2d(911, B, Fr)).cuda()
    gt = torch.from_numpy(synthetic_noise(912, (B, Fr, 17, 3))) * 0.3
    gt[:, :, 0] = 0
    gt = gt.cuda()
    t = torch.tensor([[30], [700]], dtype=torch.long)
    noise = torch.from_numpy(synthetic_noise(913, (B, Fr, 17, 3)))
    grads, preds = {}, {}
    for impl in ("valu", "mfma", "x2t", "x2"):
        if impl == "x2":                                                # both axes on the split-fp16 kernels (the default)
            monkeypatch.delenv("D3DP_TRAIN_ATTN", raising=False)
            monkeypatch.delenv("D3DP_TRAIN_ATTN_BWD", raising=False)
        elif impl == "x2t":                                             # the temporal axis only
            monkeypatch.setenv("D3DP_TRAIN_ATTN", "x2t")
            monkeypatch.delenv("D3DP_TRAIN_ATTN_BWD", raising=False)
        else:
            monkeypatch.setenv("D3DP_TRAIN_ATTN", "f32")               # (read when the context is created: one model each)
            monkeypatch.setenv("D3DP_TRAIN_ATTN_BWD", impl)
        m = D3DP(args, H36M_JOINTS_LEFT, H36M_JOINTS_RIGHT, is_train=True)
        m.load_state_dict(sd, strict=False)
        m = m.cuda().train()
        pred = m(x2d, gt, t=t, noise=noise, droppath={})
        loss = torch.mean(torch.norm(pred - gt, dim=-1))
        loss.backward(loss.clone().detach())
        torch.cuda.synchronize()
        preds[impl] = pred.detach().double().cpu()
        grads[impl] = {k: p.grad.double().cpu() for k, p in m.pose_estimator.named_parameters()}
        assert all(torch.isfinite(g).all() for g in grads[impl].values()), impl
    for impl in ("mfma", "x2t", "x2"):
        worst = 0.0
        for k in grads["valu"]:
            err = (grads["valu"][k] - grads[impl][k]).norm().item() / max(grads["valu"][k].norm().item(), 1e-30)
            worst = max(worst, err)
            assert err < 2e-5, (impl, k, err)
        assert any(not torch.equal(grads["valu"][k], grads[impl][k]) for k in grads["valu"])   # (the switch did select another kernel)
        dp = (preds[impl] - preds["valu"]).abs().max().item()
        assert dp < 2e-5, (impl, dp)
        print(f"attention of the training step, F={Fr}: {impl} vs VALU kernels, worst relative gradient difference {worst:.2e}, "
              f"prediction max |diff| {dp:.2e}")


@pytest.mark.parametrize("cs", [512, 256, 128])
def test_training_step_fp32_linears_cross_check(monkeypatch, cs):
    """D3DP_TRAIN_IMPL=f32 (read when the context is created) keeps the training Linears on the fp32 matrix cores -- the round-1
    path, left in as the cross-check of the split-fp16 one with its fused operand preparation: loss and every parameter
    gradient of one step agree between the two to fp32 noise.  cs = 256 (the reference's smaller model: 32-wide heads, so the
    attention stays on the fp32 kernels) has other tile counts in the merged weight-gradient launch: 16 tiles x 16 chunks; at
    cs = 128 no Linear has a TN shape and the weight gradients take the transposed-operand path, one launch each."""
    Fr, B, dep = 27, 2, 2
    x2d = torch.from_numpy(synthetic_inputs_2d(921, B, Fr)).cuda()
    gt = torch.from_numpy(synthetic_noise(922, (B, Fr, 17, 3))) * 0.3
    gt[:, :, 0] = 0
    gt = gt.cuda()
    t = torch.tensor([[30], [700]], dtype=torch.long)
    noise = torch.from_numpy(synthetic_noise(923, (B, Fr, 17, 3)))
    res = []
    for impl in ("f32", None):
        if impl:
            monkeypatch.setenv("D3DP_TRAIN_IMPL", impl)
        else:
            monkeypatch.delenv("D3DP_TRAIN_IMPL", raising=False)
        args = SimpleNamespace(number_of_frames=Fr, test_time_augmentation=True, timestep=1000, scale=1.0, cs=cs, dep=dep)
        m = D3DP(args, H36M_JOINTS_LEFT, H36M_JOINTS_RIGHT, is_train=True)
        m.load_state_dict(make_state_dict(13, cs, dep, Fr), strict=False)
        m = m.cuda().train()
        pred = m(x2d, gt, t=t, noise=noise, droppath={})
        loss = torch.mean(torch.norm(pred - gt, dim=-1))
        loss.backward(loss.clone().detach())
        torch.cuda.synchronize()
        assert ("fp32 MFMA Linears" in m.pose_estimator.train_arithmetic()) == (impl == "f32")
        res.append((loss.item(), {k: p.grad.double().cpu() for k, p in m.pose_estimator.named_parameters()}))
    assert abs(res[0][0] - res[1][0]) < 2e-6
    worst = 0.0
    for k in res[0][1]:
        err = (res[0][1][k] - res[1][1][k]).norm().item() / max(res[0][1][k].norm().item(), 1e-30)
        worst = max(worst, err)
        assert err < 2e-5, (k, err)
    print(f"training step, fp32-MFMA Linears vs split-fp16 Linears: worst relative gradient difference {worst:.2e}")


_TRAIN_SWITCHES = ("D3DP_TRAIN_WGRAD", "D3DP_TRAIN_TAIL", "D3DP_TRAIN_GELU", "D3DP_TRAIN_OVERLAP", "D3DP_TRAIN_LN_OPERAND")


def _training_step_under(monkeypatch, cases):
    """One configs[4]-shaped step (T = 2 x 243 x 17 = 32 x 256 + 70 rows: the fc1 products take the remainder-block path) per
    entry of `cases` (name -> environment read when the context is created): {name: (loss, {parameter: gradient})}."""
    Fr, B, cs, dep = 243, 2, 512, 2
    x2d = torch.from_numpy(synthetic_inputs_2d(931, B, Fr)).cuda()
    gt = torch.from_numpy(synthetic_noise(932, (B, Fr, 17, 3))) * 0.3
    gt[:, :, 0] = 0
    gt = gt.cuda()
    t = torch.tensor([[30], [700]], dtype=torch.long)
    noise = torch.from_numpy(synthetic_noise(933, (B, Fr, 17, 3)))
    g = torch.Generator().manual_seed(934)
    drop = {}
    for i in range(dep):                                   # recorded DropPath masks: every run scales the same samples
        for kind, n in (("STEblocks", B * Fr), ("TTEblocks", B * 17)):
            drop[f"{kind}.{i}"] = tuple((torch.rand(n, generator=g) < 0.85).float() / 0.85 for _ in range(2))
    res = {}
    for name, env in cases:
        for k in _TRAIN_SWITCHES:
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        args = SimpleNamespace(number_of_frames=Fr, test_time_augmentation=True, timestep=1000, scale=1.0, cs=cs, dep=dep)
        m = D3DP(args, H36M_JOINTS_LEFT, H36M_JOINTS_RIGHT, is_train=True)
        m.load_state_dict(make_state_dict(17, cs, dep, Fr), strict=False)
        m = m.cuda().train()
        pred = m(x2d, gt, t=t, noise=noise, droppath=drop)
        loss = torch.mean(torch.norm(pred - gt, dim=-1))
        loss.backward(loss.clone().detach())
        torch.cuda.synchronize()
        res[name] = (loss.item(), {k: p.grad.double().cpu() for k, p in m.pose_estimator.named_parameters()})
    return res


def test_training_step_stream_switches_change_no_bit(monkeypatch):
    """D3DP_TRAIN_OVERLAP=1 | 0 (one operand set on the second stream / everything on the caller's stream: the profiling switch the
    one-stream kernel tables under profiles/ are taken with) change the schedule and no arithmetic: every gradient bit-equal."""
    res = _training_step_under(monkeypatch, (("default", {}), ("one_set", {"D3DP_TRAIN_OVERLAP": "1"}), ("one_stream", {"D3DP_TRAIN_OVERLAP": "0"})))
    ref = res["default"]
    for name in ("one_set", "one_stream"):
        assert res[name][0] == ref[0] and all(torch.equal(res[name][1][k], ref[1][k]) for k in ref[1]), name


def test_product_library_refuses_the_superseded_training_launch_forms(monkeypatch):
    """round 5's A/B switches select launch forms that were measured and superseded; the product library
    REFUSES them (D3DP_ENOTSUP) instead of ignoring them -- only the `make variants` build honours them."""
    if _lib.load().d3dp_debug_x2_variants() == 1:
        pytest.skip("this is the variants build: it honours the switches (test_training_step_scheduling_switches_agree)")
    for k, v in (("D3DP_TRAIN_WGRAD", "each"), ("D3DP_TRAIN_TAIL", "split"), ("D3DP_TRAIN_GELU", "pass"), ("D3DP_TRAIN_LN_OPERAND", "pass")):
        for kk in _TRAIN_SWITCHES:
            monkeypatch.delenv(kk, raising=False)
        monkeypatch.setenv(k, v)
        args = SimpleNamespace(number_of_frames=27, test_time_augmentation=True, timestep=1000, scale=1.0, cs=64, dep=1)
        m = D3DP(args, H36M_JOINTS_LEFT, H36M_JOINTS_RIGHT, is_train=True).cuda()
        with pytest.raises(_lib.D3DPHipError, match="superseded launch forms"):
            m.pose_estimator._context(torch.device("cuda", 0))


@pytest.mark.variants
def test_training_step_scheduling_switches_agree(monkeypatch):
    """(variants build) Round 5's scheduling changes of the training step against the forms they replaced: one launch for a
    block's four weight gradients / one each (D3DP_TRAIN_WGRAD=each), a batch's last T mod 256 rows as 16 x 64 blocks / as a
    round of tiles or a split-K launch (D3DP_TRAIN_TAIL=split), d h_pre inside the operand pass / by a pass of its own
    (D3DP_TRAIN_GELU=pass).  Same arithmetic up to the grouping of fp32 partial sums and one operand scale: loss and every
    gradient agree to fp32 noise."""
    res = _training_step_under(monkeypatch, (
        ("default", {}), ("wgrad_each", {"D3DP_TRAIN_WGRAD": "each"}), ("tail_split", {"D3DP_TRAIN_TAIL": "split"}),
        ("gelu_pass", {"D3DP_TRAIN_GELU": "pass"}), ("ln_operand_pass", {"D3DP_TRAIN_LN_OPERAND": "pass"}),
        ("round4", {"D3DP_TRAIN_WGRAD": "each", "D3DP_TRAIN_TAIL": "split", "D3DP_TRAIN_GELU": "pass", "D3DP_TRAIN_OVERLAP": "1"})))
    ref = res["default"]
    for name, (loss, grads) in res.items():
        if name == "default":
            continue
        assert abs(loss - ref[0]) < 2e-6, (name, loss, ref[0])
        worst = max((grads[k] - ref[1][k]).norm().item() / max(ref[1][k].norm().item(), 1e-30) for k in grads)
        print(f"training step, {name} vs default: loss difference {abs(loss - ref[0]):.1e}, worst relative gradient difference {worst:.2e}")
        assert worst < 2e-5, (name, worst)


def test_bench_launches_its_own_ranks_when_two_gpus_are_visible():
    """`python bench.py --gpus 2` outside a torchrun job re-executes itself as 2 RCCL ranks, checks that the 2-rank
    run on sliced global noise reproduces the 1-rank H=2*H_local run bit for bit, and reports the world size and the
    all-gather.  Needs two devices on the box: skipped on the 1-GPU test boxes."""
    import json
    import subprocess
    import sys
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs on the box")
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--batch", "1", "--hyps", "2", "--ksteps", "1", "--no-cpu-baseline", "--no-parity", "--no-profile"],
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["multi_gpu"]["world_size"] == 2 and d["multi_gpu"]["sharded_equals_single_rank"]
    assert d["config"]["hypotheses_total"] == 4


def test_rccl_side_of_the_n_rank_path_executes_on_one_gpu():
    """`python bench.py --force-exchange` on ONE GPU: the N-rank code path of bench.py / d3dp_amd.dist at world size 1 over the
    nccl (= RCCL) backend -- rendezvous, communicator and probe all-reduce (`init_from_env`), the shard check through
    `all_gather_hypotheses`, every timed step ending in `all_gather_into_tensor` into the rank-major buffer with
    `d3dp_jpma_gathered` launched behind it on torch's stream, the winners exchange beside it.  At world size 1 RCCL's
    collectives are one-rank copies: what this pins is that the transport stack initialises on the box and that its results
    reach the library's kernels in order -- the part of BASELINE configs[3] that no gloo test touches.  (Two ranks on one GPU
    are refused by RCCL: profiles/r06_rccl_same_gpu_probe.json.)"""
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    s = __import__("socket").socket()
    s.bind(("127.0.0.1", 0))
    env["MASTER_PORT"] = str(s.getsockname()[1])
    s.close()
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--force-exchange", "--steps", "2", "--warmup", "1", "--batch", "2",
                        "--hyps", "3", "--ksteps", "2", "--no-profile"], capture_output=True, text=True, env=env, timeout=900)
    if r.returncode != 0 and "could not join the nccl process group" in r.stderr:
        pytest.skip("RCCL does not initialise on this box: " + r.stderr.strip().splitlines()[-1][:300])
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    mg = d["multi_gpu"]
    assert d["n_gpus"] == 1 and mg["world_size"] == 1 and mg["backend"] == "nccl (RCCL)" and "forced" in mg
    assert mg["sharded_equals_single_rank"] is True and mg["both_select_the_same_poses"] is True
    assert mg["all_gather_bytes_per_rank"] == 2 * 2 * 3 * 243 * 17 * 3 * 4 and mg["all_gather_ms"] > 0
    assert d["value"] > 0 and d["config"]["hypotheses_total"] == 3 and "fast_mode" not in d and "cpu_baseline" not in d


def _dp_case(B):
    frames, H, K = 27, 2, 2
    m = make_model(frames, 64, 2, H, K, "exact", seed=5)
    x2d = torch.from_numpy(synthetic_inputs_2d(77, B, frames)).cuda()
    noises = [torch.from_numpy(synthetic_noise(300 + k, (B, H, frames, 17, 3))).cuda() for k in range(K)]
    return m, x2d, flip_2d(x2d), noises


def test_data_parallel_wrapper_on_one_device_matches_the_bare_model():
    """The reference's caller wraps every model in nn.DataParallel (main.py:242-248) and calls it with
    `input_2d_flip=` (:698).  With one device DataParallel forwards to the module itself: same result, the same
    library context across calls, and wrapping frees nothing."""
    m, x2d, x2f, noises = _dp_case(2)
    ref = m(x2d, None, input_2d_flip=x2f, noise=noises)
    h = m.pose_estimator._ctx.value
    dp = torch.nn.DataParallel(m, device_ids=[0])
    for _ in range(2):
        out = dp(x2d, None, input_2d_flip=x2f, noise=noises)
        assert torch.equal(out, ref) and m.pose_estimator._ctx.value == h


def test_data_parallel_over_two_devices_matches_the_single_device_result():
    """nn.DataParallel(model, [0, 1]) splits the clip batch (and the injected noise tensors, scattered
    along dim 0 like every tensor argument) over two devices; every replica runs its own library context on its own
    device, one thread each, and the gathered result equals the single-device run.  Called twice: the replicas of the
    first call are garbage by then and must not have freed anything."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs on the box")
    m, x2d, x2f, noises = _dp_case(4)
    ref = m(x2d, None, input_2d_flip=x2f, noise=noises)
    h0 = m.pose_estimator._ctx.value
    dp = torch.nn.DataParallel(m, device_ids=[0, 1])
    for _ in range(2):
        out = dp(x2d, None, input_2d_flip=x2f, noise=noises)
        assert out.device == x2d.device and out.shape == ref.shape
        assert torch.equal(out, ref)                       # hypotheses and clips are independent: bit for bit
    st = m.pose_estimator._states
    assert len(st) == 2 and st[torch.device("cuda", 0)].ctx.value == h0


def test_data_parallel_replicas_follow_weight_changes_and_train():
    """ADVICE r4 (high).  Real `torch.nn.parallel.replicate` replicas (both on device 0, driven one after the other: one GPU
    is enough) have an empty `parameters()`; their packed weights must follow the SOURCE module -- main.py:242-258, :450
    loads the training weights into the evaluation model every epoch and evaluates through nn.DataParallel -- and the
    training branch must send its gradients back through the broadcast edge to the source parameters."""
    from torch.nn.parallel import replicate
    m, x2d, x2f, noises = _dp_case(2)
    ref1 = m(x2d, None, input_2d_flip=x2f, noise=noises)
    r = replicate(m, [0, 0])[1]
    assert list(r.pose_estimator.parameters()) == []
    assert torch.equal(r(x2d, None, input_2d_flip=x2f, noise=noises), ref1)
    sd = {k: (v * 1.01 if v.dtype == torch.float32 and v.dim() == 2 else v) for k, v in m.state_dict().items()}
    m.load_state_dict(sd)                                   # new epoch's weights into the wrapped module
    out2 = replicate(m, [0, 0])[1](x2d, None, input_2d_flip=x2f, noise=noises)
    ref2 = m(x2d, None, input_2d_flip=x2f, noise=noises)
    assert torch.equal(out2, ref2) and not torch.equal(ref2, ref1)   # the replica ran on the NEW weights
    # training through a replica: gradients arrive on the source module's parameters and equal the bare module's
    frames, B, cs, dep = 27, 2, 64, 2
    args = SimpleNamespace(number_of_frames=frames, test_time_augmentation=True, timestep=1000, scale=1.0, cs=cs, dep=dep)
    mt = D3DP(args, H36M_JOINTS_LEFT, H36M_JOINTS_RIGHT, is_train=True)
    mt.load_state_dict(make_state_dict(5, cs, dep, frames), strict=False)
    mt = mt.cuda().eval()                                   # DropPath off (D3DP.is_train keeps the train branch)
    gt = torch.from_numpy(synthetic_noise(9, (B, frames, 17, 3))).cuda() * 0.3
    t = torch.tensor([[10], [500]])
    nz = torch.from_numpy(synthetic_noise(10, (B, frames, 17, 3)))

    def grads(model):
        mt.zero_grad(set_to_none=True)
        out = model(x2d[:B], gt, t=t, noise=nz)
        out.square().sum().backward()
        return [None if p.grad is None else p.grad.clone() for p in mt.parameters()]

    g_ref = grads(mt)
    g_rep = grads(replicate(mt, [0, 0])[1])
    assert all(a is not None for a in g_rep) and len(g_rep) == len(g_ref) > 0
    for a, b in zip(g_rep, g_ref):
        assert torch.equal(a, b)
    with torch.no_grad():                                   # "optimizer step", then a second replica forward / backward
        for p in mt.parameters():
            p.mul_(0.99)
    g_ref2 = grads(mt)
    g_rep2 = grads(replicate(mt, [0, 0])[1])
    assert all(torch.equal(a, b) for a, b in zip(g_rep2, g_ref2)) and not torch.equal(g_ref2[0], g_ref[0])


def test_training_step_is_bit_reproducible():
    """VERDICT r4 weak 5 / ADVICE r4: no gradient of the training step is summed with float atomics any more -- LayerNorm
    gammas / betas, biases and the embedding-side gradients leave as per-workgroup partial rows that one fixed-order reduction
    adds (train.hip reduce_many_kernel), weight gradients as split-K partial products summed in order.  Two runs of the same
    step (DropPath masks injected) give bit-identical predictions and gradients, parameter by parameter."""
    Fr, B, cs, dep = 81, 3, 512, 2
    args = SimpleNamespace(number_of_frames=Fr, test_time_augmentation=True, timestep=1000, scale=1.0, cs=cs, dep=dep)
    m = D3DP(args, H36M_JOINTS_LEFT, H36M_JOINTS_RIGHT, is_train=True)
    m.load_state_dict(make_state_dict(3, cs, dep, Fr), strict=False)
    m = m.cuda().train()
    x2d = torch.from_numpy(synthetic_inputs_2d(21, B, Fr)).cuda()
    gt = (torch.from_numpy(synthetic_noise(22, (B, Fr, 17, 3))) * 0.3).cuda()
    t = torch.tensor([[5], [400], [990]])
    noise = torch.from_numpy(synthetic_noise(23, (B, Fr, 17, 3)))
    gen = torch.Generator().manual_seed(1)
    dpd = {f"{kind}.1": tuple((torch.rand(S, 1, 1, generator=gen) < 0.9).float() / 0.9 for _ in range(2))
           for kind, S in (("STEblocks", B * Fr), ("TTEblocks", B * 17))}

    def step():
        m.zero_grad(set_to_none=True)
        pred = m(x2d, gt, t=t, noise=noise, droppath=dpd)
        loss = torch.mean(torch.norm(pred - gt, dim=-1))
        loss.backward(loss.clone().detach())
        torch.cuda.synchronize()
        return pred.detach().clone(), [p.grad.clone() for p in m.parameters()]

    p0, g0 = step()
    for _ in range(2):
        p1, g1 = step()
        assert torch.equal(p0, p1)
        for (name, _), a, b in zip(m.named_parameters(), g0, g1):
            assert torch.equal(a, b), name
    assert all(torch.isfinite(g).all() for g in g0) and any(g.abs().max() > 0 for g in g0)


def _small_deep_training_model(Fr=27, B=2, cs=128, dep=8, seed=17):
    args = SimpleNamespace(number_of_frames=Fr, test_time_augmentation=True, timestep=1000, scale=1.0, cs=cs, dep=dep)
    sd = make_state_dict(seed, cs, dep, Fr)
    m = D3DP(args, H36M_JOINTS_LEFT, H36M_JOINTS_RIGHT, is_train=True)
    m.load_state_dict(sd, strict=False)
    return m.cuda().train(), sd


def test_training_first_step_at_depth_8_on_a_poisoned_workspace():
    """at the reference's default depth (dep = 8: 140 reduction items against 80 per launch) the fixed-order
    reduction of round 5 ran its first launch from INSIDE the block loop, over the shared Spatial_norm / Temporal_norm partial rows
    of blocks not yet differentiated -- stale rows of the previous step, or whatever torch.empty left on the first.  Here: one
    step on OTHER inputs, the whole workspace overwritten with NaN bit patterns, then the step under test -- every gradient
    against torch autograd through the CPU oracle."""
    Fr, B, cs, dep = 27, 2, 128, 8
    m, sd = _small_deep_training_model(Fr, B, cs, dep)
    gen = torch.Generator().manual_seed(9)
    dpd = {}
    rates = [x.item() for x in torch.linspace(0, 0.1, dep)]
    for i in range(1, dep):
        mk = lambda S: (torch.rand(S, 1, 1, generator=gen) < 1 - rates[i]).float() / (1 - rates[i])
        dpd[f"STEblocks.{i}"] = (mk(B * Fr), mk(B * Fr))
        dpd[f"TTEblocks.{i}"] = (mk(B * 17), mk(B * 17))

    def step(seed, tvals):
        x2d = torch.from_numpy(synthetic_inputs_2d(seed, B, Fr))
        gt = torch.from_numpy(synthetic_noise(seed + 1, (B, Fr, 17, 3))) * 0.3
        noise = torch.from_numpy(synthetic_noise(seed + 2, (B, Fr, 17, 3)))
        t = torch.tensor(tvals, dtype=torch.long).reshape(-1, 1)
        m.zero_grad(set_to_none=True)
        pred = m(x2d.cuda(), gt.cuda(), t=t, noise=noise, droppath=dpd)
        loss = torch.mean(torch.norm(pred - gt.cuda(), dim=-1))
        loss.backward(loss.clone().detach())
        torch.cuda.synchronize()
        return x2d, gt, noise, t

    step(700, [17, 803])                                   # allocates the workspace, leaves ITS partial rows behind
    st = m.pose_estimator._state()
    st.train_ws.fill_(0xFF)                                # every float of the workspace: a NaN
    x2d, gt, noise, t = step(800, [250, 999])
    po = {k: v.clone().requires_grad_(True) for k, v in orc.strip_prefix(sd).items()}
    xp = orc.prepare_targets(orc.cosine_schedule(1000), gt, t[:, 0], noise)
    pred_o = orc.mixste_forward(po, x2d, xp, t[:, 0], dep, droppath=dpd)
    loss_o = torch.mean(torch.norm(pred_o - gt, dim=-1))
    loss_o.backward(loss_o.clone().detach())
    worst = ("", 0.0)
    for name, p in m.pose_estimator.named_parameters():
        assert torch.isfinite(p.grad).all(), name
        ref = po[name].grad.double()
        err = (p.grad.cpu().double() - ref).norm().item() / max(ref.norm().item(), 1e-12)
        worst = max(worst, (name, err), key=lambda v: v[1])
        assert err < 2e-3, (name, err)
    print(f"dep = 8, first step on a poisoned workspace: worst relative gradient error {worst[1]:.2e} ({worst[0]})")


def test_second_backward_over_the_same_forward_gives_the_same_gradients():
    """the backward pass reads the head LayerNorm's output `z` as the forward pass left it; round 5 then
    overwrote it (scratch for the Temporal_pos_embed gradient), so a second backward over the same forward (retain_graph=True:
    no forward re-run, the generation counter is unchanged) took head.1.weight's gradient from garbage.  Two backward passes
    over one forward must agree bit for bit."""
    m, _ = _small_deep_training_model(27, 2, 128, 2, seed=5)
    x2d = torch.from_numpy(synthetic_inputs_2d(31, 2, 27)).cuda()
    gt = (torch.from_numpy(synthetic_noise(32, (2, 27, 17, 3))) * 0.3).cuda()
    pred = m(x2d, gt, t=torch.tensor([[40], [900]]), noise=torch.from_numpy(synthetic_noise(33, (2, 27, 17, 3))))
    loss = torch.mean(torch.norm(pred - gt, dim=-1))
    loss.backward(loss.clone().detach(), retain_graph=True)
    torch.cuda.synchronize()
    first = {n: p.grad.clone() for n, p in m.named_parameters()}
    m.zero_grad(set_to_none=True)
    loss.backward(loss.clone().detach())
    torch.cuda.synchronize()
    for n, p in m.named_parameters():
        assert torch.equal(first[n], p.grad), n
    assert first["pose_estimator.head.1.weight"].abs().max() > 0


@pytest.mark.parametrize("Fr", [300, 513])
def test_training_step_on_a_clip_longer_than_256_frames(Fr):
    """VERDICT r5 missing 3 / item 6: the reference trains at any `-f` (common/arguments.py:58, main.py:325); through round 5
    d3dp_train_forward refused more than 256 frames (its attention kernels held a whole sequence in LDS).  The split-fp16 attention
    kernels now pass keys (forward, pass Q) and queries (pass KV) through LDS in chunks of 128, so the step runs at 300 and 513
    frames: prediction, loss and EVERY gradient against torch autograd through the CPU oracle (cs = 512, dep = 1, B = 1, DropPath
    masks injected)."""
    B, cs, dep = 1, 512, 1
    m, sd = _small_deep_training_model(Fr, B, cs, dep, seed=29)
    x2d = torch.from_numpy(synthetic_inputs_2d(41, B, Fr))
    gt = torch.from_numpy(synthetic_noise(42, (B, Fr, 17, 3))) * 0.3
    noise = torch.from_numpy(synthetic_noise(43, (B, Fr, 17, 3)))
    t = torch.tensor([[321]], dtype=torch.long)
    pred = m(x2d.cuda(), gt.cuda(), t=t, noise=noise)
    loss = torch.mean(torch.norm(pred - gt.cuda(), dim=-1))
    loss.backward(loss.clone().detach())
    torch.cuda.synchronize()
    po = {k: v.clone().requires_grad_(True) for k, v in orc.strip_prefix(sd).items()}
    xp = orc.prepare_targets(orc.cosine_schedule(1000), gt, t[:, 0], noise)
    pred_o = orc.mixste_forward(po, x2d, xp, t[:, 0], dep, droppath=None)
    loss_o = torch.mean(torch.norm(pred_o - gt, dim=-1))
    loss_o.backward(loss_o.clone().detach())
    assert orc.mpjpe_mm(pred.detach().cpu(), pred_o.detach()) <= EXACT_TOL_MM
    assert abs(loss.item() - loss_o.item()) < 2e-6
    worst = ("", 0.0)
    for name, p in m.pose_estimator.named_parameters():
        ref = po[name].grad.double()
        err = (p.grad.cpu().double() - ref).norm().item() / max(ref.norm().item(), 1e-12)
        worst = max(worst, (name, err), key=lambda v: v[1])
        assert err < 2e-3, (name, err)
    print(f"training step at F = {Fr}: worst relative gradient error {worst[1]:.2e} ({worst[0]})")


def test_non_finite_weights_load_and_propagate():
    """a diverged checkpoint (inf / nan in ANY weight tensor) loads like it does in the reference and produces
    non-finite outputs there too, whichever tensor holds the value -- never a load-time error in one case and a silent
    implementation switch in the other.  The EXACT context moves to its split-bf16 implementation (fp32's range) and
    d3dp_status reports the non-finite output."""
    frames, cs, dep = 27, 64, 2
    x2d = torch.from_numpy(synthetic_inputs_2d(5, 1, frames)).cuda()
    nz = [torch.from_numpy(synthetic_noise(6, (1, 1, frames, 17, 3)))]
    for key, val in (("pose_estimator.STEblocks.0.attn.qkv.weight", float("nan")), ("pose_estimator.TTEblocks.1.mlp.fc2.weight", float("inf")),
                     ("pose_estimator.STEblocks.1.norm2.weight", float("inf")), ("pose_estimator.TTEblocks.0.attn.proj.weight", float("inf"))):
        sd = make_state_dict(9, cs, dep, frames)
        sd[key] = sd[key].clone()
        sd[key].view(-1)[3] = val
        args = SimpleNamespace(number_of_frames=frames, test_time_augmentation=True, timestep=1000, scale=1.0, cs=cs, dep=dep)
        m = D3DP(args, H36M_JOINTS_LEFT, H36M_JOINTS_RIGHT, is_train=False, num_proposals=1, sampling_timesteps=1, numerics="exact")
        m.load_state_dict(sd, strict=False)
        m = m.cuda().eval()
        out = m(x2d, None, input_2d_flip=flip_2d(x2d), noise=nz)       # loads and runs: no D3DP_EINVAL
        net = m.pose_estimator
        assert net.exact_scales()[2] == "bf16x3", key
        want = orc.ddim_sample_flip(orc.strip_prefix(sd), orc.cosine_schedule(1000), x2d.cpu(), flip_2d(x2d).cpu(), 1, 1, dep,
                                    H36M_JOINTS_LEFT, H36M_JOINTS_RIGHT, nz)
        assert not torch.isfinite(want).all(), key             # the reference's output is non-finite ...
        assert net.nonfinite_seen(), key                       # ... and the library says so about its own (d3dp_status)
        assert out.shape == want.shape


# ---- the training step's Linear alone: the rows behind the last whole 256-row tile ----------------------------------------------
def _train_linear(A, W, b, tail, amax_pos=0):
    import ctypes as C
    lib = _lib.load()
    fn = lib.d3dp_debug_train_linear                       # test hook (include/d3dp_hip.h, "test hooks")
    M, K = A.shape
    N = W.shape[0]
    out = torch.full((M, N), float("nan"), device="cuda")
    amax = torch.zeros(1, dtype=torch.int32, device="cuda")
    _lib.check(fn(A.data_ptr(), W.data_ptr(), _lib.ptr(b), out.data_ptr(), M, N, K, tail, amax.data_ptr(), amax_pos, stream()),
               "d3dp_debug_train_linear")
    return out, amax.view(torch.float32).item()


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K", [(16524, 512, 512), (16524, 1536, 512), (16524, 1024, 512), (16524, 512, 1024),
                                   (16524, 512, 1536), (257, 64, 512), (300, 132, 512), (256 + 255, 512, 1024),
                                   (3 * 256 + 16, 2048, 512), (2 * 256 + 1, 36, 2048), (700, 512, 480), (200, 512, 512)])
def test_training_linear_remainder_rows_as_blocks(M, N, K):
    """gemm_f16x2_dyn_kernel with the rows behind the last whole tile cut into 16 x 64 blocks (what the training step launches
    wherever they would cost a round of the persistent loop) against the same kernel with a last row of tiles, and both
    against fp64: split-fp16 operands carry 22 bits, the products accumulate in fp32."""
    g = torch.Generator(device="cuda").manual_seed(M * 31 + N * 7 + K)
    A = torch.randn(M, K, device="cuda", generator=g) * 3.0
    W = torch.randn(N, K, device="cuda", generator=g) / K ** 0.5
    b = torch.randn(N, device="cuda", generator=g)
    ref = A.double() @ W.double().t() + b.double()
    tiles, amax_t = _train_linear(A, W, b, 0)
    blocks, amax_b = _train_linear(A, W, b, 1)
    scale = (A.double().abs() @ W.double().abs().t()).max().item()       # what the rounding errors of a product scale with
    assert torch.isfinite(blocks).all()
    e_t, e_b = (tiles.double() - ref).abs().max().item() / scale, (blocks.double() - ref).abs().max().item() / scale
    print(f"training Linear M={M} N={N} K={K}: max error / max sum|a||w| = {e_t:.2e} (tiles) {e_b:.2e} (remainder blocks)")
    assert e_t < 2e-6 and e_b < 2e-6
    q = M // 256 * 256
    assert torch.equal(tiles[:q], blocks[:q])                             # the whole tiles are the same work items
    assert amax_t == tiles.abs().max().item() and amax_b == blocks.abs().max().item()
    nobias, pmax = _train_linear(A, W, None, 1, amax_pos=1)
    assert pmax == max(nobias.max().item(), 0.0)
    assert torch.allclose(nobias.double() + b.double(), blocks.double(), atol=1e-5 * scale, rtol=0)
