#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE (authoring container only).

The reference (/root/reference, read-only, never copied) ships no tests or golden
vectors, so parity is pinned on outputs of the reference itself:

  * ``timm`` (absent here) is stubbed in ``sys.modules`` -- mixste.py:18-21 import four
    timm submodules but only ``DropPath`` is used, and only in train mode.
  * ``ddim_sample_flip`` hard-codes ``device='cuda'`` (diffusionpose.py:225,230): in this
    process only, ``torch.randn/randn_like/randint`` are wrapped to stay on CPU and to
    return INJECTED draws (numpy PCG64 streams from ``d3dp_amd.weights``), and
    ``Tensor.cuda`` is the identity.  No reference file is modified.
  * weights come from ``d3dp_amd.weights.make_state_dict(seed, ...)`` and are loaded
    with ``load_state_dict`` so fixtures carry seeds, not 139 MB of parameters.

Fixtures hold data only (inputs / seeds / expected outputs).  This script cannot run on
the GPU box (no /root/reference there) and nothing in tests/ or bench.py needs it to.

Usage: python tools/make_goldens.py [--only g1,g2,...]
"""
from __future__ import annotations

import argparse
import os
import sys
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, REPO)
OUT = os.path.join(REPO, "tests", "golden")

from d3dp_amd.weights import (H36M_JOINTS_LEFT, H36M_JOINTS_RIGHT, flip_2d, make_state_dict,  # noqa: E402
                              synthetic_inputs_2d, synthetic_noise)


# ----------------------------------------------------------------------------- reference import
class _DropPathStub(torch.nn.Module):
    """timm.models.layers.DropPath semantics (per-sample Bernoulli keep mask scaled by
    1/keep, identity in eval) with injectable masks so fixtures can record them."""
    injected = None   # list of (S,1,1) tensors consumed in call order, or None -> identity
    log = []

    def __init__(self, drop_prob=0.0):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        if not self.training or self.drop_prob == 0.0:
            if self.training:
                _DropPathStub.log.append(torch.ones(x.shape[0], 1, 1))
            return x
        if _DropPathStub.injected is None:
            _DropPathStub.log.append(torch.ones(x.shape[0], 1, 1))
            return x
        m = _DropPathStub.injected.pop(0)
        _DropPathStub.log.append(m)
        return x * m


def import_reference():
    for name in ("timm", "timm.data", "timm.models", "timm.models.helpers", "timm.models.layers",
                 "timm.models.registry"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["timm.data"].IMAGENET_DEFAULT_MEAN = None
    sys.modules["timm.data"].IMAGENET_DEFAULT_STD = None
    sys.modules["timm.models.helpers"].load_pretrained = None
    lay = sys.modules["timm.models.layers"]
    lay.DropPath, lay.to_2tuple, lay.trunc_normal_ = _DropPathStub, None, None
    sys.modules["timm.models.registry"].register_model = lambda f: f
    sys.path.insert(0, REF)
    from common.diffusionpose import D3DP  # noqa
    return D3DP


class Draws:
    """Patches torch RNG entry points used by the reference; returns injected tensors."""

    def __init__(self, randn_list=None, randint_list=None):
        self.randn_list = list(randn_list or [])
        self.randint_list = list(randint_list or [])
        self.n_randn = 0

    def __enter__(self):
        self._o = (torch.randn, torch.randn_like, torch.randint, torch.Tensor.cuda)

        def randn(*size, **kw):
            self.n_randn += 1
            shape = tuple(size[0]) if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)) else tuple(size)
            t = self.randn_list.pop(0)
            assert tuple(t.shape) == shape, (t.shape, shape)
            return t.clone()

        def randn_like(x, **kw):
            return randn(tuple(x.shape))

        def randint(low, high, size, **kw):
            t = self.randint_list.pop(0)
            assert tuple(t.shape) == tuple(size)
            return t.clone()

        torch.randn, torch.randn_like, torch.randint = randn, randn_like, randint
        torch.Tensor.cuda = lambda self_, *a, **k: self_
        return self

    def __exit__(self, *exc):
        torch.randn, torch.randn_like, torch.randint, torch.Tensor.cuda = self._o


def make_args(frames, cs, dep, scale=1.0):
    return types.SimpleNamespace(number_of_frames=frames, test_time_augmentation=True, timestep=1000,
                                 scale=scale, cs=cs, dep=dep)


def build_ref(D3DP, frames, cs, dep, seed, is_train=False, H=1, K=1):
    torch.manual_seed(0)
    m = D3DP(make_args(frames, cs, dep), H36M_JOINTS_LEFT, H36M_JOINTS_RIGHT, is_train=is_train,
             num_proposals=H, sampling_timesteps=K)
    missing, unexpected = m.load_state_dict(make_state_dict(seed, cs, dep, frames), strict=False)
    assert not unexpected and all(not k.startswith("pose_estimator") for k in missing), (missing, unexpected)
    return m.train() if is_train else m.eval()


def save(name, **arrs):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrs)
    print(f"  wrote {path}  ({os.path.getsize(path) / 1024:.1f} KiB)")


# ----------------------------------------------------------------------------- fixtures
def g0(D3DP):
    """state_dict contract: every key, shape and dtype of the reference D3DP at the shipped defaults."""
    m = build_ref(D3DP, 243, 512, 8, seed=1, H=20, K=10)
    sd = m.state_dict()
    save("g0_state_dict_contract", names=np.array(list(sd.keys())),
         shapes=np.array([",".join(map(str, v.shape)) for v in sd.values()]),
         dtypes=np.array([str(v.dtype) for v in sd.values()]),
         n_params=np.int64(sum(p.numel() for p in m.parameters())))


def g1(D3DP):
    """Schedule buffers (fp64) + DDIM time pairs."""
    m = build_ref(D3DP, 9, 32, 1, seed=1)
    arrs = {k: getattr(m, k).numpy() for k in (
        "betas", "alphas_cumprod", "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod",
        "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod")}
    for K in (1, 2, 5, 10, 20):
        times = torch.linspace(-1, 999, steps=K + 1)
        times = list(reversed(times.int().tolist()))
        arrs[f"pairs_K{K}"] = np.array(list(zip(times[:-1], times[1:])), dtype=np.int64)
    save("g1_schedule", **arrs)


def g2(D3DP):
    """Tiny denoiser with per-block intermediates: cs=64, dep=2, F=9, B=2, H=3."""
    cs, dep, Fr, B, H, seed = 64, 2, 9, 2, 3, 11
    m = build_ref(D3DP, Fr, cs, dep, seed)
    x2d = torch.from_numpy(synthetic_inputs_2d(101, B, Fr))
    x3d = torch.from_numpy(synthetic_noise(102, (B, H, Fr, 17, 3)))
    t = torch.tensor([999, 37], dtype=torch.long)
    taps = {"s": [], "t": []}
    pe = m.pose_estimator
    h1 = pe.Spatial_norm.register_forward_hook(lambda mod, i, o: taps["s"].append(o.detach().clone()))
    h2 = pe.Temporal_norm.register_forward_hook(lambda mod, i, o: taps["t"].append(o.detach().clone()))
    with torch.no_grad():
        out = pe(x2d, x3d, t)
    h1.remove(); h2.remove()
    arrs = dict(cs=cs, dep=dep, frames=Fr, seed=seed, x2d=x2d.numpy(), x3d=x3d.numpy(), t=t.numpy(),
                out=out.numpy())
    BH = B * H
    for i in range(dep):
        arrs[f"ste{i}"] = taps["s"][i].reshape(BH, Fr, 17, cs).numpy()                       # (bh f) n c
        arrs[f"tte{i}"] = taps["t"][i].reshape(BH, 17, Fr, cs).permute(0, 2, 1, 3).contiguous().numpy()
    save("g2_tiny_denoiser", **arrs)


def g3(D3DP):
    """Full-width single denoiser calls: cs=512, dep=8, F in {27, 243}, B=1, H=1."""
    for Fr in (27, 243):
        seed = 7
        m = build_ref(D3DP, Fr, 512, 8, seed)
        x2d = torch.from_numpy(synthetic_inputs_2d(201, 1, Fr))
        x3d = torch.from_numpy(synthetic_noise(202, (1, 1, Fr, 17, 3)))
        arrs = dict(cs=512, dep=8, frames=Fr, seed=seed, x2d_seed=201, x3d_seed=202)
        for tt in (999, 499, 99):
            with torch.no_grad():
                arrs[f"out_t{tt}"] = m.pose_estimator(x2d, x3d, torch.tensor([tt])).numpy()
        save(f"g3_denoiser_F{Fr}", **arrs)


def run_sampler(D3DP, Fr, cs, dep, seed, B, H, K, x2d_seed, noise_seed):
    m = build_ref(D3DP, Fr, cs, dep, seed, H=H, K=K)
    x2d = synthetic_inputs_2d(x2d_seed, B, Fr)
    x2d_flip = flip_2d(x2d)
    noises = [torch.from_numpy(synthetic_noise(noise_seed + k, (B, H, Fr, 17, 3))) for k in range(K)]
    with Draws(randn_list=noises) as d:
        with torch.no_grad():
            out = m(torch.from_numpy(x2d), None, input_2d_flip=torch.from_numpy(x2d_flip))
        assert d.n_randn == K and not d.randn_list, (d.n_randn, K)
    return out.numpy()


def g4(D3DP):
    """Sampler: BASELINE config 1 exactly (F=27,H=1,K=1,B=2) and F=27,H=3,K=5,B=2; plus a tiny
    model with scale != 1 to pin the clamp/scale handling."""
    out = run_sampler(D3DP, 27, 512, 8, 7, B=2, H=1, K=1, x2d_seed=301, noise_seed=400)
    save("g4_sampler_c1", cs=512, dep=8, frames=27, seed=7, B=2, H=1, K=1, x2d_seed=301, noise_seed=400, out=out)
    out = run_sampler(D3DP, 27, 512, 8, 7, B=2, H=3, K=5, x2d_seed=302, noise_seed=500)
    save("g4_sampler_H3K5", cs=512, dep=8, frames=27, seed=7, B=2, H=3, K=5, x2d_seed=302, noise_seed=500, out=out)
    out = run_sampler(D3DP, 9, 64, 2, 11, B=3, H=4, K=10, x2d_seed=303, noise_seed=600)
    save("g4_sampler_tiny_K10", cs=64, dep=2, frames=9, seed=11, B=3, H=4, K=10, x2d_seed=303, noise_seed=600, out=out)


def g13(D3DP):
    """BASELINE config 2 at FULL size through the reference: F=243, J=17, H=5, K=5, B=4, cs=512, dep=8 (200
    (clip, hypothesis) denoiser passes of 295 GFLOP each: minutes on the host).  The (4,5,5,243,17,3) output is 5 MB, so
    the fixture keeps every 10th frame in full plus, for every (clip, step, hypothesis), fp64 checksums over ALL frames:
    sum, sum of squares and a position-weighted sum (catches permuted frames/joints)."""
    B, H, K, Fr = 4, 5, 5, 243
    out = run_sampler(D3DP, Fr, 512, 8, 7, B=B, H=H, K=K, x2d_seed=1301, noise_seed=1400)
    o = out.astype(np.float64).reshape(B, K, H, -1)
    w = np.cos(np.arange(o.shape[-1], dtype=np.float64) * 0.37) + 1.5
    frames = np.arange(0, Fr, 10)
    save("g13_sampler_c2", cs=512, dep=8, frames=Fr, seed=7, B=B, H=H, K=K, x2d_seed=1301, noise_seed=1400,
         kept_frames=frames, out_kept=out[:, :, :, frames], sum=o.sum(-1), sumsq=(o * o).sum(-1), wsum=(o * w).sum(-1))


def g14(D3DP):
    """BASELINE config 3 -- the benchmarked workload -- at FULL size through the reference: F=243, J=17, H=20, K=10,
    B=16, cs=512, dep=8 (6400 (clip, hypothesis) denoiser passes of 295 GFLOP each: about an hour and a half on 8 host
    cores, run clip by clip -- clips are independent given their 2D input, mixste.py:227-230 -- and resumable through
    /tmp/g14_clip*.npy).  The (16,10,20,243,17,3) output is 159 MB, so the fixture keeps, for EVERY (clip, step,
    hypothesis), fp64 checksums over all frames: sum, sum of squares and four Gaussian random projections (weights from
    PCG64(1414), regenerated by the test).  E[(w . e)^2] = |e|^2 for such weights, so the projections measure the RMS
    error of every slice, not only its mean: tests/test_hip_parity.py::test_c3_full_size_all_slices_vs_reference_fixture."""
    B, H, K, Fr = 16, 20, 10, 243
    x2d = synthetic_inputs_2d(1234, B, Fr)
    noises = [synthetic_noise(2000 + k, (B, H, Fr, 17, 3)) for k in range(K)]
    m = build_ref(D3DP, Fr, 512, 8, 7, H=H, K=K)
    n = H and Fr * 17 * 3
    w = np.random.Generator(np.random.PCG64(1414)).standard_normal(size=(4, n))
    sums, sqs, projs = np.zeros((B, K, H)), np.zeros((B, K, H)), np.zeros((B, K, H, 4))
    for b in range(B):
        tmp = f"/tmp/g14_clip{b}.npy"
        if os.path.exists(tmp):
            out = np.load(tmp)
        else:
            xb = x2d[b:b + 1]
            with Draws(randn_list=[torch.from_numpy(nz[b:b + 1]) for nz in noises]) as d:
                with torch.no_grad():
                    out = m(torch.from_numpy(xb), None, input_2d_flip=torch.from_numpy(flip_2d(xb))).numpy()
                assert d.n_randn == K and not d.randn_list
            np.save(tmp, out)
            print(f"  clip {b} done", flush=True)
        o = out.astype(np.float64).reshape(K, H, n)
        sums[b], sqs[b], projs[b] = o.sum(-1), (o * o).sum(-1), o @ w.T
    save("g14_sampler_c3", cs=512, dep=8, frames=Fr, seed=7, B=B, H=H, K=K, x2d_seed=1234, noise_seed=2000,
         proj_seed=1414, sum=sums, sumsq=sqs, proj=projs)


def g6(D3DP):
    """Train step forward (+loss, grad norms): F=27, B=4, cs=64, dep=2; DropPath off and on."""
    sys.path.insert(0, REF)
    from common.loss import mpjpe
    cs, dep, Fr, B, seed = 64, 2, 27, 4, 13
    x2d = torch.from_numpy(synthetic_inputs_2d(701, B, Fr))
    gt = torch.from_numpy(synthetic_noise(702, (B, Fr, 17, 3))) * 0.3
    gt[:, :, 0] = 0
    ts = [torch.tensor([v], dtype=torch.long) for v in (3, 250, 640, 999)]
    ns = [torch.from_numpy(synthetic_noise(710 + i, (Fr, 17, 3))) for i in range(B)]
    arrs = dict(cs=cs, dep=dep, frames=Fr, seed=seed, x2d=x2d.numpy(), gt=gt.numpy(),
                t=np.array([int(v) for v in ts]), noise=np.stack([n.numpy() for n in ns]))
    for tag in ("nodrop", "drop"):
        m = build_ref(D3DP, Fr, cs, dep, seed, is_train=True)
        _DropPathStub.log = []
        if tag == "drop":
            rates = [x.item() for x in torch.linspace(0, 0.1, dep)]
            rng = np.random.Generator(np.random.PCG64(720))
            inj = []
            # call order inside MixSTE2.forward: STE0(attn, mlp), TTE0(attn, mlp), STE1..., TTE1...
            for i in range(dep):
                for S in (B * Fr, B * 17):
                    for _ in range(2):
                        if rates[i] == 0.0:
                            continue
                        keep = 1 - rates[i]
                        mask = (rng.uniform(size=(S, 1, 1)) < keep).astype(np.float32) / keep
                        inj.append(torch.from_numpy(mask))
            _DropPathStub.injected = list(inj)
            arrs["drop_masks_n"] = len(inj)
            for k, mk in enumerate(inj):
                arrs[f"drop_mask{k}"] = mk.numpy()
        else:
            _DropPathStub.injected = None
        rl = []
        for i in range(B):
            rl += [ns[i]]
        with Draws(randn_list=rl, randint_list=list(ts)):
            pred = m(x2d, gt)
        loss = mpjpe(pred, gt)
        loss.backward(loss.clone().detach())        # main.py:393
        arrs[f"pred_{tag}"] = pred.detach().numpy()
        arrs[f"loss_{tag}"] = np.float64(loss.item())
        for pn in ("pose_estimator.head.1.weight", "pose_estimator.STEblocks.0.attn.qkv.weight",
                   "pose_estimator.TTEblocks.1.mlp.fc2.weight", "pose_estimator.Temporal_pos_embed"):
            g = dict(m.named_parameters())[pn].grad
            arrs[f"gradnorm_{tag}::{pn}"] = np.float64(g.double().norm().item())
        _DropPathStub.injected = None
    save("g6_train_step", **arrs)


def g5(D3DP):
    """Caller side (rows N1/N2): 60-frame synthetic sequence at F=27 -> eval_data_prepare-style
    chunking (last clip = last F frames), root zeroing, trajectory add, project_to_2d, and the four
    per-step metrics of loss.py:22-107.  main.py cannot be imported (tensorboard, datasets), so the
    chunking expectation is produced by running the reference's function body semantics through
    common.camera / common.loss only; chunk indices are recorded as data."""
    from common.camera import project_to_2d
    from common.loss import (mpjpe_diffusion, mpjpe_diffusion_all_min, mpjpe_diffusion_reproj)
    Fr, N, H, K, B = 27, 60, 3, 2, 3
    rng = np.random.Generator(np.random.PCG64(801))
    seq3d = (rng.standard_normal((N, 17, 3)) * 0.3).astype(np.float32)
    seq3d[:, :, 2] += 4.0                      # in front of the camera
    seq2d = rng.uniform(-1, 1, (N, 17, 2)).astype(np.float32)
    cam = np.array([[2.29, 2.287, 0.0254, 0.0289, -0.2070, 0.2477, -0.0030, -0.0009, -0.0014]], np.float32)
    # chunk starts per main.py:267-299 (N=60,F=27 -> clips [0:27],[27:54],[33:60])
    starts = np.array([0, 27, N - Fr], dtype=np.int64)
    inputs_3d = torch.from_numpy(np.stack([seq3d[s:s + Fr] for s in starts]))
    inputs_2d = torch.from_numpy(np.stack([seq2d[s:s + Fr] for s in starts]))
    traj = inputs_3d[:, :, :1].clone()
    inputs_3d[:, :, 0] = 0
    pred = torch.from_numpy((rng.standard_normal((B, K, H, Fr, 17, 3)) * 0.3).astype(np.float32))
    pred_in = pred.clone()
    pred[:, :, :, :, 0] = 0                    # main.py:700
    b, t_, h, f, j, c = pred.shape
    absol = pred + traj.unsqueeze(1).unsqueeze(1).repeat(1, t_, h, 1, 1, 1)
    reproj = project_to_2d(absol.reshape(b * t_ * h * f, j, c), torch.from_numpy(cam).repeat(b * t_ * h * f, 1))
    reproj = reproj.reshape(b, t_, h, f, j, 2)
    e_jbest = mpjpe_diffusion_all_min(pred, inputs_3d)
    e_pbest = mpjpe_diffusion(pred, inputs_3d)
    e_pagg = mpjpe_diffusion_all_min(pred, inputs_3d, mean_pos=True)
    e_jagg = mpjpe_diffusion_reproj(pred, inputs_3d, reproj, inputs_2d)
    save("g5_caller", frames=Fr, seq3d=seq3d, seq2d=seq2d, cam=cam, starts=starts, pred=pred_in.numpy(),
         reproj=reproj.numpy(), e_jbest=e_jbest.numpy(), e_pbest=e_pbest.numpy(), e_pagg=e_pagg.numpy(),
         e_jagg=e_jagg.numpy())



def _ref_function(path, name, extra_globals):
    """Compile ONE function of a reference script that cannot be imported as a module (main.py executes dataset
    loading on import) by lifting its FunctionDef node out of the parsed file, and return the live function.  This
    runs the reference's own code object here; no reference text is stored anywhere."""
    import ast
    tree = ast.parse(open(os.path.join(REF, path)).read())
    node = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == name)
    mod = ast.Module(body=[node], type_ignores=[])
    ns = dict(extra_globals)
    exec(compile(mod, os.path.join(REF, path), "exec"), ns)
    return ns[name]


def g7(D3DP):
    """N2 clip chunking: main.py:267-299 eval_data_prepare on sequences of 60 (ragged), 54 (exact), 27, 20 (short)
    frames at F=27."""
    from einops import rearrange
    fn = _ref_function("main.py", "eval_data_prepare", {"torch": torch, "rearrange": rearrange})
    rng = np.random.Generator(np.random.PCG64(811))
    arrs = {"frames": 27}
    for n in (60, 54, 27, 20, 1, 100):
        s2 = rng.uniform(-1, 1, (1, n, 17, 2)).astype(np.float32)
        s3 = rng.standard_normal((1, n, 17, 3)).astype(np.float32)
        if n == 1:      # torch.squeeze drops the frame axis of a 1-frame sequence in the reference: not a valid input
            continue
        c2, c3 = fn(27, torch.from_numpy(s2), torch.from_numpy(s3))
        arrs[f"seq2d_{n}"], arrs[f"seq3d_{n}"] = s2[0], s3[0]
        arrs[f"clips2d_{n}"], arrs[f"clips3d_{n}"] = c2.numpy(), c3.numpy()
    save("g7_clips", **arrs)


def _gen_dataset(seed, lengths):
    rng = np.random.Generator(np.random.PCG64(seed))
    p2 = [rng.uniform(-1, 1, (n, 17, 2)).astype(np.float32) for n in lengths]
    p3 = [(rng.standard_normal((n, 17, 3)) * 0.3).astype(np.float32) for n in lengths]
    cams = [rng.uniform(-1, 1, (9,)).astype(np.float32) for _ in lengths]
    return cams, p3, p2


def g8(D3DP):
    """N3 batches: common/generators.py ChunkedGenerator_Seq, F=27, batch 4, shuffle + flip augmentation, two epochs
    (so the RandomState carries over), plus an unshuffled/unaugmented pass with cameras."""
    from common.generators import ChunkedGenerator_Seq
    lengths = [70, 27, 100, 20]
    cams, p3, p2 = _gen_dataset(821, lengths)
    kl, kr = H36M_JOINTS_LEFT, H36M_JOINTS_RIGHT
    gen = ChunkedGenerator_Seq(4, cams, p3, p2, 27, pad=0, causal_shift=0, shuffle=True, augment=True,
                               kps_left=kl, kps_right=kr, joints_left=kl, joints_right=kr)
    arrs = {"lengths": np.array(lengths), "frames": 27, "batch": 4, "seed": 821, "num_batches": gen.batch_num(),
            "num_frames": gen.num_frames()}
    k = 0
    for ep in range(2):
        for cam, b3, b2 in gen.next_epoch():
            arrs[f"cam_{k}"], arrs[f"b3_{k}"], arrs[f"b2_{k}"] = cam.copy(), b3.astype(np.float32), b2.astype(np.float32)
            k += 1
    arrs["n_batches_total"] = k
    gen2 = ChunkedGenerator_Seq(3, None, p3, p2, 27, shuffle=False, augment=False)
    j = 0
    for _, b3, b2 in gen2.next_epoch():
        arrs[f"plain_b3_{j}"], arrs[f"plain_b2_{j}"] = b3.astype(np.float32), b2.astype(np.float32)
        j += 1
    arrs["plain_batches"] = j
    save("g8_batches", **arrs)


def g9(D3DP):
    """N4 Protocol #2: common/loss.py p_mpjpe (one pose set) and the four *_diffusion variants main.py:726-729 logs,
    on the g5 tensors."""
    from common.camera import project_to_2d
    from common.loss import (p_mpjpe, p_mpjpe_diffusion, p_mpjpe_diffusion_all_min, p_mpjpe_diffusion_reproj)
    g5 = np.load(os.path.join(OUT, "g5_caller.npz"))
    Fr = int(g5["frames"])
    starts = g5["starts"]
    inputs_3d = torch.from_numpy(np.stack([g5["seq3d"][s:s + Fr] for s in starts]))
    inputs_2d = torch.from_numpy(np.stack([g5["seq2d"][s:s + Fr] for s in starts]))
    inputs_3d[:, :, 0] = 0
    pred = torch.from_numpy(g5["pred"]).clone()
    pred[:, :, :, :, 0] = 0
    reproj = torch.from_numpy(g5["reproj"])
    with Draws():
        e_jbest = p_mpjpe_diffusion_all_min(pred, inputs_3d)
        e_pbest = p_mpjpe_diffusion(pred, inputs_3d)
        e_pagg = p_mpjpe_diffusion_all_min(pred, inputs_3d, mean_pos=True)
        e_jagg = p_mpjpe_diffusion_reproj(pred, inputs_3d, reproj, inputs_2d)
    single = p_mpjpe(pred[:, 0, 0].reshape(-1, 17, 3).numpy(), inputs_3d.reshape(-1, 17, 3).numpy())
    # a reflected target: exercises the det(R) < 0 branch (loss.py:218-222)
    refl = inputs_3d.clone()
    refl[..., 0] *= -1
    e_refl = p_mpjpe_diffusion(pred, refl)
    save("g9_pmpjpe", e_jbest=np.asarray(e_jbest), e_pbest=np.asarray(e_pbest), e_pagg=np.asarray(e_pagg),
         e_jagg=np.asarray(e_jagg), single=np.float64(single), e_pbest_reflected=np.asarray(e_refl))


def g10(D3DP):
    """N3 training loop: reference model + ChunkedGenerator_Seq + optim.AdamW(weight_decay=0.1) + mpjpe +
    backward(loss.detach()) + per-epoch lr decay (main.py:311-401, 519-522): cs=64, dep=2, F=27, batch 4, two epochs
    of four iterations, DropPath inactive (recorded t / noise draws)."""
    from common.generators import ChunkedGenerator_Seq
    from common.loss import mpjpe
    import torch.optim as optim
    cs, dep, Fr, seed = 64, 2, 27, 17
    lengths = [70, 27, 100]
    cams, p3, p2 = _gen_dataset(831, lengths)
    kl, kr = H36M_JOINTS_LEFT, H36M_JOINTS_RIGHT
    gen = ChunkedGenerator_Seq(4, cams, p3, p2, Fr, shuffle=True, augment=True, kps_left=kl, kps_right=kr,
                               joints_left=kl, joints_right=kr)
    m = build_ref(D3DP, Fr, cs, dep, seed, is_train=True)
    m.train()
    _DropPathStub.injected = None
    lr, lr_decay = 1e-4, 0.9
    opt = optim.AdamW(m.parameters(), lr=lr, weight_decay=0.1)
    rng = np.random.Generator(np.random.PCG64(832))
    arrs = dict(cs=cs, dep=dep, frames=Fr, seed=seed, lengths=np.array(lengths), data_seed=831, lr=lr, lr_decay=lr_decay)
    losses, it = [], 0
    for epoch in range(2):
        for _, b3, b2 in gen.next_epoch():
            inputs_3d = torch.from_numpy(b3.astype("float32"))
            inputs_2d = torch.from_numpy(b2.astype("float32"))
            inputs_3d[:, :, 0] = 0
            B = inputs_3d.shape[0]
            ts = [torch.tensor([int(rng.integers(0, 1000))], dtype=torch.long) for _ in range(B)]
            ns = [torch.from_numpy(rng.standard_normal((Fr, 17, 3)).astype(np.float32)) for _ in range(B)]
            arrs[f"t_{it}"] = np.array([int(v) for v in ts])
            arrs[f"noise_{it}"] = np.stack([n.numpy() for n in ns])
            opt.zero_grad()
            with Draws(randn_list=list(ns), randint_list=list(ts)):
                pred = m(inputs_2d, inputs_3d)
            loss = mpjpe(pred, inputs_3d)
            loss.backward(loss.clone().detach())
            losses.append(loss.item())
            opt.step()
            it += 1
        lr *= lr_decay
        for g in opt.param_groups:
            g["lr"] *= lr_decay
    arrs["losses"] = np.array(losses, dtype=np.float64)
    arrs["final_lr"] = np.float64(opt.param_groups[0]["lr"])
    sd = m.state_dict()
    for pn in ("pose_estimator.head.1.weight", "pose_estimator.STEblocks.0.attn.qkv.weight", "pose_estimator.TTEblocks.1.mlp.fc2.bias",
               "pose_estimator.Temporal_pos_embed", "pose_estimator.Spatial_norm.weight", "pose_estimator.time_mlp.1.weight"):
        arrs[f"param::{pn}"] = sd[pn].detach().numpy().copy()
    st = opt.state_dict()["state"]
    arrs["n_state"] = len(st)
    arrs["step0"] = np.float64(float(st[0]["step"]))
    arrs["exp_avg_sq_sum"] = np.float64(sum(float(v["exp_avg_sq"].double().sum()) for v in st.values()))
    arrs["exp_avg_abs_sum"] = np.float64(sum(float(v["exp_avg"].double().abs().sum()) for v in st.values()))
    save("g10_train_loop", **arrs)



def g11(D3DP):
    """N4 3DHP caller side: common/loss.py mpjpe_diffusion_3dhp (valid-frame mask, both modes), camera.py
    project_to_2d_linear / image_coordinates, and main_3dhp.py's cam_mm_to_pix + pose_post_process (lifted by AST),
    on synthetic millimetre poses (3 clips of F=27 cut from a 60-frame sequence)."""
    from common.camera import image_coordinates, project_to_2d, project_to_2d_linear
    from common.loss import mpjpe_diffusion_3dhp
    cam_mm_to_pix = _ref_function("main_3dhp.py", "cam_mm_to_pix", {})
    pose_post_process = _ref_function("main_3dhp.py", "pose_post_process", {})
    rng = np.random.Generator(np.random.PCG64(841))
    B, K, H, Fr, N = 3, 2, 4, 27, 60
    gt = (rng.standard_normal((B, Fr, 17, 3)) * 300).astype(np.float32)
    traj = (rng.standard_normal((B, Fr, 1, 3)) * 100 + np.array([0, 0, 4000.0])).astype(np.float32)
    gt[:, :, 14] = 0
    pred = (gt[:, None, None] + rng.standard_normal((B, K, H, Fr, 17, 3)) * 40).astype(np.float32)
    pred[:, :, :, :, 14] = 0
    valid = (rng.uniform(size=(B, Fr, 1)) < 0.8).astype(np.float32)
    x2d = rng.uniform(-1, 1, (B, Fr, 17, 2)).astype(np.float32)
    cam1 = cam_mm_to_pix(torch.tensor([7.32506, 7.32506, -0.0322884, 0.0929296, 0, 0, 0, 0, 0]), [2048, 2048, 10, 10])
    cam2 = cam_mm_to_pix(torch.tensor([8.770747185, 8.770747185, -0.104908645, 0.104899704, 0, 0, 0, 0, 0]),
                         [1920, 1080, 10, 5.625])
    P, G, V = torch.from_numpy(pred), torch.from_numpy(gt), torch.from_numpy(valid)
    e_pbest = mpjpe_diffusion_3dhp(P, G, V.type(torch.bool))
    e_pagg = mpjpe_diffusion_3dhp(P, G, V.type(torch.bool), mean_pos=True)
    absol = (P + torch.from_numpy(traj)[:, None, None]).reshape(-1, 17, 3)
    rp_lin = project_to_2d_linear(absol, cam1.unsqueeze(0).repeat(absol.shape[0], 1)).reshape(B, K, H, Fr, 17, 2)
    rp_dist = project_to_2d(absol, cam2.unsqueeze(0).repeat(absol.shape[0], 1)).reshape(B, K, H, Fr, 17, 2)
    tgt_pix = image_coordinates(x2d[..., :2], w=2048, h=2048)
    # stitching: (n_clips, K, F, J, 3) -> (3, J, N, K)
    sel = pred[:, :, 0]
    stitched = pose_post_process(sel, {"TS1": np.zeros((K, N, 17, 3))}, "TS1", Fr)["TS1"]
    save("g11_3dhp", pred=pred, gt=gt, traj=traj, valid=valid, x2d=x2d, cam1=cam1.numpy(), cam2=cam2.numpy(),
         e_pbest=e_pbest.numpy(), e_pagg=e_pagg.numpy(), reproj_linear=rp_lin.numpy(), reproj_dist=rp_dist.numpy(),
         target_pix=np.asarray(tgt_pix, dtype=np.float32), stitched=stitched.astype(np.float32), n_frames=N, frames=Fr)


def g12(D3DP):
    """N4 3DHP sampler: common/diffusionpose_3dhp.py D3DP, F=27, B=2, H=2, K=2, recorded noise -> millimetre output,
    plus one training-branch forward (targets in millimetres)."""
    from common.diffusionpose_3dhp import D3DP as D3DP_3DHP
    Fr, cs, dep, seed, B, H, K = 27, 64, 2, 19, 2, 2, 2
    jl, jr = [5, 6, 7, 11, 12, 13], [2, 3, 4, 8, 9, 10]
    torch.manual_seed(0)
    m = D3DP_3DHP(make_args(Fr, cs, dep), jl, jr, is_train=False, num_proposals=H, sampling_timesteps=K)
    m.load_state_dict(make_state_dict(seed, cs, dep, Fr), strict=False)
    m.eval()
    x2d = torch.from_numpy(synthetic_inputs_2d(851, B, Fr))
    x2f = x2d.clone()
    x2f[..., 0] *= -1
    x2f[:, :, jl + jr] = x2f[:, :, jr + jl]
    noises = [torch.from_numpy(synthetic_noise(860 + i, (B, H, Fr, 17, 3))) for i in range(K)]
    with Draws(randn_list=[n.clone() for n in noises]), torch.no_grad():
        out = m(x2d, None, input_2d_flip=x2f)
    mt = D3DP_3DHP(make_args(Fr, cs, dep), jl, jr, is_train=True)
    mt.load_state_dict(make_state_dict(seed, cs, dep, Fr), strict=False)
    mt.train()
    _DropPathStub.injected = None
    gt_mm = torch.from_numpy(synthetic_noise(870, (B, Fr, 17, 3))) * 300
    ts = [torch.tensor([v], dtype=torch.long) for v in (17, 803)]
    ns = [torch.from_numpy(synthetic_noise(880 + i, (Fr, 17, 3))) for i in range(B)]
    with Draws(randn_list=[n.clone() for n in ns], randint_list=list(ts)), torch.no_grad():
        tr = mt(x2d, gt_mm)
    save("g12_3dhp_sampler", frames=Fr, cs=cs, dep=dep, seed=seed, x2d=x2d.numpy(), x2d_flip=x2f.numpy(),
         noise=np.stack([n.numpy() for n in noises]), out=out.numpy(), gt_mm=gt_mm.numpy(),
         t=np.array([int(v) for v in ts]), train_noise=np.stack([n.numpy() for n in ns]), train_out=tr.numpy(),
         joints_left=np.array(jl), joints_right=np.array(jr))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="g0,g1,g2,g3,g4,g5,g6,g7,g8,g9,g10,g11,g12,g13")
    a = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    D3DP = import_reference()
    for name in a.only.split(","):
        print(name)
        globals()[name](D3DP)


if __name__ == "__main__":
    main()
