"""Host-side mirror of the reference's plugin interface for the hot path.

``D3DP`` and ``MixSTE2`` keep the reference's constructor/forward signatures, parameter and
buffer names (reference common/diffusionpose.py:60-126, common/mixste.py:141-210), so the
reference's callers (main.py:228-257, 450, 698; in_the_wild/utils.py:284) and its checkpoints work
unchanged -- but no arithmetic happens here: parameters are plain containers and every forward
goes through libd3dp_hip.so (include/d3dp_hip.h).  There is no PyTorch/CPU fallback.

Extensions over the reference API (default-off):
  * ``noise=[...]`` / ``generator=`` on the samplers for fixed-noise parity tests;
  * ``numerics='exact'|'fast'`` (or ``args.numerics`` / env ``D3DP_NUMERICS``): exact = fp32 MFMA
    (<= 1e-3 mm vs the reference), fast = bf16 MFMA with fp32 accumulation.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import threading
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F
from torch import nn

from . import _lib

__all__ = ["D3DP", "MixSTE2", "cosine_beta_schedule"]

NUM_JOINTS = 17


def cosine_beta_schedule(timesteps: int, s: float = 0.008) -> torch.Tensor:
    """fp64 cosine schedule, reference common/diffusionpose.py:42-52."""
    steps = timesteps + 1
    x = torch.linspace(0, timesteps, steps, dtype=torch.float64)
    ac = torch.cos(((x / timesteps) + s) / (1 + s) * math.pi * 0.5) ** 2
    ac = ac / ac[0]
    return torch.clip(1 - (ac[1:] / ac[:-1]), 0, 0.999)


def _resolve_mode(numerics: Optional[str]) -> int:
    name = (numerics or os.environ.get("D3DP_NUMERICS", "exact")).lower()
    if name not in ("exact", "fast", "train"):
        raise ValueError(f"numerics must be 'exact', 'fast' or 'train', got {name!r}")
    return {"fast": _lib.MODE_FAST, "exact": _lib.MODE_EXACT, "train": _lib.MODE_TRAIN}[name]


class _TrainStep(torch.autograd.Function):
    """Autograd bridge for the training step: forward = d3dp_train_forward (activations stay in the library
    workspace), backward = d3dp_train_backward, which fills one gradient buffer per parameter.

    The activations live in ONE per-model workspace, so every forward stamps it with a generation number; a backward
    whose forward is no longer the most recent one (gradient accumulation with deferred backwards, two losses from two
    forwards, a validation forward in between) first re-runs its forward -- same inputs, same DropPath masks, hence the
    same activations -- instead of differentiating through somebody else's."""

    @staticmethod
    def forward(ctx, net, x_2d, x_3d, t, masks, *params):
        ctx.net, ctx.masks = net, masks
        ctx.save_for_backward(x_2d, x_3d, t)
        out = net._train_forward(x_2d, x_3d, t, masks)
        ctx.gen = net._train_gen
        # the library reads the parameters' OWN storage (borrowed weights): remember what the forward saw
        ctx.versions = tuple((p.data_ptr(), p._version) for p in params)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        x_2d, x_3d, t = ctx.saved_tensors
        now = tuple((p.data_ptr(), p._version) for p in ctx.net._weight_tensors())
        if now != ctx.versions:
            # PyTorch raises in the same situation ("one of the variables needed for gradient computation has been
            # modified by an inplace operation"): differentiating with weights the forward never saw is silently wrong
            raise RuntimeError("d3dp_amd: a parameter was modified (optimizer.step(), in-place op or re-allocation) between "
                               "this forward and its backward(); the training kernels read the parameters in place, so the "
                               "gradient would be taken at the NEW weights.  Call backward() before the optimizer step.")
        if ctx.gen != ctx.net._train_gen:
            ctx.net._train_forward(x_2d, x_3d, t, ctx.masks)
        grads = ctx.net._train_backward(x_2d, x_3d, t, ctx.masks, grad_out.contiguous())
        return (None, None, None, None, None) + tuple(grads)


# ----------------------------------------------------------------------------------------------
# parameter containers (names = reference state_dict keys)
# ----------------------------------------------------------------------------------------------
class _Attention(nn.Module):          # mixste.py:46-61
    def __init__(self, dim):
        super().__init__()
        self.qkv = nn.Linear(dim, dim * 3, bias=True)
        self.proj = nn.Linear(dim, dim)


class _Mlp(nn.Module):                # mixste.py:24-35
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)


class _Block(nn.Module):              # mixste.py:84-111
    def __init__(self, dim, hidden, eps):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=eps)
        self.attn = _Attention(dim)
        self.norm2 = nn.LayerNorm(dim, eps=eps)
        self.mlp = _Mlp(dim, hidden)


class _NoParams(nn.Module):
    """Placeholder keeping nn.Sequential indices aligned with the reference (time_mlp.0 / .2)."""


class _DeviceState:
    """What one MixSTE2 holds on ONE device: the library context (bound to that device, include/d3dp_hip.h), the signature
    of the weights it was last given, and the workspaces.  ``nn.DataParallel`` (the reference's multi-GPU caller,
    main.py:242-248) drives one replica per device from one thread each: every device gets its own state."""
    __slots__ = ("ctx", "weights_sig", "keep", "ws", "train_ws", "train_gen")

    def __init__(self):
        self.ctx, self.weights_sig, self.keep, self.ws, self.train_ws, self.train_gen = None, None, None, None, None, 0


class MixSTE2(nn.Module):
    """Denoiser with the reference's signature (mixste.py:141-147, forward :278) running on libd3dp_hip."""

    def __init__(self, num_frame=9, num_joints=17, in_chans=2, embed_dim_ratio=32, depth=4, num_heads=8,
                 mlp_ratio=2., qkv_bias=True, qk_scale=None, drop_rate=0., attn_drop_rate=0., drop_path_rate=0.2,
                 norm_layer=None, is_train=True, numerics: Optional[str] = None, chunk_seqs: int = 0):
        super().__init__()
        if in_chans != 2 or not qkv_bias or qk_scale is not None or drop_rate or attn_drop_rate:
            raise NotImplementedError("libd3dp_hip implements the configuration D3DP instantiates "
                                      "(in_chans=2, qkv_bias=True, no dropout), diffusionpose.py:123-124")
        C_ = embed_dim_ratio
        self.is_train = is_train
        self.num_frame, self.num_joints, self.embed_dim, self.block_depth = num_frame, num_joints, C_, depth
        self.num_heads, self.hidden = num_heads, int(C_ * mlp_ratio)
        self.drop_path_rate = drop_path_rate
        self.eps_block, self.eps_head = 1e-6, 1e-5
        self.Spatial_patch_to_embedding = nn.Linear(in_chans + 3, C_)
        self.Spatial_pos_embed = nn.Parameter(torch.zeros(1, num_joints, C_))
        self.Temporal_pos_embed = nn.Parameter(torch.zeros(1, num_frame, C_))
        self.time_mlp = nn.Sequential(_NoParams(), nn.Linear(C_, C_ * 2), _NoParams(), nn.Linear(C_ * 2, C_))
        self.STEblocks = nn.ModuleList([_Block(C_, self.hidden, self.eps_block) for _ in range(depth)])
        self.TTEblocks = nn.ModuleList([_Block(C_, self.hidden, self.eps_block) for _ in range(depth)])
        self.Spatial_norm = nn.LayerNorm(C_, eps=self.eps_block)
        self.Temporal_norm = nn.LayerNorm(C_, eps=self.eps_block)
        self.head = nn.Sequential(nn.LayerNorm(C_), nn.Linear(C_, 3))
        self._mode = _resolve_mode(numerics)
        self._chunk_seqs = int(chunk_seqs)
        # Per-device library state.  The dict (and its lock) is SHARED between this module and its nn.DataParallel
        # replicas -- a replica is a shallow copy made on every forward (torch/nn/parallel/replicate.py) -- so the context
        # of device d is created once and reused by every later replica on d; only the module that created the dict
        # (`_owns_states`) ever destroys a context, and it does so for all devices.
        self._states: Dict[torch.device, _DeviceState] = {}
        self._states_lock = threading.Lock()
        self._owns_states = True
        self._last_device: Optional[torch.device] = None
        # Dotted names of the weights in nn.Module.parameters() order.  A DataParallel replica has NO parameters
        # (torch/nn/parallel/replicate.py empties `_parameters` and sets the broadcast copies as plain attributes), so
        # everything that needs "the weights of this module" resolves these names by attribute: _weight_tensors().
        self._param_names = tuple(n for n, _ in self.named_parameters())
        self._src_sig = None            # replicas: (data_ptr, version) of the SOURCE module's parameters at replication

    # -- library context ------------------------------------------------------------------------
    @property
    def numerics(self) -> str:
        return {_lib.MODE_FAST: "fast", _lib.MODE_EXACT: "exact", _lib.MODE_TRAIN: "train"}[self._mode]

    def set_numerics(self, numerics: str, chunk_seqs: Optional[int] = None) -> None:
        self._mode = _resolve_mode(numerics)
        if chunk_seqs is not None:
            self._chunk_seqs = int(chunk_seqs)
        self._drop_ctx()

    def _replicate_for_data_parallel(self):
        """nn.DataParallel replica (main.py:242-248 wraps every model in one): shares the per-device states with the
        module it was copied from and owns none of them -- dropping a replica must never free a context."""
        replica = super()._replicate_for_data_parallel()
        replica._owns_states = False
        # What the replica's weights are copies OF: its own tensors are fresh broadcast copies on every forward (whose
        # addresses the caching allocator recycles), so "did the weights change" can only be read off the source.
        replica._src_sig = tuple((p.data_ptr(), p._version) for p in self._weight_tensors())
        return replica

    def _weight_tensors(self) -> List[torch.Tensor]:
        """The weight tensors in ``parameters()`` order, found by attribute -- on the module itself these ARE
        ``list(self.parameters())``; on an nn.DataParallel replica (whose ``parameters()`` is empty) they are the
        per-forward broadcast copies, which carry the autograd edge back to the source parameters."""
        out = []
        for name in self._param_names:
            obj = self
            for part in name.split("."):
                obj = getattr(obj, part)
            out.append(obj)
        return out

    def _is_replica_module(self) -> bool:
        return bool(self.__dict__.get("_is_replica", False)) or not self.__dict__.get("_owns_states", True)

    def __getstate__(self):
        """copy.deepcopy / pickle: library handles do not travel; the copy starts without contexts and owns its own."""
        d = self.__dict__.copy()
        d["_states"], d["_states_lock"], d["_owns_states"], d["_last_device"] = {}, None, True, None
        return d

    def __setstate__(self, d):
        super().__setstate__(d)
        self._states_lock = threading.Lock()
        if "_param_names" not in self.__dict__:       # pickles of whole modules saved before round 5
            self._param_names = tuple(n for n, _ in self.named_parameters())
        self.__dict__.setdefault("_src_sig", None)
        self.__dict__.pop("_droppath_keep", None)      # (a cached device tensor; rebuilt on the next training step)

    def _drop_ctx(self):
        """Destroy every device's context (owner only; a replica just forgets nothing -- the states are not its own)."""
        if not self.__dict__.get("_owns_states", False):
            return
        with self._states_lock:
            states = list(self._states.values())
            self._states.clear()
        for st in states:
            if st.ctx is not None:
                _lib.load().d3dp_destroy(st.ctx)
                st.ctx = None

    def __del__(self):
        try:
            self._drop_ctx()
        except Exception:
            pass

    def _state(self, device: Optional[torch.device] = None) -> _DeviceState:
        """The state of `device` (default: the device of the most recent call)."""
        device = device if device is not None else self._last_device
        st = self._states.get(device) if device is not None else None
        assert st is not None and st.ctx is not None, "run one forward (or call _context(device)) first"
        return st

    # single-device view of the state, kept for callers and tests that predate the per-device dict
    @property
    def _ctx(self):
        st = self._states.get(self._last_device) if self._last_device is not None else None
        return st.ctx if st is not None else None

    @property
    def _train_gen(self):
        st = self._states.get(self._last_device) if self._last_device is not None else None
        return st.train_gen if st is not None else 0

    def _context(self, device: torch.device):
        if device.type != "cuda":
            raise _lib.D3DPHipError("MixSTE2 runs only on an MI355X (tensor on %s); there is no CPU fallback" % device)
        if device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        lib = _lib.load()
        with self._states_lock:
            st = self._states.get(device)
            if st is None:
                st = self._states[device] = _DeviceState()
            if st.ctx is None:
                cfg = _lib.Cfg(self.num_frame, self.num_joints, self.embed_dim, self.block_depth, self.num_heads,
                               self.hidden, self.eps_block, self.eps_head, self._mode, self._chunk_seqs)
                h = C.c_void_p()
                with torch.cuda.device(device):
                    _lib.check(lib.d3dp_create(C.byref(cfg), C.byref(h)), "d3dp_create")
                st.ctx = h
        self._last_device = device
        # TRAIN mode reads the parameters' own storage (no packed copy): only a re-allocation changes anything.
        # EXACT / FAST keep packed copies, refreshed when a parameter is replaced or modified through autograd-visible
        # in-place ops; writes that bypass the version counter (p.data.mul_(), EMA code) need refresh_weights().
        params = self._weight_tensors()
        if self._is_replica_module():
            # nn.DataParallel replica: never borrow (its tensors die with the forward), and key the packed / copied weights
            # of this device on the SOURCE parameters -- load_state_dict and optimizer steps on the wrapped module bump
            # their versions, so the next replica on this device re-packs (it never did).
            borrowed = False
            sig = ("replica", self._src_sig if self._src_sig is not None
                   else tuple((p.data_ptr(), p._version) for p in params))
        else:
            borrowed = self._mode == _lib.MODE_TRAIN and len(params) > 0 and all(
                p.device == device and p.dtype == torch.float32 and p.is_contiguous() for p in params)
            sig = (borrowed,) + tuple((p.data_ptr(),) if borrowed else (p.data_ptr(), p._version) for p in params)
        if sig != st.weights_sig:
            self._push_weights(st, device, borrowed)
            st.weights_sig = sig
        return st.ctx

    # -- EXACT mode's operand range (include/d3dp_hip.h: d3dp_exact_range_bound / d3dp_status) ------------------
    SPLIT_RANGE = 4094.0

    def exact_range_bound(self) -> float:
        """Upper bound, provable from the current weights alone, of the magnitude any split-fp16 operand of EXACT mode
        can take for ANY input.  Below ``SPLIT_RANGE`` every operand uses the default scale 2^4; above it the library
        lowers the scale of the blocks concerned (``exact_scales``) -- either way no operand can overflow.  0.0 in the other
        modes."""
        b = C.c_float()
        _lib.check(_lib.load().d3dp_exact_range_bound(self._state().ctx, C.byref(b)), "d3dp_exact_range_bound")
        return float(b.value)

    def nonfinite_seen(self) -> bool:
        """True if any denoiser call since the last query produced inf / nan (synchronises the device; resets)."""
        v = C.c_int32()
        _lib.check(_lib.load().d3dp_status(self._state().ctx, C.byref(v)), "d3dp_status")
        return bool(v.value)

    def exact_scales(self):
        """EXACT mode: ``(s_kv, s_hidden, implementation)`` -- per block (STE 0..depth-1, then TTE) the power-of-two scales of the
        q / k / v / attention-output and MLP-hidden operands chosen from the proven range (16.0 unless a bound asked for
        less), and the implementation in use ('f16x2', 'bf16x3' -- the automatic fallback when a LayerNorm's own output bound
        leaves the range --, 'f32'; None outside EXACT mode).  include/d3dp_hip.h: d3dp_exact_scales."""
        n = 2 * self.block_depth
        kv, hd, impl = (C.c_float * n)(), (C.c_float * n)(), C.c_int32()
        _lib.check(_lib.load().d3dp_exact_scales(self._state().ctx, kv, hd, C.byref(impl)), "d3dp_exact_scales")
        return list(kv), list(hd), {0: "f16x2", 1: "bf16x3", 2: "f32"}.get(impl.value)

    def refresh_weights(self) -> None:
        """Re-pack the library's weight copies on the next call.  Needed only after parameter writes that bypass
        PyTorch's version counter (``p.data.copy_()``, ``p.data.mul_()``, weight averaging through ``.data``):
        ``load_state_dict``, optimizer steps and ordinary in-place ops are picked up automatically."""
        with self._states_lock:
            for st in self._states.values():
                st.weights_sig = None

    def _push_weights(self, st, device, borrowed=False):
        lib = _lib.load()
        keep: List[torch.Tensor] = []

        def dev(t):
            t = t.detach().to(device=device, dtype=torch.float32).contiguous()
            keep.append(t)
            return t.data_ptr()

        def blocks(mods):
            arr = (_lib.BlockWeights * len(mods))()
            for i, b in enumerate(mods):
                arr[i] = _lib.BlockWeights(dev(b.norm1.weight), dev(b.norm1.bias), dev(b.attn.qkv.weight),
                                           dev(b.attn.qkv.bias), dev(b.attn.proj.weight), dev(b.attn.proj.bias),
                                           dev(b.norm2.weight), dev(b.norm2.bias), dev(b.mlp.fc1.weight),
                                           dev(b.mlp.fc1.bias), dev(b.mlp.fc2.weight), dev(b.mlp.fc2.bias))
            return arr

        half = self.embed_dim // 2
        freq = torch.exp(torch.arange(half) * -(math.log(10000) / (half - 1)))    # mixste.py:134-136, fp32 on host
        freq_keep = freq = freq.to(device=device, dtype=torch.float32).contiguous()
        ste, tte = blocks(self.STEblocks), blocks(self.TTEblocks)
        w = _lib.Weights(dev(self.Spatial_pos_embed), dev(self.Temporal_pos_embed),
                         dev(self.Spatial_patch_to_embedding.weight), dev(self.Spatial_patch_to_embedding.bias),
                         dev(freq), dev(self.time_mlp[1].weight), dev(self.time_mlp[1].bias),
                         dev(self.time_mlp[3].weight), dev(self.time_mlp[3].bias),
                         dev(self.Spatial_norm.weight), dev(self.Spatial_norm.bias),
                         dev(self.Temporal_norm.weight), dev(self.Temporal_norm.bias),
                         dev(self.head[0].weight), dev(self.head[0].bias), dev(self.head[1].weight),
                         dev(self.head[1].bias), ste, tte)
        with torch.cuda.device(device):
            if borrowed:
                _lib.check(lib.d3dp_set_weights_borrowed(st.ctx, C.byref(w)), "d3dp_set_weights_borrowed")
            else:
                _lib.check(lib.d3dp_set_weights(st.ctx, C.byref(w), _lib.current_stream()), "d3dp_set_weights")
        st.keep = [freq_keep] if borrowed else None   # packed copies: originals may go; borrowed: keep the table

    def _workspace(self, ctx, B, H, device):
        n = C.c_size_t()
        _lib.check(_lib.load().d3dp_workspace_bytes(ctx, B, H, C.byref(n)), "d3dp_workspace_bytes")
        st = self._state()
        if st.ws is None or st.ws.numel() < n.value:
            st.ws = torch.empty(n.value, dtype=torch.uint8, device=self._last_device)
        return st.ws, n.value

    def denoise(self, x_2d: torch.Tensor, x_3d: torch.Tensor, t: torch.Tensor, out: Optional[torch.Tensor] = None):
        """x_2d (B,F,J,2), x_3d (B,H,F,J,3), t (B,) int64 -> (B,H,F,J,3) fp32 (all on the GPU)."""
        B, H, Fr, J, _ = x_3d.shape
        assert x_2d.shape == (B, Fr, J, 2), (x_2d.shape, x_3d.shape)
        assert Fr == self.num_frame and J == self.num_joints, "clip shape differs from the model's (F, J)"
        assert t.shape == (B,)
        dev = x_3d.device
        ctx = self._context(dev)
        x_2d = x_2d.to(dtype=torch.float32).contiguous()
        x_3d = x_3d.to(dtype=torch.float32).contiguous()
        t = t.to(device=dev, dtype=torch.int64).contiguous()
        if out is None:
            out = torch.empty((B, H, Fr, J, 3), dtype=torch.float32, device=dev)
        ws, nbytes = self._workspace(ctx, B, H, dev)
        with torch.cuda.device(dev):
            _lib.check(_lib.load().d3dp_denoise(ctx, x_2d.data_ptr(), x_3d.data_ptr(), t.data_ptr(), out.data_ptr(),
                                                B, H, ws.data_ptr(), nbytes, _lib.current_stream()), "d3dp_denoise")
        return out

    def forward(self, x_2d, x_3d, t, droppath=None):
        """Reference signature (mixste.py:278): eval -> x_3d (B,H,F,J,3); train branch -> x_3d (B,F,J,3).
        In 'train' numerics with grad enabled the train branch is differentiable w.r.t. every parameter
        (d3dp_train_forward / d3dp_train_backward); DropPath is active in ``.train()`` mode like the reference
        (mixste.py:100,187) and ``droppath={'STEblocks.i': (m_attn, m_mlp), ...}`` injects recorded masks."""
        if x_3d.dim() == 4:
            if self._mode != _lib.MODE_TRAIN and self.training and torch.is_grad_enabled():
                raise RuntimeError("MixSTE2 train branch called with autograd enabled on a model in %r numerics: only "
                                   "numerics='train' is differentiable (build the model with is_train=True, or wrap "
                                   "inference in torch.no_grad() / call .eval())" % self.numerics)
            if self._mode == _lib.MODE_TRAIN and torch.is_grad_enabled():
                masks = self._droppath_masks(x_3d.shape[0], x_3d.device, droppath)
                return _TrainStep.apply(self, x_2d, x_3d, t, masks, *self._weight_tensors())
            return self.denoise(x_2d, x_3d[:, None], t)[:, 0]
        return self.denoise(x_2d, x_3d, t)

    # -- training step ----------------------------------------------------------------------------
    def _droppath_masks(self, B, device, injected=None):
        """(2*depth, 2, B*max(F,J)) fp32 scales (0 or 1/keep) or None.  timm DropPath: per-sample Bernoulli(keep),
        scaled by 1/keep; block i has rate linspace(0, drop_path_rate, depth)[i] (mixste.py:187); a rate of 0 builds
        nn.Identity (mixste.py:100)."""
        Fr, J, dep = self.num_frame, self.num_joints, self.block_depth
        smax = B * max(Fr, J)
        if injected is not None:
            m = torch.ones((2 * dep, 2, smax), dtype=torch.float32)
            for name, pair in injected.items():
                kind, i = name.split(".")
                blk = 2 * int(i) + (1 if kind == "TTEblocks" else 0)
                for br in (0, 1):
                    v = pair[br].reshape(-1).float()
                    m[blk, br, :v.numel()] = v
            return m.to(device).contiguous()
        if not self.training or not self.drop_path_rate:
            return None
        # every block's masks in three launches (one uniform draw over the whole table, a compare and a divide) instead of a
        # bernoulli + divide + copy per (block, branch): ~100 tiny launches per training step
        key = (device, float(self.drop_path_rate), dep)
        cached = self.__dict__.get("_droppath_keep")
        if cached is None or cached[0] != key:        # (keyed on the rate too: changing drop_path_rate takes effect)
            rates = torch.linspace(0, self.drop_path_rate, dep)
            keep = (1.0 - rates).repeat_interleave(2).reshape(2 * dep, 1, 1).to(device=device, dtype=torch.float32)
            # (rate 0 -> keep 1: rand() < 1 always, scale 1 -- an Identity, mixste.py:100; rate 1 -> keep 0: everything dropped and,
            #  like timm's `if keep_prob > 0` guard, no division by the zero keep rate)
            cached = (key, keep, torch.where(keep > 0, keep, torch.ones_like(keep)))
            self.__dict__["_droppath_keep"] = cached
        _, keep, div = cached
        return ((torch.rand((2 * dep, 2, smax), device=device) < keep).to(torch.float32) / div).contiguous()

    def _train_io(self, x_2d, x_3d, t):
        B, Fr, J, _ = x_3d.shape
        assert x_2d.shape == (B, Fr, J, 2) and Fr == self.num_frame and J == self.num_joints and t.shape == (B,)
        dev = x_3d.device
        ctx = self._context(dev)
        n = C.c_size_t()
        _lib.check(_lib.load().d3dp_train_workspace_bytes(ctx, B, C.byref(n)), "d3dp_train_workspace_bytes")
        st = self._state()
        if st.train_ws is None or st.train_ws.numel() < n.value:
            st.train_ws = torch.empty(n.value, dtype=torch.uint8, device=self._last_device)
        return (ctx, B, dev, n.value, x_2d.float().contiguous(), x_3d.float().contiguous(),
                t.to(device=dev, dtype=torch.int64).contiguous())

    def _train_forward(self, x_2d, x_3d, t, masks):
        ctx, B, dev, nbytes, x2, x3, tt = self._train_io(x_2d, x_3d, t)
        out = torch.empty_like(x3)
        st = self._state()
        st.train_gen += 1
        with torch.cuda.device(dev):
            _lib.check(_lib.load().d3dp_train_forward(ctx, x2.data_ptr(), x3.data_ptr(), tt.data_ptr(), _lib.ptr(masks),
                                                      out.data_ptr(), B, st.train_ws.data_ptr(), nbytes,
                                                      _lib.current_stream()), "d3dp_train_forward")
        return out

    def _train_backward(self, x_2d, x_3d, t, masks, grad_out):
        ctx, B, dev, nbytes, x2, x3, tt = self._train_io(x_2d, x_3d, t)
        params = self._weight_tensors()
        g = {id(p): torch.empty_like(p, dtype=torch.float32) for p in params}

        def gp(p):
            return g[id(p)].data_ptr()

        def blocks(mods):
            arr = (_lib.BlockWeights * len(mods))()
            for i, b in enumerate(mods):
                arr[i] = _lib.BlockWeights(gp(b.norm1.weight), gp(b.norm1.bias), gp(b.attn.qkv.weight), gp(b.attn.qkv.bias),
                                           gp(b.attn.proj.weight), gp(b.attn.proj.bias), gp(b.norm2.weight), gp(b.norm2.bias),
                                           gp(b.mlp.fc1.weight), gp(b.mlp.fc1.bias), gp(b.mlp.fc2.weight), gp(b.mlp.fc2.bias))
            return arr

        ste, tte = blocks(self.STEblocks), blocks(self.TTEblocks)
        dummy = torch.empty(self.embed_dim, dtype=torch.float32, device=dev)      # time_freq has no gradient
        w = _lib.Weights(gp(self.Spatial_pos_embed), gp(self.Temporal_pos_embed), gp(self.Spatial_patch_to_embedding.weight),
                         gp(self.Spatial_patch_to_embedding.bias), dummy.data_ptr(), gp(self.time_mlp[1].weight),
                         gp(self.time_mlp[1].bias), gp(self.time_mlp[3].weight), gp(self.time_mlp[3].bias),
                         gp(self.Spatial_norm.weight), gp(self.Spatial_norm.bias), gp(self.Temporal_norm.weight),
                         gp(self.Temporal_norm.bias), gp(self.head[0].weight), gp(self.head[0].bias), gp(self.head[1].weight),
                         gp(self.head[1].bias), ste, tte)
        with torch.cuda.device(dev):
            _lib.check(_lib.load().d3dp_train_backward(ctx, x2.data_ptr(), x3.data_ptr(), tt.data_ptr(), _lib.ptr(masks),
                                                       grad_out.float().contiguous().data_ptr(), C.byref(w), B,
                                                       self._state().train_ws.data_ptr(), nbytes, _lib.current_stream()),
                       "d3dp_train_backward")
        return [g[id(p)] for p in params]

    def train_arithmetic(self) -> str:
        """What the training step's Linears run on (bench.py reports it next to the step time)."""
        ta = os.environ.get("D3DP_TRAIN_ATTN", "")
        x2_ok = self.embed_dim // self.num_heads == 64 and self.num_frame <= 1024
        if ta == "f32" or not x2_ok:
            attn = "fp32 attention (fp32 MFMA: temporal forward, backward of both axes; spatial forward on the VALU)"
        elif ta == "x2t":
            attn = ("temporal attention forward and backward on split-fp16 operands (fp16 MFMA, running power-of-two scale for dS); "
                    "spatial attention forward on the VALU, backward on the fp32 matrix cores (D3DP_TRAIN_ATTN=x2t)")
        else:
            attn = ("attention of both axes, forward and backward, on split-fp16 operands (fp16 MFMA, base-2 online softmax, running "
                    "power-of-two scale for dS)")
        if os.environ.get("D3DP_TRAIN_ATTN_BWD", "")[:1] == "v":
            attn += "; D3DP_TRAIN_ATTN_BWD=valu: every fp32 attention backward on the VALU kernels instead"
        hd = self.embed_dim // self.num_heads
        inst = self.embed_dim in (64, 128, 256, 512) and hd in (8, 16, 32, 64) and self.hidden >= 64 and self.hidden % 64 == 0
        if not inst:       # (include/d3dp_hip.h d3dp_create: a width outside the instantiated set trains on the fp32 path)
            return (f"fp32 path of a width outside {{64, 128, 256, 512}} (cs = {self.embed_dim}): fp32-MFMA Linears (weight gradients split-K, "
                    "chunks added in a fixed order), fp32 row attention forward, VALU attention backward with a run-time head dim, "
                    "run-time-width row kernels")
        if os.environ.get("D3DP_TRAIN_IMPL") == "f32":
            return "fp32 MFMA Linears (D3DP_TRAIN_IMPL=f32), fp32 attention"
        wg = ("a launch per weight gradient (D3DP_TRAIN_WGRAD=each)" if os.environ.get("D3DP_TRAIN_WGRAD") == "each"
              else "the four weight gradients of a block as one TN launch")
        return ("split-fp16 Linears (three fp16-MFMA passes, fp32 accumulate, device-side operand scales; forward and dgrad on "
                "256 x 128 tiles with the batch's last T mod 256 rows as 16 x 64 blocks, " + wg + ", partial tiles summed in a "
                "fixed order; LayerNorm and attention outputs written by their producers as the next Linear's operand rows), " + attn)

    # -- profiling passthrough --------------------------------------------------------------------
    def profile_enable(self, on: bool = True):
        _lib.check(_lib.load().d3dp_profile_enable(self._state().ctx, int(on)))

    def profile_read(self):
        lib = _lib.load()
        cnt = (C.c_int64 * _lib.PROFILE_CLASSES)()
        ms = (C.c_double * _lib.PROFILE_CLASSES)()
        _lib.check(lib.d3dp_profile_read(self._state().ctx, cnt, ms))
        return {lib.d3dp_profile_class_name(i).decode(): (int(cnt[i]), float(ms[i])) for i in range(_lib.PROFILE_CLASSES)}


# ----------------------------------------------------------------------------------------------
# diffusion wrapper
# ----------------------------------------------------------------------------------------------
class D3DP(nn.Module):
    """Drop-in for the reference's ``D3DP`` (common/diffusionpose.py:55-126)."""

    def __init__(self, args, joints_left, joints_right, is_train=True, num_proposals=1, sampling_timesteps=1,
                 numerics: Optional[str] = None):
        super().__init__()
        self.frames = args.number_of_frames
        self.num_proposals = num_proposals
        self.flip = args.test_time_augmentation
        self.joints_left = list(joints_left)
        self.joints_right = list(joints_right)
        self.is_train = is_train

        self.objective = 'pred_x0'                       # diffusionpose.py:74 (attribute parity; the path never branches on it)
        betas = cosine_beta_schedule(args.timestep)
        alphas = 1. - betas
        alphas_cumprod = torch.cumprod(alphas, dim=0)
        alphas_cumprod_prev = F.pad(alphas_cumprod[:-1], (1, 0), value=1.)
        timesteps, = betas.shape
        self.num_timesteps = int(timesteps)
        self.sampling_timesteps = sampling_timesteps if sampling_timesteps is not None else timesteps
        assert self.sampling_timesteps <= timesteps
        self.is_ddim_sampling = self.sampling_timesteps < timesteps
        self.ddim_sampling_eta = 1.
        self.self_condition = False                      # diffusionpose.py:87-90
        self.scale = args.scale
        self.box_renewal = True
        self.use_ensemble = True

        # the 12 fp64 buffers of the reference state_dict (diffusionpose.py:92-117)
        self.register_buffer('betas', betas)
        self.register_buffer('alphas_cumprod', alphas_cumprod)
        self.register_buffer('alphas_cumprod_prev', alphas_cumprod_prev)
        self.register_buffer('sqrt_alphas_cumprod', torch.sqrt(alphas_cumprod))
        self.register_buffer('sqrt_one_minus_alphas_cumprod', torch.sqrt(1. - alphas_cumprod))
        self.register_buffer('log_one_minus_alphas_cumprod', torch.log(1. - alphas_cumprod))
        self.register_buffer('sqrt_recip_alphas_cumprod', torch.sqrt(1. / alphas_cumprod))
        self.register_buffer('sqrt_recipm1_alphas_cumprod', torch.sqrt(1. / alphas_cumprod - 1))
        posterior_variance = betas * (1. - alphas_cumprod_prev) / (1. - alphas_cumprod)
        self.register_buffer('posterior_variance', posterior_variance)
        self.register_buffer('posterior_log_variance_clipped', torch.log(posterior_variance.clamp(min=1e-20)))
        self.register_buffer('posterior_mean_coef1', betas * torch.sqrt(alphas_cumprod_prev) / (1. - alphas_cumprod))
        self.register_buffer('posterior_mean_coef2',
                             (1. - alphas_cumprod_prev) * torch.sqrt(alphas) / (1. - alphas_cumprod))

        # the train model is always differentiable ('train' numerics); args.numerics / D3DP_NUMERICS select the
        # inference arithmetic of the evaluation models only (one argparse namespace builds all three, main.py:228-230)
        numerics = "train" if is_train else (numerics or getattr(args, "numerics", None))
        self.pose_estimator = MixSTE2(num_frame=self.frames, num_joints=NUM_JOINTS, in_chans=2,
                                      embed_dim_ratio=args.cs, depth=args.dep, num_heads=8, mlp_ratio=2.,
                                      qkv_bias=True, qk_scale=None, drop_path_rate=0.1 if is_train else 0,
                                      is_train=is_train, numerics=numerics,
                                      chunk_seqs=int(getattr(args, "chunk_seqs", 0) or 0))
        self._perm_cache = {}
        self._sched_cache = None

    # ---- helpers ----------------------------------------------------------------------------------
    def _perm(self, device):
        """perm[j] = source joint of joint j under the left/right swap (diffusionpose.py:152-153)."""
        if device not in self._perm_cache:
            perm = list(range(NUM_JOINTS))
            for dst, src in zip(self.joints_left + self.joints_right, self.joints_right + self.joints_left):
                perm[dst] = src
            self._perm_cache[device] = torch.tensor(perm, dtype=torch.int32, device=device)
        return self._perm_cache[device]

    def _sched(self):
        """Host fp64 copies of the buffers the sampler reads as scalars."""
        if self._sched_cache is None:
            self._sched_cache = {k: getattr(self, k).detach().cpu().numpy().astype("float64")
                                 for k in ("alphas_cumprod", "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod")}
        return self._sched_cache

    def _load_from_state_dict(self, *a, **k):
        self._sched_cache = None
        return super()._load_from_state_dict(*a, **k)

    def time_pairs(self):
        times = torch.linspace(-1, self.num_timesteps - 1, steps=self.sampling_timesteps + 1)
        times = list(reversed(times.int().tolist()))
        return list(zip(times[:-1], times[1:]))

    def _draw(self, shape, device, noise, idx, generator):
        if noise is not None:
            n = noise[idx].to(device=device, dtype=torch.float32).contiguous()
            assert tuple(n.shape) == tuple(shape), (n.shape, shape)
            return n
        return torch.randn(shape, device=device, generator=generator)

    # ---- samplers ---------------------------------------------------------------------------------
    @torch.no_grad()
    def ddim_sample_flip(self, inputs_2d, inputs_3d, clip_denoised=True, do_postprocess=True, input_2d_flip=None,
                         noise: Optional[Sequence[torch.Tensor]] = None, generator=None):
        """reference diffusionpose.py:214-256.  Returns (B, K, H, F, 17, 3) fp32."""
        lib = _lib.load()
        dev = inputs_2d.device
        if dev.type != "cuda":
            raise _lib.D3DPHipError("D3DP samples only on an MI355X (inputs on %s); there is no CPU fallback" % dev)
        B, H, Fr, J, K = inputs_2d.shape[0], self.num_proposals, self.frames, NUM_JOINTS, self.sampling_timesteps
        shape = (B, H, Fr, J, 3)
        sch = self._sched()
        perm = self._perm(dev)
        x2 = torch.cat((inputs_2d, input_2d_flip), dim=0).to(dtype=torch.float32).contiguous()
        img = self._draw(shape, dev, noise, 0, generator).clone()
        xt2 = torch.empty((2 * B,) + shape[1:], dtype=torch.float32, device=dev)
        pred2 = torch.empty_like(xt2)
        preds = torch.empty((B, K, H, Fr, J, 3), dtype=torch.float32, device=dev)
        per_b = H * Fr * J * 3
        scale = float(self.scale)
        eta = self.ddim_sampling_eta
        ac = sch["alphas_cumprod"]
        with torch.cuda.device(dev):
            for k, (time, time_next) in enumerate(self.time_pairs()):
                st = _lib.current_stream()
                _lib.check(lib.d3dp_ddim_pre(img.data_ptr(), xt2.data_ptr(), perm.data_ptr(), scale, B, H, Fr, J, st),
                           "d3dp_ddim_pre")
                t2 = torch.full((2 * B,), time, dtype=torch.long, device=dev)
                self.pose_estimator.denoise(x2, xt2, t2, out=pred2)
                last = time_next < 0
                if last:
                    c_x, c_n, sigma, nz = 0.0, 0.0, 0.0, None
                else:
                    alpha, alpha_next = ac[time], ac[time_next]
                    sg = eta * math.sqrt((1 - alpha / alpha_next) * (1 - alpha_next) / (1 - alpha))
                    c_n = math.sqrt(1 - alpha_next - sg ** 2)
                    c_x, sigma = math.sqrt(alpha_next), sg
                    nz = self._draw(shape, dev, noise, k + 1, generator)
                xs = preds[:, k]
                _lib.check(lib.d3dp_ddim_post(pred2.data_ptr(), img.data_ptr(), _lib.ptr(nz), perm.data_ptr(), scale,
                                              float(sch["sqrt_recip_alphas_cumprod"][time]),
                                              float(sch["sqrt_recipm1_alphas_cumprod"][time]),
                                              c_x, c_n, sigma, int(last), xs.data_ptr(), K * per_b, img.data_ptr(),
                                              B, H, Fr, J, st), "d3dp_ddim_post")
        return preds

    @torch.no_grad()
    def ddim_sample(self, inputs_2d, inputs_3d, clip_denoised=True, do_postprocess=True,
                    noise: Optional[Sequence[torch.Tensor]] = None, generator=None):
        """reference diffusionpose.py:171-212 (dead there: it reads an undefined ``self.device``).  Made to work
        and to return the list of K x_start tensors it was written to return.  No flip augmentation; the
        elementwise glue is a handful of torch ops on the GPU around the HIP denoiser."""
        dev = inputs_2d.device
        B, H = inputs_2d.shape[0], self.num_proposals
        shape = (B, H, self.frames, NUM_JOINTS, 3)
        img = self._draw(shape, dev, noise, 0, generator).clone()
        scale, preds_all = self.scale, []
        for k, (time, time_next) in enumerate(self.time_pairs()):
            t = torch.full((B,), time, device=dev, dtype=torch.long)
            x_t = torch.clamp(img, min=-1.1 * scale, max=1.1 * scale) / scale
            x_start = torch.clamp(self.pose_estimator.denoise(inputs_2d, x_t, t) * scale, min=-1.1 * scale, max=1.1 * scale)
            pred_noise = self.predict_noise_from_start(img, t, x_start)
            preds_all.append(x_start)
            if time_next < 0:
                img = x_start
                continue
            alpha, alpha_next = self.alphas_cumprod[time], self.alphas_cumprod[time_next]
            sigma = self.ddim_sampling_eta * ((1 - alpha / alpha_next) * (1 - alpha_next) / (1 - alpha)).sqrt()
            c = (1 - alpha_next - sigma ** 2).sqrt()
            nz = self._draw(shape, dev, noise, k + 1, generator)
            img = x_start * alpha_next.sqrt() + c * pred_noise + sigma * nz
        return preds_all

    def predict_noise_from_start(self, x_t, t, x0):
        """reference diffusionpose.py:129-133 (fp64 by buffer promotion)."""
        shp = (t.shape[0],) + (1,) * (x_t.dim() - 1)
        a = self.sqrt_recip_alphas_cumprod.gather(-1, t).reshape(shp)
        b = self.sqrt_recipm1_alphas_cumprod.gather(-1, t).reshape(shp)
        return (a * x_t - x0) / b

    def q_sample(self, x_start, t, noise=None):
        """reference diffusionpose.py:260-267 (returns fp64 like the reference)."""
        if noise is None:
            noise = torch.randn_like(x_start)
        shp = (t.shape[0],) + (1,) * (x_start.dim() - 1)
        a = self.sqrt_alphas_cumprod.gather(-1, t).reshape(shp)
        b = self.sqrt_one_minus_alphas_cumprod.gather(-1, t).reshape(shp)
        return a * x_start + b * noise

    def prepare_targets(self, targets, t: Optional[torch.Tensor] = None, noise: Optional[torch.Tensor] = None):
        """reference diffusionpose.py:290-320 as ONE fused kernel over the batch (the reference loops over samples
        in Python).  ``t`` (B,1) int64 and ``noise`` (B,F,17,3) may be injected; otherwise drawn like the reference
        (per sample: randint then randn)."""
        lib = _lib.load()
        dev = targets.device
        B = targets.shape[0]
        if t is None or noise is None:
            ts, ns = [], []
            for _ in range(B):
                ts.append(torch.randint(0, self.num_timesteps, (1,), device=dev).long())
                ns.append(torch.randn(self.frames, NUM_JOINTS, 3, device=dev))
            t = torch.stack(ts) if t is None else t
            noise = torch.stack(ns) if noise is None else noise
        t = t.to(dev).long().reshape(B, 1)
        noise = noise.to(device=dev, dtype=torch.float32).contiguous()
        x0 = targets.to(dtype=torch.float32).contiguous()
        a = self.sqrt_alphas_cumprod.to(dev).gather(-1, t[:, 0]).contiguous()
        s = self.sqrt_one_minus_alphas_cumprod.to(dev).gather(-1, t[:, 0]).contiguous()
        out = torch.empty_like(x0)
        with torch.cuda.device(dev):
            _lib.check(lib.d3dp_q_sample(x0.data_ptr(), noise.data_ptr(), a.data_ptr(), s.data_ptr(), float(self.scale),
                                         out.data_ptr(), B, x0[0].numel(), _lib.current_stream()), "d3dp_q_sample")
        return out, noise, t

    def forward(self, input_2d, input_3d, input_2d_flip=None, **kw):
        """reference diffusionpose.py:269-287."""
        if not self.is_train:
            out = (self.ddim_sample_flip(input_2d, input_3d, input_2d_flip=input_2d_flip, **kw) if self.flip
                   else self.ddim_sample(input_2d, input_3d, **kw))
            if os.environ.get("D3DP_CHECK_FINITE") == "1" and self.pose_estimator.nonfinite_seen():   # (synchronises)
                raise _lib.D3DPHipError("non-finite denoiser output: the input or the weights hold inf / nan, or the fp32 "
                                        "arithmetic itself overflowed (EXACT mode's split-fp16 operands cannot: their scales "
                                        "follow the range the weights prove, MixSTE2.exact_scales())")
            return out
        droppath = kw.pop("droppath", None)
        x_poses, _, t = self.prepare_targets(input_3d, **kw)
        return self.pose_estimator(input_2d, x_poses.float(), t.squeeze(-1), droppath=droppath)


class D3DP3DHP(D3DP):
    """Drop-in for the MPI-INF-3DHP variant (reference common/diffusionpose_3dhp.py): the same sampler and denoiser
    working in metres internally, with millimetre poses at the boundary -- sampler outputs are multiplied by 1000
    (:212, :256), training targets arrive in millimetres and are divided by 1000 (:280) and the training prediction
    is returned in millimetres (:287)."""

    MM = 1000.0

    def ddim_sample_flip(self, *a, **k):
        return super().ddim_sample_flip(*a, **k).mul_(self.MM)

    def ddim_sample(self, *a, **k):
        return [p * self.MM for p in super().ddim_sample(*a, **k)]

    def forward(self, input_2d, input_3d, input_2d_flip=None, **kw):
        if not self.is_train:
            return super().forward(input_2d, input_3d, input_2d_flip=input_2d_flip, **kw)
        return super().forward(input_2d, input_3d / self.MM, **kw) * self.MM
