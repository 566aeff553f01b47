"""Deterministic synthetic weights for the MixSTE2 denoiser.

There is no network and no checkpoint in this environment, so parity fixtures,
tests and the benchmark all draw their weights from one platform-stable stream
(numpy PCG64).  Key names and shapes follow the reference ``state_dict`` exactly
(reference common/mixste.py:166-210 and
common/diffusionpose.py:123-124) so a real ``h36m_best_epoch.bin`` loads into the
same slots.

The reference zero-initialises both position embeddings (mixste.py:171,174); here
they are filled with N(0, 0.02) so that a pos-embed indexing bug cannot hide.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np

NUM_JOINTS = 17
IN_CHANS = 5  # (u, v) 2D input + (x, y, z) noisy 3D hypothesis, mixste.py:166


def block_param_shapes(cs: int) -> "OrderedDict[str, tuple]":
    hid = int(cs * 2.0)  # mlp_ratio = 2., diffusionpose.py:124
    return OrderedDict([
        ("norm1.weight", (cs,)), ("norm1.bias", (cs,)),
        ("attn.qkv.weight", (3 * cs, cs)), ("attn.qkv.bias", (3 * cs,)),
        ("attn.proj.weight", (cs, cs)), ("attn.proj.bias", (cs,)),
        ("norm2.weight", (cs,)), ("norm2.bias", (cs,)),
        ("mlp.fc1.weight", (hid, cs)), ("mlp.fc1.bias", (hid,)),
        ("mlp.fc2.weight", (cs, hid)), ("mlp.fc2.bias", (cs,)),
    ])


def param_shapes(cs: int, dep: int, frames: int, joints: int = NUM_JOINTS) -> "OrderedDict[str, tuple]":
    """Ordered {name: shape} of every MixSTE2 parameter (names relative to the
    ``pose_estimator.`` prefix).  ``joints``: MixSTE2's ``num_joints`` (mixste.py:141; D3DP itself builds 17)."""
    out: "OrderedDict[str, tuple]" = OrderedDict()
    out["Spatial_pos_embed"] = (1, joints, cs)
    out["Temporal_pos_embed"] = (1, frames, cs)
    out["Spatial_patch_to_embedding.weight"] = (cs, IN_CHANS)
    out["Spatial_patch_to_embedding.bias"] = (cs,)
    out["time_mlp.1.weight"] = (2 * cs, cs)
    out["time_mlp.1.bias"] = (2 * cs,)
    out["time_mlp.3.weight"] = (cs, 2 * cs)
    out["time_mlp.3.bias"] = (cs,)
    for kind in ("STEblocks", "TTEblocks"):
        for i in range(dep):
            for k, shp in block_param_shapes(cs).items():
                out[f"{kind}.{i}.{k}"] = shp
    out["Spatial_norm.weight"] = (cs,)
    out["Spatial_norm.bias"] = (cs,)
    out["Temporal_norm.weight"] = (cs,)
    out["Temporal_norm.bias"] = (cs,)
    out["head.0.weight"] = (cs,)
    out["head.0.bias"] = (cs,)
    out["head.1.weight"] = (3, cs)
    out["head.1.bias"] = (3,)
    return out


def _is_norm(name: str) -> bool:
    return (".norm1." in name or ".norm2." in name or name.startswith("Spatial_norm")
            or name.startswith("Temporal_norm") or name.startswith("head.0."))


def make_numpy_weights(seed: int, cs: int, dep: int, frames: int, joints: int = NUM_JOINTS) -> "OrderedDict[str, np.ndarray]":
    """fp32 numpy arrays keyed by the reference parameter names (no prefix)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    shapes = param_shapes(cs, dep, frames, joints)
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    fan_in_of_bias = {}
    for name, shp in shapes.items():
        if name.endswith("pos_embed"):
            a = rng.normal(0.0, 0.02, size=shp)
        elif _is_norm(name):
            base = 1.0 if name.endswith("weight") else 0.0
            a = base + rng.normal(0.0, 0.02, size=shp)
        elif name.endswith(".weight"):
            bound = 1.0 / math.sqrt(shp[1])
            fan_in_of_bias[name[:-len("weight")] + "bias"] = shp[1]
            a = rng.uniform(-bound, bound, size=shp)
        else:  # Linear bias: U(+-1/sqrt(fan_in)) like torch.nn.Linear's default
            bound = 1.0 / math.sqrt(fan_in_of_bias[name])
            a = rng.uniform(-bound, bound, size=shp)
        out[name] = np.ascontiguousarray(a.astype(np.float32))
    return out


def make_state_dict(seed: int, cs: int, dep: int, frames: int, prefix: str = "pose_estimator.", joints: int = NUM_JOINTS):
    """torch fp32 tensors keyed ``pose_estimator.<name>`` (loadable with
    ``D3DP.load_state_dict(..., strict=False)``; the 12 fp64 diffusion buffers are
    rebuilt by the constructor)."""
    import torch
    return OrderedDict((prefix + k, torch.from_numpy(v.copy()))
                       for k, v in make_numpy_weights(seed, cs, dep, frames, joints).items())


def synthetic_inputs_2d(seed: int, batch: int, frames: int):
    """``x2d ~ U(-1,1)`` of shape (B, F, 17, 2) fp32."""
    rng = np.random.Generator(np.random.PCG64(seed))
    return rng.uniform(-1.0, 1.0, size=(batch, frames, NUM_JOINTS, 2)).astype(np.float32)


def synthetic_noise(seed: int, shape):
    rng = np.random.Generator(np.random.PCG64(seed))
    return rng.standard_normal(size=shape).astype(np.float32)


# 17-joint Human3.6M skeleton left/right lists (reference common/h36m_dataset.py:14-17 after
# remove_joints; the same lists are hard-coded at in_the_wild/utils.py:251).
H36M_JOINTS_LEFT = [4, 5, 6, 11, 12, 13]
H36M_JOINTS_RIGHT = [1, 2, 3, 14, 15, 16]


def flip_2d(x2d, kps_left=H36M_JOINTS_LEFT, kps_right=H36M_JOINTS_RIGHT):
    """Flipped copy of the 2D input as the reference caller builds it (main.py:646-648)."""
    out = x2d.copy() if isinstance(x2d, np.ndarray) else x2d.clone()
    out[..., 0] *= -1
    out[..., kps_left + kps_right, :] = out[..., kps_right + kps_left, :]
    return out
