"""d3dp_amd -- MI355X (gfx950) implementation of the D3DP hot path: the DDIM multi-hypothesis sampler
(reference common/diffusionpose.py) and the MixSTE2 denoiser it calls (reference common/mixste.py), behind
the reference's own Python API.  All arithmetic runs in libd3dp_hip.so (include/d3dp_hip.h)."""
from .model import D3DP, D3DP3DHP, MixSTE2, cosine_beta_schedule  # noqa: F401

__version__ = "0.1.0"
