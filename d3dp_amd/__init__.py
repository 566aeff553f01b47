"""d3dp_amd -- MI355X (gfx950) implementation of the D3DP hot path: the DDIM multi-hypothesis sampler
(reference common/diffusionpose.py) and the MixSTE2 denoiser it calls (reference common/mixste.py), behind
the reference's own Python API.  All arithmetic runs in libd3dp_hip.so (include/d3dp_hip.h)."""
import os as _os

# dmabuf IPC for multi-process device memory (RCCL, nn.DataParallel's peers): ROCr reads the flag at the process's first HIP
# call, so it is exported at import, before anything can initialise the runtime (d3dp_amd/dist.py).
_os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

from .model import D3DP, D3DP3DHP, MixSTE2, cosine_beta_schedule  # noqa: F401

__version__ = "0.1.0"
