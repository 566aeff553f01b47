import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "variants: exercises the measured-negative experiment kernels of gemm_x2.hip, which the product "
                                       "library does not carry: runs only against a `make -C d3dp_amd/csrc variants` build "
                                       "(D3DP_LIB=d3dp_amd/lib/variants/libd3dp_variants.so), skipped otherwise")


def pytest_collection_modifyitems(config, items):
    """gpu-marked tests need an MI355X and the built library: skip (not fail) them anywhere else."""
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    has_lib = os.path.exists(os.path.join(REPO, "d3dp_amd", "lib", "libd3dp_hip.so"))
    if has_gpu and has_lib:
        return
    why = "needs an MI355X" if not has_gpu else "libd3dp_hip.so not built (python -c 'import __graft_entry__ as g; g.build()')"
    skip = pytest.mark.skip(reason=why)
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(autouse=True)
def _variants_need_their_library(request):
    """Tests marked `variants` run only when the loaded library was built with -DD3DP_X2_VARIANTS=1."""
    if "variants" in request.keywords:
        from d3dp_amd import _lib
        lib = _lib.load()
        if not (hasattr(lib, "d3dp_debug_x2_variants") and lib.d3dp_debug_x2_variants() == 1):
            pytest.skip("the product library carries no experiment kernels (make -C d3dp_amd/csrc variants; "
                        "D3DP_LIB=d3dp_amd/lib/variants/libd3dp_variants.so)")
