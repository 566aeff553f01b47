import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


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
