import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # GPU tests must never silently pass on a box without a GPU.
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:  # pragma: no cover
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU in this container (selected only with -m gpu on the GPU box)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture
def tensor_backend():
    """Puts the oracle's tensor-operation restatement of the voxeliser's two sorting steps into ``preprocessing.backend`` for tests of
    the HOST-SIDE logic that run without a GPU (samplers, sharding over gloo).  The product's own backend is the HIP library and
    refuses host tensors."""
    from oracle import preprocess as OP
    from pointstowood_amd import preprocessing as PP
    old, PP.backend = PP.backend, OP.TensorBackend
    try:
        yield OP.TensorBackend
    finally:
        PP.backend = old
