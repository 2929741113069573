"""Shim for pointstowood/src/pointnet.py:17 (only scatter_max is called, :122)."""
from oracle.ops import scatter_max  # noqa: F401


def scatter_mean(*a, **k):
    raise NotImplementedError


def scatter_std(*a, **k):
    raise NotImplementedError


def scatter_min(src, index, dim=0, out=None, dim_size=None):
    """torch-scatter ``scatter_min`` as pointstowood/src/preprocessing.py:49 calls it (1-D src, every segment
    non-empty): per-segment minimum; the argmin is returned for signature compatibility (unused there)."""
    import torch
    assert dim == 0 and src.dim() == 1
    size = int(index.max()) + 1 if dim_size is None else dim_size
    res = torch.full((size,), float("inf"), dtype=src.dtype).scatter_reduce(0, index, src, reduce="amin", include_self=True)
    return res, torch.full((size,), -1, dtype=torch.long)
