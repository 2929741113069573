"""Shim for ``pykdtree.kdtree.KDTree`` (imported at pointstowood/src/predicter.py:6, used at :133-134): exact k nearest
neighbours in float64 through scipy's cKDTree, the same ``query(x, k) -> (distances, indices)`` call.  pykdtree is not
installed here and not vendored by the reference; both return the exact k nearest, ascending distance (ties may
order differently).  See oracle/stubs/README.md."""
from scipy.spatial import cKDTree


class KDTree:
    def __init__(self, data, leafsize=16):
        self._t = cKDTree(data, leafsize=leafsize)

    def query(self, x, k=1, **kw):
        return self._t.query(x, k=k)
