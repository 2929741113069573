"""Shim for ``numba`` (pointstowood/src/predicter.py:14): ``jit`` becomes the identity decorator, so
``PointCloudClassifier.compute_labels`` (:112-127) runs as the plain numpy code it is written as; ``prange`` = ``range``.
See oracle/stubs/README.md."""


def jit(*args, **kwargs):
    if len(args) == 1 and callable(args[0]) and not kwargs:
        return args[0]
    return lambda f: f


njit = jit
prange = range


def set_num_threads(n):
    return None
