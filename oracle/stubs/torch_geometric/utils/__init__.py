"""Shim for pointstowood/src/pointnet.py:16 (unused on this path: add_self_loops=False at model.py:93)."""


def add_self_loops(*a, **k):
    raise NotImplementedError


def remove_self_loops(*a, **k):
    raise NotImplementedError
