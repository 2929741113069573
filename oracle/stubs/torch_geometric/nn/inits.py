"""Shim for ``reset`` (pointstowood/src/pointnet.py:7)."""


def reset(value):
    if value is None:
        return
    if hasattr(value, "reset_parameters"):
        value.reset_parameters()
    else:
        for child in value.children() if hasattr(value, "children") else []:
            reset(child)
