"""Shim for pointstowood/src/model.py:7."""
from oracle.ops import consecutive_cluster  # noqa: F401
