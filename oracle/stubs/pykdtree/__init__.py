"""Shim: see oracle/stubs/README.md."""
