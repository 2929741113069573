"""pointstowood_amd - MI355X (gfx950) inference forward for PointsToWood.

Python host code on PyTorch-ROCm (device memory, streams, torch.distributed) over
hand-written HIP kernels behind a C ABI (``include/p2w.h``, ``libp2w_gfx950.so``).
There is no CPU fallback: every operator raises if the library is missing.
"""
__version__ = "0.1.0"

from .model import Net, checkpoint_layout  # noqa: E402,F401
from .data import Batch, Data, DataLoader  # noqa: E402,F401
