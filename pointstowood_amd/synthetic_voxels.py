"""Synthetic voxel generators: inputs for bench.py, the profiling tools and the tests.
No reference arithmetic lives here (the CPU oracle is ``oracle/``; ``oracle.synth`` re-exports this module).

``uniform_voxel`` is the canonical ``U(side, N, seed)``: N points uniform in a cube,
centred, with the feed-side quantities of ``TestingDataset.__getitem__``
(``pointstowood/src/predicter.py:78-94``): ``local_shift`` (= mean, here of the raw
cube), centred ``pos`` and ``sf = max ||pos||``.
"""
from __future__ import annotations

import math

import torch


def uniform_voxel(side: float, n: int, seed: int, reflectance: bool = False, offset=(0.0, 0.0, 0.0)):
    g = torch.Generator().manual_seed(seed)
    p = torch.rand(n, 3, generator=g) * side + torch.tensor(offset, dtype=torch.float32)
    refl = (torch.rand(n, generator=g) * 2 - 1) if reflectance else torch.zeros(n)
    return _finish(p, refl)


def surface_voxel(side: float, n: int, seed: int, reflectance: bool = True):
    """Points on a few random cylinders ("stems/branches", r=0.03-0.3 m) plus gaussian
    leaf blobs: dense surfaces hit the ball-query cap and produce small level sizes."""
    g = torch.Generator().manual_seed(seed)
    n_cyl = 4
    per = n // (n_cyl + 2)
    parts = []
    for c in range(n_cyl):
        r = 0.03 + 0.27 * float(torch.rand(1, generator=g))
        base = torch.rand(3, generator=g) * side
        axis = torch.randn(3, generator=g)
        axis = axis / axis.norm()
        u = torch.linalg.cross(axis, torch.tensor([1.0, 0.0, 0.0]))
        if float(u.norm()) < 1e-3:
            u = torch.linalg.cross(axis, torch.tensor([0.0, 1.0, 0.0]))
        u = u / u.norm()
        v = torch.linalg.cross(axis, u)
        t = (torch.rand(per, generator=g) - 0.5) * side
        a = torch.rand(per, generator=g) * (2 * math.pi)
        pts = base[None] + t[:, None] * axis[None] + r * (torch.cos(a)[:, None] * u[None] + torch.sin(a)[:, None] * v[None])
        parts.append(pts + 0.002 * torch.randn(per, 3, generator=g))
    rest = n - per * n_cyl
    for b in range(2):
        m = rest // 2 if b == 0 else rest - rest // 2
        c = torch.rand(3, generator=g) * side
        parts.append(c[None] + 0.12 * torch.randn(m, 3, generator=g))
    p = torch.cat(parts, 0)
    p = p[torch.randperm(n, generator=g)]
    p = torch.minimum(torch.maximum(p, torch.zeros(3)), torch.full((3,), side))
    refl = (torch.rand(n, generator=g) * 2 - 1) if reflectance else torch.zeros(n)
    return _finish(p.contiguous(), refl)


def _finish(p, refl):
    shift = p.mean(dim=0)
    pos = p - shift
    sf = torch.sqrt((pos ** 2).sum(dim=1)).max()
    return {"pos": pos.contiguous(), "reflectance": refl.contiguous(), "local_shift": shift, "sf": sf}


def collate(voxels):
    """PyG ``Batch`` collation rules (predicter.py:177 via DataLoader): concat along dim 0,
    0-dim ``sf`` -> [B], ``local_shift`` [3] -> [3B], plus ``batch`` and ``ptr``."""
    n = [v["pos"].shape[0] for v in voxels]
    out = {
        "pos": torch.cat([v["pos"] for v in voxels], 0),
        "reflectance": torch.cat([v["reflectance"] for v in voxels], 0),
        "local_shift": torch.cat([v["local_shift"].reshape(-1) for v in voxels], 0),
        "sf": torch.stack([v["sf"].reshape(()) for v in voxels], 0),
        "batch": torch.repeat_interleave(torch.arange(len(voxels)), torch.tensor(n)),
        "ptr": torch.tensor([0] + list(torch.cumsum(torch.tensor(n), 0)), dtype=torch.long),
    }
    return out
