"""Synthetic voxel generators: inputs for bench.py, the profiling tools and the tests.
No reference arithmetic lives here (the CPU oracle is ``oracle/``).

``forest_plot`` generates a whole plot for the voxeliser; ``mixed_sizes`` the voxel sizes of BASELINE configs[4].
``uniform_voxel`` is the canonical ``U(side, N, seed)``: N points uniform in a cube,
centred, with the feed-side quantities of ``TestingDataset.__getitem__``
(``pointstowood/src/predicter.py:78-94``): ``local_shift`` (= mean, here of the raw
cube), centred ``pos`` and ``sf = max ||pos||``.
"""
from __future__ import annotations

import math

import torch


def uniform_points(side: float, n: int, seed: int, reflectance: bool = False, offset=(0.0, 0.0, 0.0)):
    """The raw points of ``uniform_voxel`` (before centring): seeded CPU generator, element-wise arithmetic - the same bits on
    every host.  (The centring is not: ``mean`` and ``sqrt`` round differently on different CPUs' vector units, so fixtures at
    sizes too large to store keep ``local_shift`` / ``sf`` and rebuild ``pos`` = these points - the stored shift.)"""
    g = torch.Generator().manual_seed(seed)
    p = torch.rand(n, 3, generator=g) * side + torch.tensor(offset, dtype=torch.float32)
    refl = (torch.rand(n, generator=g) * 2 - 1) if reflectance else torch.zeros(n)
    return p, refl


def uniform_voxel(side: float, n: int, seed: int, reflectance: bool = False, offset=(0.0, 0.0, 0.0)):
    return _finish(*uniform_points(side, n, seed, reflectance, offset))


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


def forest_plot(n, seed=0, side=100.0, height=30.0):
    """Synthetic plot (BASELINE configs[3] shape): [n, 4] = x, y, z, reflectance with a tree-like density - stems (thin
    vertical cylinders), crowns (gaussian blobs) and a ground sheet over a side x side m square, plot-local coordinates."""
    g = torch.Generator().manual_seed(seed)
    n_tree = max(4, int(side * side / 60))
    cx = torch.rand(n_tree, 2, generator=g) * side - side / 2
    h = 8 + torch.rand(n_tree, generator=g) * (height - 10)
    which = torch.randint(0, n_tree, (n,), generator=g)
    kind = torch.rand(n, generator=g)
    stem, crown = kind < 0.25, (kind >= 0.25) & (kind < 0.85)
    p = torch.empty(n, 3)
    ang = torch.rand(n, generator=g) * 6.2832
    rad = 0.1 + 0.15 * torch.rand(n, generator=g)
    p[:, 0] = cx[which, 0] + torch.where(stem, rad * torch.cos(ang), torch.randn(n, generator=g) * 1.6)
    p[:, 1] = cx[which, 1] + torch.where(stem, rad * torch.sin(ang), torch.randn(n, generator=g) * 1.6)
    p[:, 2] = torch.where(stem, torch.rand(n, generator=g) * h[which] * 0.7,
                          h[which] * (0.65 + 0.12 * torch.randn(n, generator=g)))
    gr = ~(stem | crown)
    p[gr, 0] = torch.rand(int(gr.sum()), generator=g) * side - side / 2
    p[gr, 1] = torch.rand(int(gr.sum()), generator=g) * side - side / 2
    p[gr, 2] = 0.05 * torch.randn(int(gr.sum()), generator=g)
    refl = torch.rand(n, 1, generator=g) * 30 - 25
    return torch.cat([p, refl], 1)


def mixed_sizes(count: int = 128, lo: int = 512, hi: int = 16384, seed: int = 7):
    """Voxel sizes of BASELINE configs[4]: ``count`` sizes log-uniform in [lo, hi] (seeded)."""
    g = torch.Generator().manual_seed(seed)
    return [int(round(math.exp(float(torch.rand(1, generator=g)) * math.log(hi / lo) + math.log(lo)))) for _ in range(count)]
