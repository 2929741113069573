#!/usr/bin/env python3
"""Run the BASELINE.json workload shapes that are not the bench line (single GPU): configs[2] (B=64 x 16384, reflectance)
and configs[4] (B=128, voxel sizes log-uniform 512..16384), reporting time, points/s and peak memory."""
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointstowood_amd import synthetic_voxels as synth, synthetic_weights as weights  # noqa: E402
from pointstowood_amd import Net  # noqa: E402

dev = torch.device("cuda", 0)
net = Net(1, C=32, k=32)
net.load_state_dict(weights.synth_state_dict(1, 32, seed=0))
net = net.to(dev).eval()


class D:
    pass


def mk(vox):
    b = synth.collate(vox)
    d = D()
    d.pos, d.batch, d.reflectance, d.sf, d.ptr = (b[k].to(dev) for k in ("pos", "batch", "reflectance", "sf", "ptr"))
    return d


g = torch.Generator().manual_seed(7)
sizes = [int(round(math.exp(float(torch.rand(1, generator=g)) * math.log(16384 / 512) + math.log(512)))) for _ in range(128)]
cases = {
    "configs[2] B=64 x 16384 xyz+reflectance": [synth.uniform_voxel(2.0, 16384, 200 + i, True) for i in range(64)],
    "configs[4] B=128 mixed 512..16384": [synth.uniform_voxel(2.0, n, 400 + i, True) for i, n in enumerate(sizes)],
}
for name, vox in cases.items():
    d = mk(vox)
    n = d.pos.shape[0]
    torch.cuda.reset_peak_memory_stats()
    out = net(d)
    torch.cuda.synchronize()
    assert out.shape == (n,) and bool(torch.isfinite(out).all())
    reps = 3
    t0 = time.perf_counter()
    for o in net.stream(d for _ in range(reps)):
        pass
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"{name}: {n} points, {dt*1e3:.1f} ms/batch, {n/dt/1e6:.2f} M points/s, peak mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB",
          flush=True)
