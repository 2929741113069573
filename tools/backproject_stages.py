#!/usr/bin/env python3
"""Wall time of the back-projection's stages on a synthetic plot (synchronising after every library call; diagnostic)."""
import os, sys, time, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointstowood_amd import synthetic_voxels as synth
from pointstowood_amd import backproject as bp

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
dev = torch.device("cuda", 0)
plot = synth.forest_plot(n, seed=1, side=(n / 1000.0) ** 0.5).to(dev)
xyz = plot[:, :3].contiguous()
cls = torch.cat([xyz, xyz], 0)[: int(1.9 * n)].contiguous()
pred = (torch.rand(cls.shape[0], device=dev) > 0.5).float()
prob = torch.rand(cls.shape[0], device=dev)
acc = collections.OrderedDict()
orig = bp.check
last = [0.0]


def check(status, name):
    torch.cuda.synchronize()
    now = time.perf_counter()
    acc[name] = acc.get(name, 0.0) + now - last[0]
    last[0] = now
    return orig(status, name)


for rep in range(2):
    acc.clear()
    bp.check = check
    torch.cuda.synchronize()
    t0 = last[0] = time.perf_counter()
    bp.collect_predictions(cls, pred, prob, xyz)
    torch.cuda.synchronize()
    total = time.perf_counter() - t0
    bp.check = orig
print(f"{n} queries, {cls.shape[0]} classified points: {total * 1e3:.1f} ms (with a sync after every call); time up to and including each call:")
for k, v in acc.items():
    print(f"  {k:14s} {v * 1e3:8.1f} ms")
print(f"  {'rest':14s} {(total - sum(acc.values())) * 1e3:8.1f} ms")
