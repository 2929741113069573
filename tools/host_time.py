#!/usr/bin/env python3
"""Host-side cost of one pipelined step: time spent launching the geometry / feature phases and waiting for the level sizes."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pointstowood_amd import synthetic_weights as weights
from pointstowood_amd import Net
from pointstowood_amd import engine as E

dev = torch.device("cuda", 0)
net = Net(num_classes=1, C=bench.C, k=bench.K_NBR).to(dev).eval()
net.load_state_dict(weights.synth_state_dict(1, bench.C, seed=0), strict=True)
net = net.to(dev)
data = [bench.make_batch(0, dev, j) for j in range(4)]
for d in data:
    net(d)
acc = {"geo_launch": 0.0, "feat_launch": 0.0, "size_wait": 0.0}
def wrap(name, key):
    orig = getattr(E.Engine, name)
    def f(self, *a, **k):
        t0 = time.perf_counter()
        r = orig(self, *a, **k)
        acc[key] += time.perf_counter() - t0
        return r
    setattr(E.Engine, name, f)
wrap("_geometry_async", "geo_launch"); wrap("features", "feat_launch"); wrap("_geometry_finish", "size_wait")
n = 40
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in net.stream(data[i % 4] for i in range(n)):
    pass
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"{n} steps: host loop {t_host / n * 1e3:.3f} ms/step, with final sync {t_all / n * 1e3:.3f} ms/step; "
      + ", ".join(f"{k} {v / n * 1e3:.3f}" for k, v in acc.items()))
