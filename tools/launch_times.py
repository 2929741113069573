#!/usr/bin/env python3
"""Per-launch HIP-event times of one forward on the bench workload (diagnostic)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pointstowood_amd import synthetic_weights as weights
from pointstowood_amd import Net

dev = torch.device("cuda", 0)
net = Net(num_classes=1, C=bench.C, k=bench.K_NBR, precision=os.environ.get("P2W_PRECISION", "f16x3")).to(dev).eval()
net.load_state_dict(weights.synth_state_dict(1, bench.C, seed=0), strict=True)
net = net.to(dev)
# LT_WORKLOAD: uniform (the bench batch, default) | surface | config2 | config4 | small
from pointstowood_amd import synthetic_voxels as synth
wl = os.environ.get("LT_WORKLOAD", "uniform")
if wl == "surface":
    data = bench.device_feed([synth.surface_voxel(2.0, bench.NPTS, 300 + i, False) for i in range(bench.BATCH)], dev)
elif wl == "config2":
    data = bench.device_feed([synth.uniform_voxel(2.0, bench.NPTS, 200 + i, True) for i in range(64)], dev)
elif wl == "small":    # the reference's default --batch_size 8 on plot-sized voxels
    data = bench.device_feed([synth.uniform_voxel(2.0, 1355, 100 + i, False) for i in range(bench.BATCH)], dev)
elif wl == "config4":
    data = bench.device_feed([synth.uniform_voxel(2.0, n, 400 + i, True) for i, n in enumerate(synth.mixed_sizes())], dev)
else:
    data = bench.make_batch(0, dev)
for _ in range(3):
    net(data)
best = None
net._engine.res_streams = 1   # one kernel in flight while timing
for rep in range(5):
    eng = net._engine
    eng.events = []
    net(data)
    torch.cuda.synchronize()
    ev, eng.events = eng.events, None
    t = [(n, s.elapsed_time(e)) for n, s, e in ev]
    best = t if best is None else [(n, min(a, b)) for (n, a), (_, b) in zip(best, t)]
only = sys.argv[1:] 
tot = 0.0
for n, ms in best:
    tot += ms
    if not only or any(n.startswith(o) for o in only):
        print(f"{n:16s} {ms*1e3:8.1f} us")
print("total", round(tot, 3), "ms")
