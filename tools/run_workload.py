#!/usr/bin/env python3
"""Run one of the bench's workloads for N sequential forwards (for rocprofv3): uniform | surface | config2 | config4.
    python3 tools/run_workload.py config2 3 [precision]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from pointstowood_amd import Net  # noqa: E402
from pointstowood_amd import synthetic_voxels as synth  # noqa: E402
from pointstowood_amd import synthetic_weights as weights  # noqa: E402

wl, n = sys.argv[1], int(sys.argv[2])
prec = sys.argv[3] if len(sys.argv) > 3 else "f16x3"
dev = torch.device("cuda", 0)
net = Net(num_classes=1, C=bench.C, k=bench.K_NBR, precision=prec)
net.load_state_dict(weights.synth_state_dict(1, bench.C, seed=0), strict=True)
net = net.to(dev).eval()
if wl == "surface":
    data = bench.device_feed([synth.surface_voxel(2.0, bench.NPTS, 300 + i, False) for i in range(bench.BATCH)], dev)
elif wl == "config2":
    data = bench.device_feed([synth.uniform_voxel(2.0, bench.NPTS, 200 + i, True) for i in range(64)], dev)
elif wl == "config4":
    data = bench.device_feed([synth.uniform_voxel(2.0, m, 400 + i, True) for i, m in enumerate(synth.mixed_sizes())], dev)
else:
    data = bench.make_batch(0, dev)
for _ in range(n):
    out = net(data)
torch.cuda.synchronize()
print(wl, n, "forwards of", int(data.pos.shape[0]), "points; finite:", bool(torch.isfinite(out).all()))
