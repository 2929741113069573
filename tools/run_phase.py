#!/usr/bin/env python3
"""Profiling helper: run only the geometry phase (or the whole forward) of BASELINE configs[1] a few times.
    python tools/run_phase.py geometry|forward [reps]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from pointstowood_amd import synthetic_weights as weights  # noqa: E402
from pointstowood_amd import Net  # noqa: E402

phase = sys.argv[1] if len(sys.argv) > 1 else "forward"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda", 0)
net = Net(1, C=bench.C, k=bench.K_NBR, precision=os.environ.get("P2W_PRECISION", "f16x3"))
net.load_state_dict(weights.synth_state_dict(1, bench.C, seed=0))
net = net.to(dev).eval()
d = bench.make_batch(0, dev)
net(d)
torch.cuda.synchronize()
eng = net._engine
for _ in range(reps):
    if phase == "geometry":
        eng.geometry(d.pos, d.reflectance, d.ptr.to(torch.int32), d.sf)
    else:
        net(d)
torch.cuda.synchronize()
print("done", phase, reps)
