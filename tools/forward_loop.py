#!/usr/bin/env python3
"""N lone forwards (model(data) + synchronise) of one workload on one batch: the process tools/launch_count.sh traces.
    python tools/forward_loop.py [uniform|small] [N]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from pointstowood_amd import Net, synthetic_voxels as synth, synthetic_weights as weights
wl = sys.argv[1] if len(sys.argv) > 1 else "uniform"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda")
net = Net(num_classes=1, C=32, k=32)
net.load_state_dict(weights.synth_state_dict(1, 32, seed=0), strict=True)
net = net.to(dev).eval()
if wl == "small":
    data = bench.device_feed([synth.uniform_voxel(2.0, 1355, 100 + i, False) for i in range(8)], dev)
else:
    data = bench.make_batch(0, dev)
for _ in range(3 + n):
    net(data)
    torch.cuda.synchronize()
