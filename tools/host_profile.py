#!/usr/bin/env python3
"""cProfile of the host side of the pipelined forward on a small batch (where the host sets the pace)."""
import cProfile, os, pstats, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pointstowood_amd import synthetic_weights as weights, Net
from pointstowood_amd import synthetic_voxels as synth

dev = torch.device("cuda", 0)
net = Net(1, C=32, k=32)
net.load_state_dict(weights.synth_state_dict(1, 32, seed=0))
net = net.to(dev).eval()
B, n = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (8, 1355)
data = [bench.device_feed([synth.uniform_voxel(2.0, n, 100 * j + i, False) for i in range(B)], dev) for j in range(4)]
for d in data:
    net(d)
for _ in net.stream(data[i % 4] for i in range(8)):
    pass
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in net.stream(data[i % 4] for i in range(100)):
    pass
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
