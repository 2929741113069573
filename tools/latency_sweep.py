#!/usr/bin/env python3
"""Forward latency (one batch at a time, synchronised) and pipelined throughput for small batches: where the host, not the GPU,
sets the time (diagnostic)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pointstowood_amd import synthetic_weights as weights, Net
from pointstowood_amd import synthetic_voxels as synth

dev = torch.device("cuda", 0)
net = Net(1, C=32, k=32)
net.load_state_dict(weights.synth_state_dict(1, 32, seed=0))
net = net.to(dev).eval()
for B, n in ((1, 2048), (1, 16384), (8, 1355), (8, 2048), (32, 1355), (8, 16384)):
    data = [bench.device_feed([synth.uniform_voxel(2.0, n, 100 * j + i, False) for i in range(B)], dev) for j in range(4)]
    for d in data:
        net(d)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(20):
        net(data[i % 4]); torch.cuda.synchronize()
    lat = (time.perf_counter() - t0) / 20
    for _ in net.stream(data[i % 4] for i in range(8)):
        pass
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in net.stream(data[i % 4] for i in range(40)):
        pass
    torch.cuda.synchronize()
    thr = (time.perf_counter() - t0) / 40
    print(f"B={B:3d} x {n:6d} points: latency {lat * 1e3:6.2f} ms, pipelined {thr * 1e3:6.2f} ms/batch = {B * n / thr / 1e6:6.2f} M points/s", flush=True)
