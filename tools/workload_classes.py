#!/usr/bin/env python3
"""ns per point and kernel class of one sequential forward on each bench workload (diagnostic: anything a workload pays that the
bench batch does not shows up as a class whose per-point time differs)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pointstowood_amd import synthetic_weights as weights, Net
from pointstowood_amd import synthetic_voxels as synth

dev = torch.device("cuda", 0)
net = Net(1, C=bench.C, k=bench.K_NBR)
net.load_state_dict(weights.synth_state_dict(1, bench.C, seed=0))
net = net.to(dev).eval()
cases = [("bench B=8 x 16384", lambda: [synth.uniform_voxel(2.0, bench.NPTS, 123 + i, False) for i in range(8)]),
         ("configs[2] B=64 x 16384 refl", lambda: [synth.uniform_voxel(2.0, bench.NPTS, 200 + i, True) for i in range(64)]),
         ("configs[4] B=128 mixed", lambda: [synth.uniform_voxel(2.0, n, 400 + i, True) for i, n in enumerate(synth.mixed_sizes())]),
         ("surface B=8 x 16384", lambda: [synth.surface_voxel(2.0, bench.NPTS, 300 + i, False) for i in range(8)]),
         ("surface B=64 x 16384", lambda: [synth.surface_voxel(2.0, bench.NPTS, 300 + i, False) for i in range(64)]),
         ("small voxels B=512 x 1024", lambda: [synth.uniform_voxel(2.0, 1024, 900 + i, False) for i in range(512)])]
for name, make in cases:
    d = bench.device_feed(make(), dev)
    n = int(d.pos.shape[0])
    for _ in range(2):
        net(d)
    per, geo = bench.profile_step(net, d)
    tot = sum(v[0] for v in per.values())
    print(f"{name:30s} {n:8d} pts, levels {[geo.levels[l].n for l in (1, 2, 3)]}: {tot:7.2f} ms = {tot / n * 1e6:6.1f} ns/pt | "
          + ", ".join(f"{k.replace('_kernel', '')} {v[0] / n * 1e6:.2f}" for k, v in sorted(per.items(), key=lambda kv: -kv[1][0])[:11]), flush=True)
    del d
    torch.cuda.empty_cache()
