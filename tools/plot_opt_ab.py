#!/usr/bin/env python3
"""One EngineOptions setting against the defaults on the configs[3] plot (pipeline.segment_plot, 10 M points): stage times of
alternating runs (best of 3 each).   python tools/plot_opt_ab.py key=value [points]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from pointstowood_amd import Net, pipeline, synthetic_weights as weights
from pointstowood_amd.synthetic_voxels import forest_plot
opt = bench.engine_options([sys.argv[1]])
points = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
dev = torch.device("cuda", 0)
nets = {}
for name, kw in (("default", {}), (sys.argv[1], opt)):
    net = Net(num_classes=1, C=32, k=32, **kw)
    net.load_state_dict(weights.synth_state_dict(1, 32, seed=0), strict=True)
    nets[name] = net.to(dev).eval()
pc = forest_plot(points, side=100.0 * (points / 10_000_000) ** 0.5).to(dev)
gen = lambda: torch.Generator(device=dev).manual_seed(0)
best = {n: {} for n in nets}
for rnd in range(4):
    for name, net in nets.items():
        st = {}
        pipeline.segment_plot(pc, net, generator=gen(), stats=st)
        torch.cuda.synchronize()
        if rnd:   # (round 0 sizes the allocator)
            for k in ("voxelise_s", "classify_s", "backproject_s"):
                best[name][k] = min(best[name].get(k, 1e9), st[k])
for name, b in best.items():
    print(f"{name:24s} " + "  ".join(f"{k} {v:.4f}" for k, v in b.items()) + f"  | total {sum(b.values()):.4f} s")
