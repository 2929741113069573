#!/usr/bin/env python3
"""Classification stage of the plot pipeline against the forward budget (points / voxels per forward), each budget timed on its
second full pass (the first one sizes the caching allocator's pools)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointstowood_amd import synthetic_weights as weights, Net
from pointstowood_amd.synthetic_voxels import forest_plot
from pointstowood_amd.pipeline import segment_plot

dev = torch.device("cuda", 0)
net = Net(1, C=32, k=32)
net.load_state_dict(weights.synth_state_dict(1, 32, seed=0))
net = net.to(dev).eval()
pc = forest_plot(int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000, side=100.0).to(dev)
gen = lambda: torch.Generator(device=dev).manual_seed(0)
for mp in (131072, 262144, 524288, 1048576, 2097152, 4194304):
    for rep in range(2):
        stats = {}
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        segment_plot(pc, net, generator=gen(), stats=stats, max_points=mp, max_voxels=mp // 1024)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(f"budget {mp:8d} points / {mp // 1024:5d} voxels: classify {stats['classify_s']:.3f} s, back-project {stats['backproject_s']:.3f} s, "
          f"end to end {dt:.3f} s, peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
    torch.cuda.empty_cache(); torch.cuda.reset_peak_memory_stats()
