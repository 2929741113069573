#!/usr/bin/env python3
"""Stage times of the spatially owned back-projection for a few ranks of a pretended world (one GPU, replaying stand-in for
torch.distributed as in tools/scaling_predict.py):   python tools/spatial_parts.py [world] [ranks, e.g. 0,3] [points]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import bench
from scaling_predict import FakeDist, _Stop
from pointstowood_amd import Net, pipeline, synthetic_weights as weights
from pointstowood_amd.synthetic_voxels import forest_plot
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ranks = [int(r) for r in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, world // 2]
points = int(sys.argv[3]) if len(sys.argv) > 3 else 10_000_000
dev = torch.device("cuda")
net = Net(num_classes=1, C=bench.C, k=bench.K_NBR)
net.load_state_dict(weights.synth_state_dict(1, bench.C, seed=0), strict=True)
net = net.to(dev).eval()
pc = forest_plot(points, side=max(10.0, 100.0 * (points / 1e7) ** 0.5)).to(dev)
gen = lambda: torch.Generator(device=dev).manual_seed(0)
pipeline.segment_plot(pc, net, generator=gen())
store, log = {}, []
for r in range(world):
    try:
        pipeline.segment_plot(pc, net, generator=gen(), dist=FakeDist(r, world, store, log))
    except _Stop:
        torch.cuda.synchronize()
for r in ranks:
    for rep in range(2):
        st = {}
        pipeline.segment_plot(pc, net, generator=gen(), stats=st, dist=FakeDist(r, world, store, [], stop=False))
        torch.cuda.synchronize()
    print(f"rank {r}/{world}: voxelise {st['voxelise_s']:.3f} classify {st['classify_s']:.3f} backproject {st['backproject_s']:.3f}  parts {st.get('backproject_parts_s')}  tiers {st.get('backproject_tiers')}", flush=True)
# where a tier-1 search goes: the pieces of collect_predictions_checked on rank ranks[-1]'s candidate set
import cProfile, pstats
pr = cProfile.Profile()
pr.enable()
st = {}
pipeline.segment_plot(pc, net, generator=gen(), stats=st, dist=FakeDist(ranks[-1], world, store, [], stop=False))
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
