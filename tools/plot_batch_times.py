#!/usr/bin/env python3
"""Per-kernel-class time of one point-budget batch of a synthetic plot (many small voxels) vs the bench batch."""
import os, sys, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointstowood_amd import synthetic_weights as weights
from pointstowood_amd import Net
from pointstowood_amd.predicter import PointBudgetSampler, collate_device
from pointstowood_amd.preprocessing import voxelise
from tools.run_plot import synth_plot

dev = torch.device("cuda", 0)
net = Net(1, C=32, k=32)
net.load_state_dict(weights.synth_state_dict(1, 32, seed=0))
net = net.to(dev).eval()
pc = synth_plot(2_000_000, side=45.0).to(dev)
vox, _ = voxelise(pc, (2.0, 4.0), 128, 16384, generator=torch.Generator(device=dev).manual_seed(0))
lengths = [int(v.shape[0]) for v in vox]
batches = list(PointBudgetSampler(lengths, 131072))
print(len(batches), "batches; voxels per batch:", sorted(len(b) for b in batches)[::max(1, len(batches)//8)])
for bi in (len(batches) // 2, 0, len(batches) - 2):
    d = collate_device([vox[i] for i in batches[bi]])
    for _ in range(2):
        net(d)
    eng = net._engine
    best = None
    for rep in range(3):
        eng.events = []
        net(d)
        torch.cuda.synchronize()
        ev, eng.events = eng.events, None
        per = collections.OrderedDict()
        for n, s, e in ev:
            per[n] = per.get(n, 0.0) + s.elapsed_time(e)
        best = per if best is None else {k: min(best[k], per[k]) for k in per}
    tot = sum(best.values())
    print(f"batch {bi}: {len(batches[bi])} voxels, {d.pos.shape[0]} pts, kernel sum {tot:.2f} ms = {d.pos.shape[0]/tot/1e3:.2f} M pts/s :: " +
          " ".join(f"{k}={v:.2f}" for k, v in sorted(best.items(), key=lambda kv: -kv[1])[:9]))
