#!/usr/bin/env python3
"""Randomised equality of the three ways a batch can be run: lone forward with the searches beside the features (default), lone
forward strictly sequential (overlap=0, one chunk chain), Net.stream over the same sequence.  Ragged batches, 2 m and 4 m voxels
(table overflow -> geometry redone while searches are in flight), surface voxels, tiny voxels.   python tools/overlap_fuzz.py [N=60] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from pointstowood_amd import Net, synthetic_voxels as synth, synthetic_weights as weights
n_batches = int(sys.argv[1]) if len(sys.argv) > 1 else 60
g = torch.Generator().manual_seed(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device("cuda")
C = 8
nets = {}
for name, kw in (("overlap", {}), ("sequential", dict(overlap=False, single_res_streams=1)), ("stream", {})):
    net = Net(num_classes=1, C=C, k=32, **kw)
    net.load_state_dict(weights.synth_state_dict(1, C, seed=1), strict=True)
    nets[name] = net.to(dev).eval()
batches = []
for i in range(n_batches):
    B = int(torch.randint(1, 9, (1,), generator=g))
    vox = []
    for b in range(B):
        kind = float(torch.rand(1, generator=g))
        n = int(2 ** (float(torch.rand(1, generator=g)) * 7.5 + 6))          # 64 .. 11585
        seed = 1000 * i + b
        if kind < 0.15:
            vox.append(synth.uniform_voxel(4.0, n, seed, True))
        elif kind < 0.3:
            vox.append(synth.surface_voxel(2.0, max(n, 256), seed, True))
        elif kind < 0.4:
            vox.append(synth.uniform_voxel(0.3, min(n, 200), seed, False))
        else:
            vox.append(synth.uniform_voxel(2.0, n, seed, bool(b % 2)))
    batches.append(bench.device_feed(vox, dev))
ref = [nets["sequential"](d).clone() for d in batches]
got = [nets["overlap"](d).clone() for d in batches]
torch.cuda.synchronize()
bad = [i for i, (a, b) in enumerate(zip(ref, got)) if not torch.equal(a, b)]
st = [o.clone() for o in nets["stream"].stream(iter(batches))]
torch.cuda.synchronize()
bad_s = [i for i, (a, b) in enumerate(zip(ref, st)) if not torch.equal(a, b)]
pts = sum(int(d.pos.shape[0]) for d in batches)
print(f"{n_batches} batches, {pts} points: lone-overlap != sequential in {len(bad)} batches {bad[:5]}; stream != sequential in {len(bad_s)} {bad_s[:5]}; "
      f"range fallbacks {[n._engine.range_fallbacks for n in nets.values()]}; table scale {nets['overlap']._engine._table_scale}")
sys.exit(1 if (bad or bad_s) else 0)
