#!/usr/bin/env python3
"""In-process interleaved A/B of engine knobs on the feature phase (MI355X, BASELINE configs[1]).
    python tools/ab_features.py res_chunk_rows 0 32768 65536 131072
"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from pointstowood_amd import synthetic_weights as weights  # noqa: E402
from pointstowood_amd import Net  # noqa: E402

knob, values = sys.argv[1], [int(v) for v in sys.argv[2:]]
dev = torch.device("cuda", 0)
net = Net(1, C=bench.C, k=bench.K_NBR)
net.load_state_dict(weights.synth_state_dict(1, bench.C, seed=0))
net = net.to(dev).eval()
d = bench.make_batch(0, dev)
net(d)
eng = net._engine
geo = eng.geometry(d.pos, d.reflectance, d.ptr.to(torch.int32), d.sf)
res = {v: [] for v in values}
for rnd in range(8):
    for v in values:
        setattr(eng, knob, v)
        eng.features(geo)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(3):
            eng.features(geo)
        e.record()
        torch.cuda.synchronize()
        res[v].append(s.elapsed_time(e) / 3)
for v in values:
    print(f"{knob}={v}: median {statistics.median(res[v]):.3f} ms  min {min(res[v]):.3f} ms  (features only)")
