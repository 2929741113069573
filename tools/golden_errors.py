#!/usr/bin/env python3
"""Max |logit - reference logit| over the golden cases, both precision modes (diagnostic for DESIGN.md)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointstowood_amd import synthetic_weights as weights
from pointstowood_amd import Net
from tests import golden_util as G

for prec in ("f16x3", "fp32", "fp16", "bf16"):
    worst = 0.0
    for name in G.CASES:
        g, inp, meta = G.load(name)
        net = Net(1, C=meta["C"], k=meta["k"], precision=prec)
        net.load_state_dict(weights.synth_state_dict(1, meta["C"], seed=meta["wseed"]))
        net = net.cuda().eval()

        class D: pass
        d = D()
        for k_, v in inp.items():
            setattr(d, k_, v.cuda())
        out = net(d).cpu().numpy().astype(np.float64)
        key = "out.logits" if "out.logits" in g or "out.logits__sample" in g else [k for k in g if "logit" in k][0].split("__")[0]
        if key in g:
            err = np.abs(out - g[key].astype(np.float64)).max()
        else:
            rows = g[key + "__rows"].astype(np.int64)
            err = np.abs(out[rows] - g[key + "__sample"].astype(np.float64)).max()
        worst = max(worst, err)
        print(f"{prec:6s} {name:22s} max |dlogit| = {err:.2e}")
    print(f"{prec}: worst {worst:.2e}")
