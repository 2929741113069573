#!/usr/bin/env python3
"""Timing-only ablations of the wave-specialised PointNetConv kernel (diagnostic build: tools/build_variant.sh saabl
"-DP2W_SA_ABLATE" feat).  For each ablation mask: the class' time per sequential bench step (median of 5).
    python tools/sa_ablate_s.py build_variants/saabl.so"""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointstowood_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
import bench
from pointstowood_amd import Net
from pointstowood_amd import synthetic_weights as weights

dev = torch.device("cuda")
data = bench.make_batch(0, dev, 0)
names = {0: "all on", 2: "no W2 DMA", 16: "no P gather", 8: "no producer VALU", 4: "no MFMA", 2 | 16: "no DMA, no gather", 2 | 16 | 8: "consumers only",
         4 | 8: "memory streams only (no MFMA, no VALU)", 4 | 8 | 16: "W2 DMA only", 4 | 8 | 2: "P gather only"}
only = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else None   # optional: ablation masks to run
for spec in (True, False):
    for mask, what in names.items():
        if only is not None and mask not in only:
            continue
        net = Net(num_classes=1, C=32, k=32, sa_specialized=spec, sa_flags=mask << 16)
        net.load_state_dict(weights.synth_state_dict(1, 32, seed=0), strict=True)
        net = net.to(dev).eval()
        net(data)
        per, _ = bench.profile_step(net, data, reps=5)
        print(f"{'specialised' if spec else 'production '} {what:45s} sa_conv {per['sa_conv_kernel'][0]:.3f} ms", flush=True)
        del net
