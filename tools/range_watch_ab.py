#!/usr/bin/env python3
"""What the GEMM epilogue's range watch costs, piece by piece: the GEMM class' time of a sequential bench step (median of 9) under
variant libraries built with -DP2W_RANGE_AB=1 (no compares), 2 (no report stores), 3 (no tracking at all); tools/build_variant.sh
NAME "-DP2W_RANGE_AB=n" feat.   python tools/range_watch_ab.py build_variants/x.so"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointstowood_amd import _lib as _libmod
if len(sys.argv) > 1:
    _libmod.LIB_PATH = os.path.abspath(sys.argv[1])
    from pointstowood_amd import build as _b
    _b._stale = lambda: False
import torch
import bench
from pointstowood_amd import Net, synthetic_weights as weights
dev = torch.device("cuda")
net = Net(num_classes=1, C=32, k=32)
net.load_state_dict(weights.synth_state_dict(1, 32, seed=0), strict=True)
net = net.to(dev).eval()
data = bench.make_batch(0, dev, 0)
net(data)   # (builds the engine)
net._engine.range_violations = lambda watch: []   # (the ablated builds report nothing: no fallback, the kernels are what is timed)
per, _ = bench.profile_step(net, data, reps=9)
print(f"{os.path.basename(_libmod.LIB_PATH):24s} gemm {per['gemm_kernel'][0]:.3f} ms  sa_conv {per['sa_conv_kernel'][0]:.3f} ms", flush=True)
