#!/usr/bin/env python3
"""Class times of one sequential bench step for a given library build (timing-only variants give wrong results):
    python tools/sa_time.py [lib.so] [--spec]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointstowood_amd import _lib
args = [a for a in sys.argv[1:] if not a.startswith("--")]
if args:
    _lib.LIB_PATH = os.path.abspath(args[0])
import bench
from pointstowood_amd import Net
from pointstowood_amd import synthetic_weights as weights
dev = torch.device("cuda")
data = bench.make_batch(0, dev, 0)
for spec in ((False, True) if "--both" in sys.argv else ("--spec" in sys.argv,)):
    net = Net(num_classes=1, C=32, k=32, sa_specialized=spec)
    net.load_state_dict(weights.synth_state_dict(1, 32, seed=0), strict=True)
    net = net.to(dev).eval()
    net(data)
    per, _ = bench.profile_step(net, data, reps=7)
    print(f"{args[0] if args else 'in-tree'} {'specialised' if spec else 'production '}: sa_conv {per['sa_conv_kernel'][0]:.3f} ms  gemm {per['gemm_kernel'][0]:.3f} ms", flush=True)
