#!/usr/bin/env python3
"""One process = one library: pipelined ms per step (Net.stream over the 8 bench batches), lone-forward ms, sequential class times.
Alternate processes of two libraries for a same-box A/B:   python tools/step_ab.py [build_variants/x.so] [key=value ...]"""
import os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointstowood_amd import _lib as _libmod
args = sys.argv[1:]
if args and args[0].endswith(".so"):
    _libmod.LIB_PATH = os.path.abspath(args.pop(0))
    from pointstowood_amd import build as _b
    _b._stale = lambda: False
import torch
import bench
from pointstowood_amd import Net, synthetic_weights as weights
dev = torch.device("cuda")
net = Net(num_classes=1, C=32, k=32, **bench.engine_options(args))
net.load_state_dict(weights.synth_state_dict(1, 32, seed=0), strict=True)
net = net.to(dev).eval()
data = [bench.make_batch(0, dev, j) for j in range(8)]
for d in data:
    net(d)
for _ in net.stream(data[i % 8] for i in range(10)):
    pass
torch.cuda.synchronize()
pipe, lone = [], []
for r in range(5):
    t0 = time.perf_counter()
    for _ in net.stream(data[i % 8] for i in range(24)):
        pass
    torch.cuda.synchronize()
    pipe.append((time.perf_counter() - t0) / 24 * 1e3)
    t0 = time.perf_counter()
    for i in range(16):
        net(data[i % 8])
        torch.cuda.synchronize()
    lone.append((time.perf_counter() - t0) / 16 * 1e3)
per, _ = bench.profile_step(net, data[0], reps=5)
print(f"{os.path.basename(_libmod.LIB_PATH):22s} {' '.join(args):24s} pipelined {statistics.median(pipe):.3f} ms  lone {statistics.median(lone):.3f} ms  | seq classes: "
      + "  ".join(f"{k} {per[k][0]:.3f}" for k in ("gemm_kernel", "sa_conv_kernel", "knn", "knn2", "ball_query") if k in per), flush=True)
