#!/usr/bin/env python3
"""Same-box A/B of the fused PointNetConv kernels: production (8 waves, 64 x 64 wave tiles) against the wave-specialised one
(P2W_SA_SPECIALIZED).  Checks that the logits are bit-identical, then prints the class' time per sequential bench step (HIP
events, median of 7) and the pipelined step time of both.   python tools/sa_ab.py [precision]"""
import os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from pointstowood_amd import Net
from pointstowood_amd import synthetic_weights as weights

prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
dev = torch.device("cuda")
nets = {}
for name, opt in (("production", False), ("specialised", True)):
    net = Net(num_classes=1, C=32, k=32, precision=prec, sa_specialized=opt)
    net.load_state_dict(weights.synth_state_dict(1, 32, seed=0), strict=True)
    nets[name] = net.to(dev).eval()
batches = [bench.make_batch(0, dev, j) for j in range(4)]
surf = bench.device_feed([__import__("pointstowood_amd.synthetic_voxels", fromlist=["x"]).surface_voxel(2.0, 16384, 300 + i, False) for i in range(8)], dev)
for d, what in ((batches[0], "uniform"), (surf, "surface")):
    a, b = nets["production"](d), nets["specialised"](d)
    torch.cuda.synchronize()
    print(f"{what}: logits bit-identical: {torch.equal(a, b)}  max |d| {float((a - b).abs().max()):.3e}", flush=True)
for rnd in range(2):
    for name, net in nets.items():
        per, _ = bench.profile_step(net, batches[0], reps=7)
        for _ in net.stream(batches[i % 4] for i in range(8)):
            pass
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 32
        for _ in net.stream(batches[i % 4] for i in range(n)):
            pass
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n * 1e3
        print(f"{name:12s} sa_conv {per['sa_conv_kernel'][0]:.3f} ms  gemm {per['gemm_kernel'][0]:.3f} ms  pipelined step {dt:.3f} ms", flush=True)
