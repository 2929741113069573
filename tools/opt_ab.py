#!/usr/bin/env python3
"""Same-box A/B of one EngineOptions setting against the defaults: logits equal?, class times of a sequential bench step (HIP
events, median of 7), pipelined step time (two rounds, alternating).   python tools/opt_ab.py key=value [precision]
e.g. tools/opt_ab.py fp1_cell_order=0"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from pointstowood_amd import Net
from pointstowood_amd import synthetic_voxels as synth, synthetic_weights as weights

opt = bench.engine_options([sys.argv[1]])
prec = sys.argv[2] if len(sys.argv) > 2 else "f16x3"
dev = torch.device("cuda")
nets = {}
for name, kw in (("default", {}), (sys.argv[1], opt)):
    net = Net(num_classes=1, C=32, k=32, precision=prec, **kw)
    net.load_state_dict(weights.synth_state_dict(1, 32, seed=0), strict=True)
    nets[name] = net.to(dev).eval()
batches = [bench.make_batch(0, dev, j) for j in range(4)]
surf = bench.device_feed([synth.surface_voxel(2.0, 16384, 300 + i, False) for i in range(8)], dev)
mixed = bench.device_feed([synth.uniform_voxel(2.0, n, 400 + i, True) for i, n in enumerate(synth.mixed_sizes(24, 200, 9000, seed=3))], dev)
for d, what in ((batches[0], "uniform"), (surf, "surface"), (mixed, "mixed sizes + reflectance")):
    a, b = (n(d) for n in nets.values())
    torch.cuda.synchronize()
    print(f"{what}: logits bit-identical: {torch.equal(a, b)}  max |d| {float((a - b).abs().max()):.3e}", flush=True)
keys = ("gemm_kernel", "sa_conv_kernel", "interp_concat", "stem", "knn2")
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
        print(f"{name:22s} " + "  ".join(f"{k_} {per[k_][0]:.3f}" for k_ in keys if k_ in per) + f"  | pipelined step {dt:.3f} ms", flush=True)
