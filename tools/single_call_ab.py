#!/usr/bin/env python3
"""Same-box, same-process A/B of the lone forward (model(data) + synchronise per step) under EngineOptions variants: several
Nets, measured in alternation (ROUNDS rounds of STEPS steps each, 8 distinct bench batches), median per variant.
    python tools/single_call_ab.py "overlap=0" "overlap=1,early_first=0" "" ...     ("" = the defaults)
    SC_WORKLOAD=small: B = 8 x 1355"""
import os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from pointstowood_amd import Net, synthetic_voxels as synth, synthetic_weights as weights

variants = sys.argv[1:] or ["overlap=0", ""]
dev = torch.device("cuda")
small = os.environ.get("SC_WORKLOAD") == "small"
if small:
    data = [bench.device_feed([synth.uniform_voxel(2.0, 1355, 100 * j + i, False) for i in range(8)], dev) for j in range(4)]
else:
    data = [bench.make_batch(0, dev, j) for j in range(8)]
nets = []
for v in variants:
    kw = bench.engine_options([kv for kv in v.split(",") if kv])
    net = Net(num_classes=1, C=32, k=32, **kw)
    net.load_state_dict(weights.synth_state_dict(1, 32, seed=0), strict=True)
    nets.append(net.to(dev).eval())
ref = None
for net in nets:
    outs = [net(d).clone() for d in data]
    if ref is None:
        ref = outs
    else:
        assert all(torch.equal(a, b) for a, b in zip(ref, outs)), "variants disagree"
ROUNDS, STEPS = int(os.environ.get("ROUNDS", "7")), int(os.environ.get("STEPS", "24"))
# CALLER_PRIORITY=-1: the caller's stream (= the feature phase's) is a high-priority one, the searches' stays at search_priority
caller = torch.cuda.Stream(priority=int(os.environ["CALLER_PRIORITY"])) if os.environ.get("CALLER_PRIORITY") else torch.cuda.current_stream()
t = [[] for _ in nets]
with torch.cuda.stream(caller):
    for net in nets:
        for d in data:
            net(d)
    for r in range(ROUNDS):
        for i, net in enumerate(nets):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for s in range(STEPS):
                net(data[s % len(data)])
                torch.cuda.synchronize()
            t[i].append((time.perf_counter() - t0) / STEPS * 1e3)
for v, ts in zip(variants, t):
    print(f"{v or '(defaults)':40s} {statistics.median(ts):7.3f} ms per forward   (rounds: {' '.join(f'{x:.3f}' for x in ts)})")
print("logits bit-identical across the variants")
