#!/usr/bin/env python3
"""Every GEMM launch of one sequential forward on the bench batch with its own HIP-event bracket: shape, time and algorithmic
TFLOP/s under the library's tile choice (flags 0) and with each tile forced (P2W_GEMM_TILE_128 = 1, P2W_GEMM_TILE_256 = 2).
Median of 5 forwards per setting.   python tools/gemm_launches.py"""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from pointstowood_amd import Net, synthetic_weights as weights
from pointstowood_amd import engine as eng_mod

dev = torch.device("cuda")
data = bench.make_batch(0, dev, 0)
E = eng_mod.Engine
orig = E._gemm_h2
shapes = []


def spy(self, name, A, ldh_a, M, lin, *a, **kw):
    if self.events is not None and getattr(self, "_spy_on", False):
        shapes.append((name, int(M), int(lin.K), int(lin.N)))
    return orig(self, name, A, ldh_a, M, lin, *a, **kw)


E._gemm_h2 = spy
res = {}
for flags in (0, 1, 2):
    net = Net(num_classes=1, C=32, k=32, gemm_flags=flags)
    net.load_state_dict(weights.synth_state_dict(1, 32, seed=0), strict=True)
    net = net.to(dev).eval()
    net(data); net(data)
    eng = net._engine
    runs = []
    for rep in range(5):
        shapes.clear()
        eng.events, eng.events_grouped, eng._spy_on = [], False, True
        net(data)
        torch.cuda.synchronize()
        ev, eng.events = eng.events, None
        ts = [s.elapsed_time(e) * 1e3 for n, s, e in ev if n in ("gemm_hoist", "gemm_res", "gemm_mlp")]
        runs.append(ts)
    res[flags] = ([statistics.median(r[i] for r in runs) for i in range(len(runs[0]))], list(shapes))
    del net
t0, sh = res[0]
print(f"{'launch':11s} {'M':>7s} {'K':>5s} {'N':>5s} | {'default us':>10s} {'TF':>6s} | {'128^2 us':>9s} {'256^2 us':>9s}")
tot = [0.0, 0.0, 0.0]
n_sh = len(sh)
for i in range(len(t0)):
    name, M, K, N = sh[i] if i < n_sh else ("head", data.pos.shape[0], 512, 512)
    tf = 2.0 * M * K * N / (t0[i] * 1e-6) / 1e12
    t1 = res[1][0][i] if i < len(res[1][0]) else float("nan")
    t2 = res[2][0][i] if i < len(res[2][0]) else float("nan")
    for j, t in enumerate((t0[i], t1, t2)):
        tot[j] += t
    mark = " <-128" if abs(t0[i] - t1) < abs(t0[i] - t2) else ""
    print(f"{name:11s} {M:7d} {K:5d} {N:5d} | {t0[i]:10.1f} {tf:6.0f} | {t1:9.1f} {t2:9.1f}{mark}")
print("sum (us): default %.0f, forced 128^2 %.0f, forced 256^2 %.0f, best per launch %.0f" % (tot[0], tot[1], tot[2], sum(min(a, b, c) for a, b, c in zip(res[0][0], res[1][0], res[2][0]))))
