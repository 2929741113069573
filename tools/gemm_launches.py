#!/usr/bin/env python3
"""Every GEMM launch of one sequential forward on the bench batch with its own HIP-event bracket: shape, time and algorithmic
TFLOP/s under the library's tile choice (flags 0) and with each tile forced (P2W_GEMM_TILE_128 = 1, P2W_GEMM_TILE_256 = 2).
Median of 5 forwards per setting.   python tools/gemm_launches.py [variant.so] [B points_per_voxel]"""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointstowood_amd import _lib as _libmod
VARIANT = None
if len(sys.argv) > 1 and sys.argv[1].endswith(".so"):   # a variant library (tools/build_variant.sh): the library's choice only
    VARIANT = sys.argv.pop(1)
    _libmod.LIB_PATH = os.path.abspath(VARIANT)
    from pointstowood_amd import build as _b
    _b._stale = lambda: False
import torch
import bench
from pointstowood_amd import Net, synthetic_weights as weights
from pointstowood_amd import engine as eng_mod

dev = torch.device("cuda")
if len(sys.argv) > 2:   # B points_per_voxel: a small batch instead of the bench batch
    from pointstowood_amd import synthetic_voxels as synth
    data = bench.device_feed([synth.uniform_voxel(2.0, int(sys.argv[2]), 100 + i, False) for i in range(int(sys.argv[1]))], dev)
else:
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
FLAGSETS = (0, 0, 0, 0) if VARIANT else (0, 1, 2, 1 << 24)
if os.environ.get("FLAGSETS"):   # four flag sets of one's own, e.g. FLAGSETS=0,512,768,1024 (start stagger 0 / 4 / 6 / 8 us: bits 8..13)
    FLAGSETS = tuple(int(v) for v in os.environ["FLAGSETS"].split(","))   # library's choice, forced 128 x 128, forced 256 x 256 (a variant library: three times the library's choice)
for flags in FLAGSETS:
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
print(f"{'launch':11s} {'M':>7s} {'K':>5s} {'N':>5s} | {'default us':>10s} {'TF':>6s} | {'128^2':>7s} {'256^2':>7s} {'64x128':>7s}")
tot = [0.0] * len(FLAGSETS)
n_sh = len(sh)
for i in range(len(t0)):
    name, M, K, N = sh[i] if i < n_sh else ("head", data.pos.shape[0], 512, 512)
    tf = 2.0 * M * K * N / (t0[i] * 1e-6) / 1e12
    ts = [res[f][0][i] if i < len(res[f][0]) else float("nan") for f in FLAGSETS]
    for j, t in enumerate(ts):
        tot[j] += t
    mark = " <-128" if abs(ts[0] - ts[1]) < abs(ts[0] - ts[2]) else ""
    print(f"{name:11s} {M:7d} {K:5d} {N:5d} | {ts[0]:10.1f} {tf:6.0f} | {ts[1]:7.1f} {ts[2]:7.1f} {ts[3]:7.1f}{mark}{' <-64' if ts[3] < 0.97 * ts[0] else ''}")
print("sum (us): " + ", ".join(f"flags {f}: {t:.0f}" for f, t in zip(FLAGSETS, tot)) + ", best per launch %.0f" % sum(min(res[f][0][i] for f in FLAGSETS) for i in range(len(t0))))
