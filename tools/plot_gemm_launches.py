#!/usr/bin/env python3
"""Every GEMM launch of one sequential forward on a point-budget batch of the configs[3] plot, grouped by layer shape: launches,
rows, time (one HIP-event bracket per launch, median of 3 forwards) and algorithmic TFLOP/s - where the plot's GEMM class loses
against the bench batch's.   python tools/plot_gemm_launches.py [budget=2097152] [key=value ...]"""
import collections, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from pointstowood_amd import Net, synthetic_weights as weights
from pointstowood_amd import engine as eng_mod
from pointstowood_amd.synthetic_voxels import forest_plot
from pointstowood_amd.predicter import PointBudgetSampler, collate_device
from pointstowood_amd.preprocessing import voxelise

args = sys.argv[1:]
budget = int(args.pop(0)) if args and args[0].isdigit() else 2097152
dev = torch.device("cuda", 0)
net = Net(1, C=32, k=32, **bench.engine_options(args))
net.load_state_dict(weights.synth_state_dict(1, 32, seed=0))
net = net.to(dev).eval()
pc = forest_plot(10_000_000, side=100.0).to(dev)
vox, _ = voxelise(pc, (2.0, 4.0), 128, 16384, generator=torch.Generator(device=dev).manual_seed(0))
lengths = [int(v.shape[0]) for v in vox]
batches = list(PointBudgetSampler(lengths, budget, budget // 1024))
data = collate_device([vox[i] for i in batches[len(batches) // 2]])
del pc, vox
E = eng_mod.Engine
orig = E._gemm_h2
shapes = []


def spy(self, name, A, ldh_a, M, lin, *a, **kw):
    if self.events is not None:
        shapes.append((name, int(M), int(lin.K), int(lin.N), "interp" if kw.get("interp") is not None else ""))
    return orig(self, name, A, ldh_a, M, lin, *a, **kw)


E._gemm_h2 = spy
net(data); net(data)
eng = net._engine
eng.res_streams = 1
runs = []
for rep in range(3):
    shapes.clear()
    eng.events, eng.events_grouped = [], False
    net(data)
    torch.cuda.synchronize()
    ev, eng.events = eng.events, None
    runs.append([s.elapsed_time(e) * 1e3 for n, s, e in ev if n in ("gemm_hoist", "gemm_res", "gemm_mlp")])
med = [statistics.median(r[i] for r in runs) for i in range(len(runs[0]))]
grp = collections.OrderedDict()
for i, t in enumerate(med):
    name, M, K, N, tag = shapes[i] if i < len(shapes) else ("head", 0, 512, 512, "")
    g = grp.setdefault((name, K, N, tag), [0, 0, 0.0, []])
    g[0] += 1; g[1] += M; g[2] += t; g[3].append(M)
print(f"batch: {int(data.pos.shape[0])} points, {int(data.ptr.numel()) - 1} voxels; {len(med)} GEMM launches, {sum(med) / 1e3:.2f} ms")
print(f"{'layer':11s} {'K':>5s} {'N':>5s} {'':7s} {'launches':>8s} {'rows':>9s} {'ms':>8s} {'TF':>6s}  rows per launch")
for (name, K, N, tag), (n, rows, t, ms) in grp.items():
    tf = 2.0 * rows * K * N / (t * 1e-6) / 1e12 if rows else 0.0
    sizes = sorted(set(ms))
    print(f"{name:11s} {K:5d} {N:5d} {tag:7s} {n:8d} {rows:9d} {t / 1e3:8.3f} {tf:6.0f}  {sizes[:3]}{'...' if len(sizes) > 3 else ''}")
