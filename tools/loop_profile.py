#!/usr/bin/env python3
"""Where does a batch of the reference-shaped loop go?  

The bench line's ``batch_size_8_loop`` = BalancedBatchSampler(8) + DataLoader + one ``classify_batch`` (forward + D2H) per batch
over plot voxels.  This script builds the same voxel list (a smaller plot by default), runs the loop with a wall-clock timer
around every stage (synchronising after each, so a stage's time is its own), then once more without the extra synchronisation,
then under cProfile.

    python tools/loop_profile.py [plot_points] [batch_size] [--threads N]
"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from pointstowood_amd import DataLoader, Net, predicter
from pointstowood_amd import engine as E
from pointstowood_amd import synthetic_weights as weights
from pointstowood_amd.preprocessing import voxelise
from pointstowood_amd.synthetic_voxels import forest_plot

args = [a for a in sys.argv[1:] if not a.startswith("--")]
n_plot = int(args[0]) if args else 2_500_000
bs = int(args[1]) if len(args) > 1 else 8
if "--threads" in sys.argv:
    torch.set_num_threads(int(sys.argv[sys.argv.index("--threads") + 1]))
dev = torch.device("cuda")
print(f"torch threads {torch.get_num_threads()}, host cores {os.cpu_count()}", flush=True)
net = Net(num_classes=1, C=32, k=32)
net.load_state_dict(weights.synth_state_dict(1, 32, seed=0), strict=True)
net = net.to(dev).eval()
side = max(10.0, 100.0 * (n_plot / 10_000_000) ** 0.5)
pc = forest_plot(n_plot, side=side).to(dev)
vox, _ = voxelise(pc, (2.0, 4.0), 128, 16384, generator=torch.Generator(device=dev).manual_seed(0))
host_vox = [v.cpu() for v in vox]
del vox, pc
sub = predicter.VoxelDataset(host_vox[::4])
print(f"{len(sub)} voxels, {sum(len(v) for v in sub._mem)} points, median {sorted(len(v) for v in sub._mem)[len(sub) // 2]}", flush=True)


def loop(sync_stages, acc):
    sampler = predicter.BalancedBatchSampler(sub, bs)
    it = iter(DataLoader(sub, batch_sampler=sampler, num_workers=0))
    n = nb = 0
    sync = torch.cuda.synchronize if sync_stages else (lambda: None)
    while True:
        t0 = time.perf_counter()
        data = next(it, None)
        if data is None:
            break
        t1 = time.perf_counter()
        data = data.to(dev)
        sync()
        t2 = time.perf_counter()
        logits = net(data)
        sync()
        t3 = time.perf_counter()
        probs = torch.sigmoid(torch.nan_to_num(logits)).reshape(-1)
        preds = (probs >= 0.5).to(torch.int64)
        rows = predicter._rows(data, preds, probs)
        sync()
        t4 = time.perf_counter()
        out = rows.cpu().numpy()
        t5 = time.perf_counter()
        for k, v in (("loader", t1 - t0), ("h2d", t2 - t1), ("forward", t3 - t2), ("post", t4 - t3), ("d2h", t5 - t4)):
            acc[k] = acc.get(k, 0.0) + v
        n += out.shape[0]
        nb += 1
    return n, nb


loop(False, {})   # allocator, table state
for sync_stages in (True, False):
    acc = {}
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n, nb = loop(sync_stages, acc)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"sync_stages={sync_stages}: {nb} batches, {n} points, {dt:.3f} s = {dt / nb * 1e3:.2f} ms/batch, {n / dt / 1e6:.3f} M points/s; "
          + ", ".join(f"{k} {v / nb * 1e3:.3f}" for k, v in acc.items()) + " (ms per batch)", flush=True)

# inside the forward
tacc = {}


def wrap(obj, name, key):
    orig = getattr(obj, name)

    def f(self, *a, **k):
        t0 = time.perf_counter()
        r = orig(self, *a, **k)
        tacc[key] = tacc.get(key, 0.0) + time.perf_counter() - t0
        return r
    setattr(obj, name, f)


wrap(E.Engine, "_geometry_async", "geo_launch")
wrap(E.Engine, "_geometry_finish", "size_wait")
wrap(E.Engine, "features", "feat_launch")
wrap(Net, "_inputs", "inputs")
acc = {}
n, nb = loop(False, acc)
torch.cuda.synchronize()
print("inside forward (ms per batch): " + ", ".join(f"{k} {v / nb * 1e3:.3f}" for k, v in tacc.items()), flush=True)
eng = net._engine
print("table scale", eng._table_scale, "rest", eng._table_rest, "probe_rest", eng._table_probe_rest, "range_fallbacks", eng.range_fallbacks, flush=True)

pr = cProfile.Profile()
pr.enable()
loop(False, {})
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(30)
st.sort_stats("cumulative").print_stats(25)
