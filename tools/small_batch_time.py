#!/usr/bin/env python3
"""Small batches (the reference's default --batch_size 8 on plot voxels: ~1355 points each): pipelined time per batch, the host's
launch time per phase and the wait for the level sizes, GPU time of the two phases (sum of kernel times, HIP events).
    python tools/small_batch_time.py [B] [points_per_voxel]"""
import os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from pointstowood_amd import Net, synthetic_voxels as synth, synthetic_weights as weights
from pointstowood_amd import engine as E

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1355
dev = torch.device("cuda")
net = Net(num_classes=1, C=32, k=32)
net.load_state_dict(weights.synth_state_dict(1, 32, seed=0), strict=True)
net = net.to(dev).eval()
data = [bench.device_feed([synth.uniform_voxel(2.0, n, 100 * j + i, False) for i in range(B)], dev) for j in range(4)]
for d in data:
    net(d)
for _ in net.stream(data[i % 4] for i in range(8)):
    pass
acc = {"geo_launch": 0.0, "feat_launch": 0.0, "size_wait": 0.0}
def wrap(name, key):
    orig = getattr(E.Engine, name)
    def f(self, *a, **k):
        t0 = time.perf_counter(); r = orig(self, *a, **k); acc[key] += time.perf_counter() - t0
        return r
    setattr(E.Engine, name, f)
wrap("_geometry_async", "geo_launch"); wrap("features", "feat_launch"); wrap("_geometry_finish", "size_wait")
N = 60
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in net.stream(data[i % 4] for i in range(N)):
    pass
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / N
print(f"B = {B} x {n}: pipelined {dt * 1e3:.3f} ms/batch ({B * n / dt / 1e6:.2f} M points/s); host per batch: " + ", ".join(f"{k} {v / N * 1e3:.3f} ms" for k, v in acc.items()))
eng = net._engine
geo_names = {"pack_xyzr", "voxel_sample", "index_records", "ball_query", "knn", "knn2", "knn_hint", "level_gather", "tile_bbox"}
g, f, ng, nf = [], [], 0, 0
for rep in range(5):
    eng.events, eng.events_grouped = [], False
    net(data[0]); torch.cuda.synchronize()
    ev, eng.events = eng.events, None
    tg = sum(s.elapsed_time(e) for nme, s, e in ev if nme in geo_names); tf = sum(s.elapsed_time(e) for nme, s, e in ev if nme not in geo_names)
    ng = sum(1 for nme, _, _ in ev if nme in geo_names); nf = len(ev) - ng
    g.append(tg); f.append(tf)
print(f"GPU kernel time per forward (event brackets, incl. ~8 us of dispatch per bracket): geometry {statistics.median(g):.3f} ms in {ng} calls, features {statistics.median(f):.3f} ms in {nf} calls")
