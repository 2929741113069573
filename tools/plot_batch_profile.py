#!/usr/bin/env python3
"""Per-kernel-class time of ONE forward on a plot's voxel batch, for several forward budgets (sequential, HIP-event brackets):
shows what does not scale linearly with the batch (diagnostic)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pointstowood_amd import synthetic_weights as weights, Net
from pointstowood_amd.synthetic_voxels import forest_plot
from pointstowood_amd.predicter import PointBudgetSampler, collate_device
from pointstowood_amd.preprocessing import voxelise

dev = torch.device("cuda", 0)
net = Net(1, C=32, k=32)
net.load_state_dict(weights.synth_state_dict(1, 32, seed=0))
net = net.to(dev).eval()
pc = forest_plot(10_000_000, side=100.0).to(dev)
vox, _ = voxelise(pc, (2.0, 4.0), 128, 16384, generator=torch.Generator(device=dev).manual_seed(0))
lengths = [int(v.shape[0]) for v in vox]
for mp in (131072, 524288, 2097152):
    batches = list(PointBudgetSampler(lengths, mp, mp // 1024))
    b = batches[len(batches) // 2]
    data = collate_device([vox[i] for i in b])
    n = int(data.pos.shape[0])
    for _ in range(2):
        net(data)
    per, geo = bench.profile_step(net, data)
    tot = sum(v[0] for v in per.values())
    top = sorted(per.items(), key=lambda kv: -kv[1][0])[:9]
    total_macs, kmacs, _ = bench.algorithmic_macs(geo)
    tf = {k: 2.0 * kmacs[k] / (per[k][0] * 1e-3) / 1e12 for k in ("gemm_kernel", "sa_conv_kernel")}
    print(f"budget {mp}: batch of {len(b)} voxels / {n} points, levels {[geo.levels[l].n for l in (1, 2, 3)]}: {tot:.2f} ms = {tot / n * 1e6:.2f} ns/point "
          f"(GEMM class {tf['gemm_kernel']:.0f} TF, PointNetConv {tf['sa_conv_kernel']:.0f} TF) | "
          + ", ".join(f"{k} {v[0] / n * 1e6:.2f}" for k, v in top), flush=True)

if os.environ.get("SLAB"):      # in-kernel phase profile of the searches (build with P2W_EXTRA_CFLAGS=-DP2W_SLAB_PROFILE)
    import ctypes as C
    from pointstowood_amd import engine as E
    from pointstowood_amd._lib import lib
    L = lib()
    L.p2w_debug_slab_prof.argtypes = [C.c_void_p, C.c_int]
    buf = (C.c_ulonglong * 16)()
    names = ["setup", "probe", "plan", "stage", "scan", "check", "output"]
    orig = E.Engine._call

    def call(self, name, fn, *args):
        if name in ("knn", "knn2", "ball_query"):
            torch.cuda.synchronize()
            L.p2w_debug_slab_prof(buf, 1)
        r = orig(self, name, fn, *args)
        if name in ("knn", "knn2", "ball_query"):
            torch.cuda.synchronize()
            L.p2w_debug_slab_prof(buf, 1)
            tot = sum(buf[i] for i in range(7))
            print(f"   {name:10s} builds={buf[8]:8d} cand/build={buf[9] / max(buf[8], 1):8.1f} active/build={buf[10] / max(buf[8], 1):5.1f} cycles/build={tot / max(buf[8], 1):9.0f} | "
                  + " ".join(f"{n}={100 * buf[i] / max(tot, 1):4.1f}%" for i, n in enumerate(names)), flush=True)
        return r
    E.Engine._call = call
    for mp in (131072, 2097152):
        batches = list(PointBudgetSampler(lengths, mp, mp // 1024))
        data = collate_device([vox[i] for i in batches[len(batches) // 2]])
        net(data)
        print(f"budget {mp}: {int(data.pos.shape[0])} points")
        net(data)
