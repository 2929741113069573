#!/usr/bin/env python3
"""In-kernel phase profile of the back-projection's k = 64 search (diagnostic build -DP2W_SLAB_PROFILE) on a synthetic plot:
classified set = the plot's points twice (every point sits in a 2 m and a 4 m voxel), queries = the plot's points."""
import ctypes as C, os, sys, time
os.environ.setdefault("P2W_EXTRA_CFLAGS", "-DP2W_SLAB_PROFILE")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointstowood_amd import synthetic_voxels as synth
from pointstowood_amd import backproject as bp
from pointstowood_amd._lib import lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
cell = float(sys.argv[2]) if len(sys.argv) > 2 else None
dev = torch.device("cuda", 0)
side = (n / 1000.0) ** 0.5            # the plot bench's density: 10 M points on 100 m x 100 m
plot = synth.forest_plot(n, seed=1, side=side).to(dev)
xyz = plot[:, :3].contiguous()
cls = torch.cat([xyz, xyz], 0)
L = lib()
L.p2w_debug_slab_prof.argtypes = [C.c_void_p, C.c_int]
buf = (C.c_ulonglong * 16)()
names = ["setup", "probe", "plan", "stage", "scan", "check", "output"]
for rep in range(2):
    L.p2w_debug_slab_prof(buf, 1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for rows, nbr, deg in bp.neighbours(cls, xyz, 64, cell):
        pass
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    L.p2w_debug_slab_prof(buf, 1)
    blocks = max(buf[11], 1)
    print(f"n={n} cell={cell}: {dt*1e3:.1f} ms ({n/dt/1e6:.1f} M queries/s)  blocks={buf[11]} passes/blk={buf[8]/blocks:.2f} "
          f"cand/pass={buf[9]/max(buf[8],1):.1f} active/pass={buf[10]/max(buf[8],1):.1f} | "
          + " ".join(f"{nm}={buf[i]/blocks:.0f}" for i, nm in enumerate(names)))
