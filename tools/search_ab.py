#!/usr/bin/env python3
"""Same-box A/B of the k >= 8 selection of the grid searches: collected candidates + sorting networks (P2W_SEARCH_COLLECT) against
per-candidate sorted insertion (the default).  Times the geometry phase's kernels of the bench batch (HIP events per
kernel class, median of several forwards) and the back-projection's k = 64 search on a synthetic plot.
    python tools/search_ab.py [plot_points] [lib.so ...]
Extra arguments are alternative library builds (e.g. variants built with P2W_EXTRA_CFLAGS) measured after the in-tree one."""
import os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

plot_points = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
libs = [None] + sys.argv[2:]


def run(libpath):
    from pointstowood_amd import _lib
    if libpath:
        _lib.LIB_PATH = os.path.abspath(libpath)
        _lib._lib = None
    from pointstowood_amd import Net, backproject
    from pointstowood_amd import synthetic_weights as weights
    from pointstowood_amd.synthetic_voxels import forest_plot
    import bench
    dev = torch.device("cuda")
    for ins in (False, True):
        net = Net(num_classes=1, C=32, k=32, search_collect=not ins)
        net.load_state_dict(weights.synth_state_dict(1, 32, seed=0), strict=True)
        net = net.to(dev).eval()
        data = bench.make_batch(0, dev, 0)
        net(data)
        eng = net._engine
        acc = {}
        for _ in range(7):
            eng.events, eng.events_grouped = [], False
            net(data, keep={"geometry_only": True})
            torch.cuda.synchronize()
            ev, eng.events = eng.events, None
            per = {}
            for name, s, e in ev:
                per[name] = per.get(name, 0.0) + s.elapsed_time(e)
            for k_, v in per.items():
                acc.setdefault(k_, []).append(v)
        med = {k_: statistics.median(v) for k_, v in acc.items()}
        print(f"{libpath or 'in-tree'} {'insert ' if ins else 'collect'}: knn {med['knn']:.3f} ms  knn2 {med['knn2']:.3f}  ball {med['ball_query']:.3f}", flush=True)
        del net
    pc = forest_plot(plot_points, side=max(10.0, 100.0 * (plot_points / 10_000_000) ** 0.5)).to(dev)
    g = torch.Generator(device=dev).manual_seed(0)
    cls = torch.cat([pc[:, :3], pc[:, :3] + 1e-3 * torch.randn(pc.shape[0], 3, device=dev, generator=g)])[: int(1.9 * pc.shape[0])].contiguous()
    prob = torch.rand(cls.shape[0], device=dev, generator=g)
    for ins in (False, True, False, True):
        backproject.EXTRA_SEARCH_FLAGS = 0 if ins else _lib.SEARCH_COLLECT
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        lab, pw = backproject.collect_predictions(cls, (prob > 0.5).float(), prob, pc[:, :3].contiguous())
        torch.cuda.synchronize()
        print(f"{libpath or 'in-tree'} back-projection {'insert ' if ins else 'collect'}: {time.perf_counter() - t0:.3f} s  (checksum {float(pw.double().sum()):.6f})", flush=True)
    backproject.EXTRA_SEARCH_FLAGS = 0


for lp in libs:
    run(lp)
