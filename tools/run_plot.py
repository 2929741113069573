#!/usr/bin/env python3
"""Plot-level run (BASELINE configs[3] shape): synthetic forest plot -> GPU voxeliser (2 m + 4 m grids, min/max points)
-> length-balanced voxel batches -> pipelined classification, sharded over the ranks of torch.distributed when launched
with `python -m torch.distributed.run --nproc-per-node N tools/run_plot.py ...` (one all-gather of results at the end).

    python tools/run_plot.py [--points 2000000] [--batch_size 8]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointstowood_amd import synthetic_weights as weights  # noqa: E402
from pointstowood_amd.synthetic_voxels import forest_plot as synth_plot  # noqa: E402
from pointstowood_amd import Batch, Net  # noqa: E402
from pointstowood_amd.dist import partition_batches  # noqa: E402
from pointstowood_amd.predicter import BalancedBatchSampler, PointBudgetSampler, collate_device  # noqa: E402
from pointstowood_amd.preprocessing import voxelise  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=2_000_000)
    ap.add_argument("--batch_size", type=int, default=0,
                    help="voxels per forward with the reference-style sampler; 0 (default) = point-budget batching")
    ap.add_argument("--max_points", type=int, default=524288, help="point budget per forward (point-budget batching)")
    ap.add_argument("--max_voxels", type=int, default=512, help="voxel cap per forward (point-budget batching)")
    ap.add_argument("--min_pts", type=int, default=128)
    ap.add_argument("--max_pts", type=int, default=16384)
    args = ap.parse_args()
    world, rank, local = (int(os.environ.get(k, d)) for k, d in (("WORLD_SIZE", "1"), ("RANK", "0"), ("LOCAL_RANK", "0")))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)
    net = Net(1, C=32, k=32)
    net.load_state_dict(weights.synth_state_dict(1, 32, seed=0))
    net = net.to(dev).eval()
    side = 100.0 * (args.points / 10_000_000) ** 0.5
    pc = synth_plot(args.points, side=max(side, 10.0)).to(dev)
    from pointstowood_amd.pipeline import segment_plot
    gen = lambda: torch.Generator(device=dev).manual_seed(0)
    if args.batch_size > 0:   # reference-style fixed voxel count per forward: classification only (host-bound; for comparison)
        from pointstowood_amd.predicter import VoxelDataset
        vox, _ = voxelise(pc, (2.0, 4.0), args.min_pts, args.max_pts, generator=gen())
        batches = list(BalancedBatchSampler(VoxelDataset(vox), args.batch_size))
        torch.cuda.synchronize()
        t0, n_pts = time.perf_counter(), 0
        for logits in net.stream(collate_device([vox[i] for i in b]) for b in batches):
            n_pts += logits.numel()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"reference-style batches of {args.batch_size}: {n_pts} points in {dt:.2f} s = {n_pts / dt / 1e6:.2f} M points/s")
        return
    segment_plot(pc[:200000], net, min_pts=args.min_pts, max_pts=args.max_pts, generator=gen(), dist=dist)   # warm-up
    torch.cuda.synchronize()
    stats = {}
    t0 = time.perf_counter()
    n_z, label, pwood = segment_plot(pc, net, (2.0, 4.0), args.min_pts, args.max_pts, max_points=args.max_points,
                                     generator=gen(), stats=stats, dist=dist, max_voxels=args.max_voxels)
    torch.cuda.synchronize()
    total = time.perf_counter() - t0
    if rank == 0:
        print(f"plot {args.points} pts -> {stats['voxels']} voxels, {stats['classified_points']} classified points incl. "
              f"2 m / 4 m overlap on {world} GPU(s): voxelise {stats['voxelise_s']:.2f} s, classify {stats['classify_s']:.2f} s "
              f"({stats['classified_points'] / stats['classify_s'] / 1e6:.2f} M classified points/s), back-project "
              f"{stats['backproject_s']:.2f} s ({args.points / stats['backproject_s'] / 1e6:.2f} M points/s, k=64); end to end "
              f"{total:.2f} s = {args.points / total / 1e6:.2f} M plot points/s; wood fraction {float(label.mean()):.3f}", flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
