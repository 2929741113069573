#!/usr/bin/env python3
"""predict.py - command line of the MI355X build, with the reference's flags (``pointstowood/predict.py:61-74``).

What runs on the GPU is the hot path this repository implements: per-voxel classification of pre-voxelised
clouds (``voxel_*.pt`` files as written by the reference's ``Voxelise.write_voxels``,
``pointstowood/src/preprocessing.py:79-127``: float32 ``[n, >=4]`` = x, y, z, reflectance, ...).  Use

    python predict.py --voxels DIR --model model.pth [--batch_size 8 --is-wood 0.5 --odir OUT]

``--point-cloud FILE.ply ...`` runs the reference's whole flow in memory on the GPU (``pointstowood/predict.py:116-156``):
PLY -> column handling -> height normalisation + voxeliser (2 m / 4 m grids) -> forward over every voxel ->
back-projection (k-nearest classified points, median probability, vote) -> ``<name>_ours.ply`` next to the input with
the input's columns + ``n_z, label, pwood``.  All flags carry the reference's names, types and defaults.
"""
from __future__ import annotations

import argparse
import os
import sys
import time

import numpy as np
import torch


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument('--point-cloud', '-p', default=[], nargs='+', type=str, help='list of point cloud files')
    p.add_argument('--odir', type=str, default='.', help='output directory')
    p.add_argument('--batch_size', default=8, type=int, help="voxels per forward with --reference-sampler (the default packs forwards by a point budget)")
    p.add_argument('--num_procs', default=-1, type=int, help="Number of CPU cores you want to use.")
    p.add_argument('--resolution', type=float, default=0.01, help='Resolution to which point cloud is downsampled [m]')
    p.add_argument('--grid_size', type=float, nargs='+', default=[2.0, 4.0], help='Grid sizes for voxelization')
    p.add_argument('--min_pts', type=int, default=128, help='Minimum number of points in voxel')
    p.add_argument('--max_pts', type=int, default=16384, help='Maximum number of points in voxel')
    p.add_argument('--model', type=str, default='model.pth', help='path to candidate model')
    p.add_argument('--is-wood', default=0.5, type=float, help='probability above which points are classified as wood')
    p.add_argument('--any-wood', default=1, type=float, help='a probability above which ANY point within KNN is classified as wood')
    p.add_argument('--output_fmt', default='ply', help="file type of output")
    p.add_argument('--verbose', action='store_true', help="print stuff")
    # extensions of this build
    p.add_argument('--voxels', type=str, default=None, help='directory of voxel_*.pt files (skips preprocessing)')
    p.add_argument('--precision', default='f16x3', choices=['f16x3', 'fp32', 'fp16', 'bf16'],
                   help='MFMA mode of the forward: f16x3 (default) and fp32 meet the 1e-4 probability bar of the fp32 CPU path; '
                        'fp16 / bf16 are the arithmetic of the reference autocast GPU path (about 1.7x faster, 1e-3..1e-2 off)')
    p.add_argument('--reference-sampler', action='store_true',
                   help='reproduce the reference BalancedBatchSampler exactly (drops the remainder, unseeded shuffles)')
    return p


def segment_file(path, args):
    """One point cloud file through the whole flow; returns the output path."""
    if not torch.cuda.is_available():
        raise SystemExit('predict.py needs an MI355X: the HIP path has no CPU fallback')
    from pointstowood_amd import Net
    from pointstowood_amd import io as pio
    from pointstowood_amd.pipeline import segment_plot
    from pointstowood_amd.predicter import load_model
    if os.path.splitext(path)[1].lower() != '.ply':
        raise SystemExit(f'{path}: only .ply input is built (the reference also reads .las / .pcd through laspy / its own parser)')
    # one process per GPU under `python -m torch.distributed.run --nproc-per-node N predict.py ...`: every rank reads the
    # file, the voxel batches and the back-projection are sharded (pipeline.segment_plot), rank 0 writes the result
    world, rank, local = (int(os.environ.get(k, d)) for k, d in (("WORLD_SIZE", "1"), ("RANK", "0"), ("LOCAL_RANK", "0")))
    dist = None
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local)
        if not dist.is_initialized():
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    device = torch.device('cuda', local)
    t0 = time.time()
    cols, headers, had_refl = pio.prepare_columns(pio.read_ply(path))
    if rank == 0:
        print('Reflectance detected' if had_refl else 'No reflectance detected, column added with zeros.')
    names = list(cols)
    xyz64 = np.stack([np.asarray(cols[c], dtype=np.float64) for c in names[:3]], 1)
    origin = xyz64.min(0)                                   # plot-local coordinates: fp32 keeps millimetres at any easting
    arr = np.concatenate([xyz64 - origin] + [np.asarray(cols[c], dtype=np.float64)[:, None] for c in names[3:]], 1)
    pc = torch.from_numpy(arr.astype(np.float32)).to(device)
    model = Net(num_classes=1, precision=args.precision).to(device)
    try:
        load_model(args.model, model, device)
    except KeyError:
        raise Exception(f'No model loaded at {args.model}')
    model.eval()
    if rank == 0:
        print(f'Voxelising to {args.grid_size} grid sizes')
    stats = {}
    gen = torch.Generator(device=device).manual_seed(0) if world > 1 else None   # identical max_pts sampling on every rank
    # input that already carries n_z (e.g. an *_ours.ply fed back in): the reference skips the height normalisation and
    # takes the LAST column for n_z (preprocessing.py:81-86,127)
    n_z, label, pwood = segment_plot(pc, model, args.grid_size, args.min_pts, args.max_pts, args.is_wood, args.any_wood,
                                     stats=stats, generator=gen, dist=dist, ground='n_z' not in names)
    opath = os.path.join(os.path.dirname(path), os.path.splitext(os.path.basename(path))[0] + '_ours.ply')
    if rank != 0:
        return opath
    out = {c: cols[c] for c in names[:3]}
    for h in dict.fromkeys(headers):                       # predicter.py:233: headers + n_z, label, pwood, de-duplicated
        out[h] = cols[h]
    out['n_z'], out['label'], out['pwood'] = n_z.cpu().numpy(), label.cpu().numpy(), pwood.cpu().numpy()
    pio.write_ply(opath, out)
    if args.verbose:
        print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in stats.items()})
        print(f'{len(arr)} points -> {opath} in {time.time() - t0:.2f} s')
    return opath


def main(argv=None):
    args = build_parser().parse_args(argv)
    torch.set_num_threads(os.cpu_count() if args.num_procs == -1 else args.num_procs)
    if args.verbose:
        print('\n---- parameters used ----')
        for k, v in vars(args).items():
            print('{:<35}{}'.format(k, v))
    if args.voxels is None:
        if not args.point_cloud:
            raise SystemExit('no input specified, please specify --voxels DIR (or --point-cloud, see below)')
        for f in args.point_cloud:
            if not os.path.isfile(f):
                raise FileNotFoundError(f'Point cloud file not found: {f}')
        return [segment_file(f, args) for f in args.point_cloud]
    if not torch.cuda.is_available():
        raise SystemExit('predict.py needs an MI355X: the HIP path has no CPU fallback')

    from pointstowood_amd import DataLoader, Net
    from pointstowood_amd.predicter import VoxelDataset, classify_sharded, classify_voxels, load_model, plan_batches
    # one process per GPU under `python -m torch.distributed.run --nproc-per-node N predict.py --voxels ...`: the voxel
    # batches are sharded over the ranks (predicter.classify_sharded), rank 0 writes the result
    world, rank, local = (int(os.environ.get(k, d)) for k, d in (("WORLD_SIZE", "1"), ("RANK", "0"), ("LOCAL_RANK", "0")))
    dist = None
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local)
        if not dist.is_initialized():
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    device = torch.device('cuda', local)
    model = Net(num_classes=1, precision=args.precision).to(device)
    try:
        load_model(args.model, model, device)
    except KeyError:
        raise Exception(f'No model loaded at {args.model}')
    model.eval()
    ds = VoxelDataset(args.voxels)
    if len(ds) == 0:
        raise SystemExit(f'no voxel_*.pt files in {args.voxels}')
    t0 = time.time()
    if world > 1:
        # the SAME plan as one process (predicter.plan_batches: point-budget forwards, or the reference's sampler), dealt to the ranks
        if args.reference_sampler:
            np.random.seed(0)      # the reference's sampler draws from the global numpy RNG: every rank needs the same batches
        out = classify_sharded(model, ds, plan_batches(ds, args.batch_size, args.reference_sampler), args.is_wood, device, dist)
    else:
        # the reference's loop (predicter.py:193-215) through the stream pipeline: voxels read once, forwards packed by a point
        # budget (--reference-sampler: its BalancedBatchSampler at --batch_size voxels per forward), one D2H copy at the end
        out = classify_voxels(model, ds, args.is_wood, device, batch_size=args.batch_size, reference_sampler=args.reference_sampler)
    n = out.shape[0]
    if rank != 0:
        return out
    os.makedirs(args.odir, exist_ok=True)
    path = os.path.join(args.odir, 'classified_voxels.npy')
    np.save(path, out)
    if args.verbose:
        dt = time.time() - t0
        print(f'classified {n} points of {len(ds)} voxels in {dt:.2f} s ({n / dt:.0f} points/s incl. disk) -> {path}')
    return out


if __name__ == '__main__':
    main()
