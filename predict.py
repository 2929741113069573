#!/usr/bin/env python3
"""predict.py - command line of the MI355X build, with the reference's flags (``pointstowood/predict.py:61-74``).

What runs on the GPU is the hot path this repository implements: per-voxel classification of pre-voxelised
clouds (``voxel_*.pt`` files as written by the reference's ``Voxelise.write_voxels``,
``pointstowood/src/preprocessing.py:79-127``: float32 ``[n, >=4]`` = x, y, z, reflectance, ...).  Use

    python predict.py --voxels DIR --model model.pth [--batch_size 8 --is-wood 0.5 --odir OUT]

The reference's ``--point-cloud`` entry (file I/O -> height normalisation -> voxeliser -> forward -> KD-tree
back-projection and vote -> PLY) needs the components SURVEY.md section 8f lists as "next" (voxeliser,
back-projection, PLY I/O); they are not built yet, so ``--point-cloud`` stops with an explanatory error instead of
silently doing something else.  All other flags are accepted with the reference's names, types and defaults.
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
    p.add_argument('--batch_size', default=8, type=int, help="voxels per forward")
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
    p.add_argument('--precision', default='f16x3', choices=['f16x3', 'fp32'], help='MFMA mode of the forward')
    p.add_argument('--reference-sampler', action='store_true',
                   help='reproduce the reference BalancedBatchSampler exactly (drops the remainder, unseeded shuffles)')
    return p


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
        raise SystemExit("--point-cloud needs the voxeliser / back-projection / PLY I/O rows (SURVEY.md 8f), which this "
                         "build does not contain yet. Voxelise with the reference's preprocessing and pass --voxels DIR.")
    if not torch.cuda.is_available():
        raise SystemExit('predict.py needs an MI355X: the HIP path has no CPU fallback')

    from pointstowood_amd import DataLoader, Net
    from pointstowood_amd.predicter import BalancedBatchSampler, VoxelDataset, classify_batch, load_model
    device = torch.device('cuda')
    model = Net(num_classes=1, precision=args.precision).to(device)
    try:
        load_model(args.model, model, device)
    except KeyError:
        raise Exception(f'No model loaded at {args.model}')
    model.eval()
    ds = VoxelDataset(args.voxels)
    if len(ds) == 0:
        raise SystemExit(f'no voxel_*.pt files in {args.voxels}')
    sampler = BalancedBatchSampler(ds, args.batch_size, reference=args.reference_sampler)
    loader = DataLoader(ds, batch_sampler=sampler, num_workers=0, pin_memory=True)
    t0, outs, n = time.time(), [], 0
    for data in loader:
        outs.append(classify_batch(model, data, args.is_wood, device))
        n += outs[-1].shape[0]
    out = np.vstack(outs)
    os.makedirs(args.odir, exist_ok=True)
    path = os.path.join(args.odir, 'classified_voxels.npy')
    np.save(path, out)
    if args.verbose:
        dt = time.time() - t0
        print(f'classified {n} points of {len(ds)} voxels in {dt:.2f} s ({n / dt:.0f} points/s incl. disk) -> {path}')
    return out


if __name__ == '__main__':
    main()
