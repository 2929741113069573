"""Worker of tests/test_gpu_multi.py: one process per GPU under ``python -m torch.distributed.run``.

Runs the two sharded entry points of the product over the real collective backend (``nccl`` = RCCL on ROCm) and compares
them, bit for bit, with the single-process result computed on rank 0's GPU:

* ``pipeline.segment_plot(dist=...)``  - LPT-sharded voxel batches, all-gather of the probabilities, back-projection owned
                                          by x-slabs of the plot, all-gather of labels / pwood;
* ``predicter.classify_sharded``        - the sharded form of the reference's inference loop (predicter.py:193-213).

``--dry`` (CPU suite): gloo backend, a stand-in model and vote on the CPU - exercises this script's own logic where no
second GPU exists.  Exit code 0 = every rank agrees with the single-process result.
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


class _FakeNet:
    """Net's interface on the CPU (``--dry``): logits = a fixed function of the batch."""
    def stream(self, batches):
        for d in batches:
            yield self(d)

    def __call__(self, d):
        return (d.pos.sum(dim=1) + 0.01 * d.local_shift.view(-1, 3)[d.batch.long()].sum(dim=1)) * 0.7 - 0.2


def _fake_collect(cls_xyz, cls_pred, cls_prob, query_xyz, any_wood=1.0, k=1):
    if cls_xyz.shape[0] == 0 or query_xyz.shape[0] == 0:
        z = torch.zeros(query_xyz.shape[0])
        return z, z.clone()
    j = torch.cat([torch.cdist(q.to(torch.float64), cls_xyz.to(torch.float64)).argmin(dim=1) for q in query_xyz.split(4096)])
    return cls_pred[j].clone(), cls_prob[j].clone()


def _fake_collect_checked(cls_xyz, cls_pred, cls_prob, query_xyz, any_wood=1.0, k=1):
    """... and the distance to that nearest point (the spatially sharded flow's exactness test at k = 1)."""
    nq = query_xyz.shape[0]
    if cls_xyz.shape[0] == 0 or nq == 0:
        z = torch.zeros(nq)
        return z, z.clone(), torch.full((nq,), float("inf"), dtype=torch.float64)
    c64 = cls_xyz.to(torch.float64)
    parts = [torch.cdist(q.to(torch.float64), c64).min(dim=1) for q in query_xyz.split(4096)]     # (chunk by chunk: never the full matrix)
    dk, j = torch.cat([p.values for p in parts]), torch.cat([p.indices for p in parts])
    return cls_pred[j].clone(), cls_prob[j].clone(), dk


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dry", action="store_true")
    ap.add_argument("--points", type=int, default=300_000)
    args = ap.parse_args()
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
    from pointstowood_amd import pipeline, predicter
    from pointstowood_amd import synthetic_voxels as synth
    if args.dry:
        dist.init_process_group("gloo")
        dev = torch.device("cpu")
        torch.set_num_threads(1)
        net = _FakeNet()
        pipeline.collect_predictions = _fake_collect
        pipeline.collect_predictions_checked = _fake_collect_checked
        from oracle import preprocess as OP                    # the CPU stand-in of the voxeliser's HIP grid step
        from pointstowood_amd import preprocessing
        preprocessing.backend = OP.TensorBackend
        budget = dict(max_points=20000, max_voxels=16)
    else:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
        dist.init_process_group("nccl", device_id=dev)
        from pointstowood_amd import Net
        from pointstowood_amd import synthetic_weights as weights
        net = Net(num_classes=1, C=32, k=32)
        net.load_state_dict(weights.synth_state_dict(1, 32, seed=0), strict=True)
        net = net.to(dev).eval()
        budget = dict(max_points=65536, max_voxels=64)        # several forwards per rank
    gen = lambda: torch.Generator(device=dev).manual_seed(0)
    side = max(10.0, 100.0 * (args.points / 10_000_000) ** 0.5)
    pc = synth.forest_plot(args.points, seed=2, side=side).to(dev)
    ok = True

    # ---- segment_plot: sharded vs single process ---------------------------------------------------------------
    stats = {}
    n_z, label, pwood = pipeline.segment_plot(pc, net, (2.0, 4.0), min_pts=128, max_pts=16384, generator=gen(), dist=dist,
                                              stats=stats, **budget)
    n_z1, label1, pwood1 = pipeline.segment_plot(pc, net, (2.0, 4.0), min_pts=128, max_pts=16384, generator=gen(), **budget)
    same_plot = torch.equal(n_z, n_z1) and torch.equal(label, label1) and torch.equal(pwood, pwood1)
    ok &= same_plot

    # ---- classify_sharded vs the single-process loop ---------------------------------------------------------------
    from pointstowood_amd.preprocessing import voxelise
    vox, _ = voxelise(pc, (2.0,), 128, 16384, generator=gen())
    vox = [v.cpu() for v in vox[: 6 * world + 3]]
    ds = predicter.VoxelDataset(vox)
    ds.lengths = [int(v.shape[0]) for v in vox]
    batches = [list(range(i, min(i + 2, len(vox)))) for i in range(0, len(vox), 2)]
    got = predicter.classify_sharded(net, ds, batches, 0.5, dev, dist)
    from pointstowood_amd.data import Batch
    import numpy as np
    ref = np.vstack([predicter.classify_batch(net, Batch.from_data_list([ds[i] for i in b]), 0.5, dev) for b in batches])
    same_cls = got.shape == ref.shape and bool((got == ref).all())
    ok &= same_cls

    flag = torch.tensor([1 if ok else 0], device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if rank == 0:
        print(f"multi_gpu_worker world={world} backend={dist.get_backend()} plot_equal={same_plot} classify_equal={same_cls} "
              f"voxels={stats.get('voxels')} rank0_forwards={len(stats.get('batch_points', []))} all_ranks_ok={int(flag)}", flush=True)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if int(flag) == 1 else 1)


if __name__ == "__main__":
    main()
