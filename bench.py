#!/usr/bin/env python3
"""Throughput benchmark of the PointsToWood inference hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

One "step" = one pass of the hot path (``Net.forward``: geometry + features, reference
``pointstowood/src/model.py:226-245``) over one voxel batch of BASELINE.json ``configs[1]``:
batch_size 8 x 16384-point 2 m voxels, k=32, xyz only (reflectance = 0), synthetic uniform
points (seeds 123..130 + 1000*rank), recipe-generated weights of the reference architecture
(C=32, 18.16 M parameters).  Inputs are resident in HBM before the timed region.  For N > 1
(launched by ``python -m torch.distributed.run``) every rank runs its own batch (weak scaling,
voxel batches are independent) and each step ends with the path's only collective: an RCCL
all-gather of the per-point logits.  Rank 0 prints ONE JSON line.

``roofline`` is measured live: after the timed region one extra step is run with HIP events
around every run of consecutive launches of one kernel class (on the launch stream) and the dominant
kernel's algorithmic FLOPs are divided by its measured time.  ``cpu_baseline`` times the CPU oracle (``oracle/net.py``, a port
of the reference forward pinned to the reference's own outputs) on one 16384-point voxel.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# MI355X_MICROARCH.md dense MFMA peaks: v_mfma_f32_32x32x2_f32 and v_mfma_f32_32x32x16_f16
PEAK_TFLOPS = {"fp32": 157.3, "f16x3": 2500.0}
C, K_NBR, BATCH, NPTS = 32, 32, 8, 16384


def make_batch(rank: int, device):
    from pointstowood_amd import synthetic_voxels as synth
    vox = [synth.uniform_voxel(2.0, NPTS, 123 + i + 1000 * rank, False) for i in range(BATCH)]
    b = synth.collate(vox)

    class D:
        pass
    d = D()
    d.pos, d.batch = b["pos"].to(device), b["batch"].to(device)
    d.reflectance, d.sf, d.ptr = b["reflectance"].to(device), b["sf"].to(device), b["ptr"].to(device)
    return d


def algorithmic_macs(geo):
    """SURVEY.md 8(d) / BASELINE.md 3: MACs of the reference forward for the level sizes actually produced."""
    N, (M1, M2, M3) = geo.N, [geo.levels[l].n for l in (1, 2, 3)]
    E = [int(geo.levels[l].deg[: geo.levels[l].n].sum()) for l in (1, 2, 3)]
    total = (803424 * N + 1245184 * M1 + 3440640 * M2 + 12191232 * M3 + 10496 * E[0] + 74496 * E[1] + 296448 * E[2])
    # the layers the GEMM kernel executes (everything except the per-edge MLPs and the 3-wide stem) ...
    gemm = ((540672 + 262656) * N + (655360 + 589824) * M1 + (2621440 + 819200) * M2
            + (10485760 + 525824 + 1179648) * M3)
    # ... plus the hoisted layer-1 products it runs once per SOURCE point (<= the per-edge MACs they replace)
    gemm += N * 32 * 64 + M1 * 128 * 192 + M2 * 256 * 384
    # fused PointNetConv kernel: layer-2 MACs of the real edges (padded slots are not counted)
    sa = 64 * 128 * E[0] + 192 * 256 * E[1] + 384 * 512 * E[2]
    return total, {"gemm_kernel": gemm, "sa_conv_kernel": sa}, dict(N=N, M=[M1, M2, M3], E=E)


KERNEL_OF = {"gemm_hoist": "gemm_kernel", "gemm_res": "gemm_kernel", "gemm_mlp": "gemm_kernel", "sa_conv": "sa_conv_kernel"}


def profile_step(net, data):
    eng = net._engine
    eng.events, eng.events_grouped = [], True   # one HIP-event pair per run of consecutive launches of one kernel class
    keep = {}
    streams, eng.res_streams = eng.res_streams, 1   # per-kernel durations: no two kernels in flight while they are timed
    net(data, keep=keep)
    eng.flush_events()
    torch.cuda.synchronize()
    ev, eng.events, eng.events_grouped, eng.res_streams = eng.events, None, False, streams
    per = {}
    for name, s, e, launches in ev:
        kname = KERNEL_OF.get(name, name)
        t, n = per.get(kname, (0.0, 0))
        per[kname] = (t + s.elapsed_time(e), n + launches)
    return per, keep["geometry"]


def cpu_baseline():
    from oracle import net as onet
    from pointstowood_amd import synthetic_voxels as synth, synthetic_weights as weights
    torch.set_num_threads(min(os.cpu_count() or 1, 16))  # more threads only add contention on this path
    sd = weights.synth_state_dict(1, C, seed=0)
    v = synth.collate([synth.uniform_voxel(2.0, NPTS, 123, False)])
    run = lambda: onet.forward(sd, v["pos"], v["batch"], v["reflectance"], v["sf"], k=K_NBR)
    run()  # warm-up
    reps, t0 = 2, time.perf_counter()
    for _ in range(reps):
        run()
    dt = (time.perf_counter() - t0) / reps
    return {"value": NPTS / dt, "unit": "points/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"1 voxel x {NPTS} pts (U2-16k seed 123), k={K_NBR}, C={C}, fp32, {reps} timed passes after 1 warm-up, "
                      f"{dt:.2f} s per pass"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--pipeline", type=int, default=1,
                    help="1: two-stream pipeline over the step sequence (default); 0: strictly sequential forwards")
    ap.add_argument("--precision", default="f16x3", choices=["f16x3", "fp32"],
                    help="f16x3: split-fp16 MFMA (3 fp16 MFMAs per product, fp32 accumulate); fp32: fp32 MFMA")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if args.gpus != 1:
            raise SystemExit(f"--gpus {args.gpus} needs `python -m torch.distributed.run --nproc-per-node {args.gpus}` "
                             f"(WORLD_SIZE={world})")
        world = 1
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=device)

    from pointstowood_amd import synthetic_weights as weights
    from pointstowood_amd import Net
    from pointstowood_amd.dist import gather_logits
    net = Net(num_classes=1, C=C, k=K_NBR, precision=args.precision)
    net.load_state_dict(weights.synth_state_dict(1, C, seed=0), strict=True)
    net = net.to(device).eval()
    data = make_batch(rank, device)
    peak = PEAK_TFLOPS[args.precision]

    def run(n):
        """n steps = n full forwards (geometry + features) of one voxel batch each.  With --pipeline the engine's
        two-stream software pipeline overlaps the geometry phase of step i+1 with the feature phase of step i."""
        out = None
        if args.pipeline:
            for logits in net.stream(data for _ in range(n)):
                out = gather_logits(logits, dist) if world > 1 else logits
        else:
            for _ in range(n):
                logits = net(data)
                out = gather_logits(logits, dist) if world > 1 else logits
        return out

    run(args.warmup)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = run(args.steps)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    assert bool(torch.isfinite(out).all()) or os.environ.get("P2W_SA_DBG") or os.environ.get("P2W_GEMM_DBG")

    if rank == 0:
        per, geo = profile_step(net, data)
        total_macs, kmacs, sizes = algorithmic_macs(geo)
        dom = max(kmacs, key=lambda kname: per[kname][0])
        dom_ms, dom_launches = per[dom]
        achieved = 2.0 * kmacs[dom] / (dom_ms * 1e-3) / 1e12
        pts = world * args.steps * BATCH * NPTS
        kname = {"gemm_kernel": ("gemm_h2g_kernel (256x256 and 128x128 tile instantiations, all launches)"
                                 if args.precision == "f16x3" else "gemm_kernel"),
                 "sa_conv_kernel": ("sa_conv16p_kernel (+ sa_edge_meta_kernel)" if args.precision == "f16x3" else "sa_conv_kernel")}
        line = {
            "metric": "classified points/sec", "value": pts / dt, "unit": "points/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: batch_size 8 x 16384-pt 2 m voxels, k=32, xyz-only, 1 batch per GPU per step",
                       "global_batch_voxels": world * BATCH, "points_per_step": world * BATCH * NPTS, "C": C,
                       "level_sizes": sizes, "parallelism": f"voxel-batch sharding x{world}, RCCL all-gather of logits",
                       "pipeline": "2 HIP streams: geometry(i+1) || features(i) (features high priority)" if args.pipeline else "sequential"},
            "end_to_end_tflops_algorithmic": 2.0 * total_macs * args.steps / dt / 1e12,
            "roofline": {"bound": "mfma", "kernel": kname.get(dom, dom), "achieved": achieved, "peak": peak,
                         "unit": "TFLOP/s", "frac": achieved / peak, "traffic": None,
                         "note": ("f16x3 issues 3 fp16 MFMAs per algorithmic product: ceiling for algorithmic FLOPs is peak/3"
                                  if args.precision == "f16x3" else "exact fp32 MFMA"),
                         "launches_per_step": dom_launches, "kernel_ms_per_step": dom_ms,
                         "algorithmic_gflop_per_step": 2.0 * kmacs[dom] / 1e9},
            "kernel_ms_per_step": {kname: round(v[0], 4) for kname, v in sorted(per.items(), key=lambda kv: -kv[1][0])},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
