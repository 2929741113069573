#!/usr/bin/env python3
"""Throughput benchmark of the PointsToWood inference hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--precision f16x3|fp16|bf16|fp32]

One "step" = one pass of the hot path (``Net.forward``: geometry + features, reference
``pointstowood/src/model.py:226-245``) over one voxel batch of BASELINE.json ``configs[1]``:
batch_size 8 x 16384-point 2 m voxels, k=32, xyz only (reflectance = 0), synthetic uniform
points, recipe-generated weights of the reference architecture (C=32, 18.16 M parameters).
The steps rotate over ``--batches`` (default 8) DISTINCT seeded batches (voxel seeds
123 + 8 j + i + 1000 rank), all resident in HBM before the timed region - ``value`` /
``ms_per_step`` are measured that way, as the contract requires.  A second timed region feeds
the same batches from pinned host memory and copies the logits back (H2D + D2H inside the
region): reported as ``pcie_inclusive`` (SURVEY.md 8d's wording of the metric), never as ``value``.

``--gpus N`` (N > 1) from a plain shell spawns ``python -m torch.distributed.run`` with N ranks
as a CHILD process (the parent never touches the GPU) and exits with its code; when the driver
has already launched the ranks (WORLD_SIZE == N) it just runs.  Every rank runs its own batches
(weak scaling, voxel batches are independent) and each step ends with the path's only
collective: an RCCL all-gather of the per-point logits.  Rank 0 prints ONE JSON line.

``roofline`` is measured live: after the timed regions one extra, sequential step is run with HIP
events around every run of consecutive launches of one kernel class (on the launch stream) and the
dominant kernel's algorithmic FLOPs are divided by its measured time; ``traffic`` comes from the
committed rocprofv3 PMC summary of the same command (``profiles/r2_<precision>_hbm_traffic.json``).
``hbm_kernels`` gives the memory-bound kernels' algorithmic bytes / measured time
against the 8 TB/s HBM peak.  ``cpu_baseline`` times the CPU oracle (``oracle/net.py``, a port of
the reference forward pinned to the reference's own outputs) on a bounded sample of batch 0.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# MI355X_MICROARCH.md dense MFMA peaks: v_mfma_f32_32x32x2_f32 and v_mfma_f32_32x32x16_{f16,bf16}; HBM3E spec peak
PEAK_TFLOPS = {"fp32": 157.3, "f16x3": 2500.0, "fp16": 2500.0, "bf16": 2500.0}
PEAK_HBM_GBPS = 8000.0
C, K_NBR, BATCH, NPTS = 32, 32, 8, 16384
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "r2_{precision}_hbm_traffic.json")   # written by tools/profile_r2.sh


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=48)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batches", type=int, default=8, help="distinct seeded voxel batches the steps rotate over")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pcie", action="store_true", help="skip the second (H2D/D2H-inclusive) timed region")
    ap.add_argument("--pipeline", type=int, default=1,
                    help="1: two-stream pipeline over the step sequence (default); 0: strictly sequential forwards")
    ap.add_argument("--precision", default="f16x3", choices=["f16x3", "fp16", "bf16", "fp32"],
                    help="f16x3: split-fp16 MFMA (3 MFMAs per product, fp32-class accuracy; the parity default); fp16 / bf16: one "
                         "MFMA per product (the reference's autocast arithmetic; outside the 1e-4 bar); fp32: fp32 MFMA")
    return ap.parse_args(argv)


def self_launch(args) -> int:
    """`python bench.py --gpus N` without a launcher: run the N ranks as children of this (GPU-free) process."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def host_batch(rank: int, j: int):
    """Batch j of this rank on the host: BATCH uniform 2 m voxels, collated like PyG's Batch (predicter.py:177)."""
    from pointstowood_amd import synthetic_voxels as synth
    vox = [synth.uniform_voxel(2.0, NPTS, 123 + BATCH * j + i + 1000 * rank, False) for i in range(BATCH)]
    return synth.collate(vox)


class Feed:
    """Duck-typed batch (pos, batch, reflectance, sf, ptr) - what ``model(data)`` reads (predicter.py:198)."""
    FIELDS = ("pos", "batch", "reflectance", "sf", "ptr")

    def __init__(self, **kw):
        for k, v in kw.items():
            setattr(self, k, v)

    @classmethod
    def to_device(cls, b, device, non_blocking=False):
        return cls(**{k: b[k].to(device, non_blocking=non_blocking) for k in cls.FIELDS})


def make_batch(rank: int, device, j: int = 0):
    """Batch j of `rank`, resident on `device` (used by the tools/ scripts)."""
    return Feed.to_device(host_batch(rank, j), device)


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


def algorithmic_bytes(geo):
    """SURVEY.md 8(d) algorithmic bytes of the memory-bound kernels (fp32 features, int32 indices, each datum once)."""
    N, B = geo.N, geo.B
    M = [N] + [geo.levels[l].n for l in (1, 2, 3)]
    Fc = [16 * C, 16 * C, 16 * C, 16 * C]   # width of the interpolated (coarse) features entering fp4, fp3, fp2, fp1
    out = {
        "voxel_sample": sum(16 * M[l] + 4 * M[l + 1] for l in range(3)),
        "level_gather": sum((4 + 16 + 16) * M[l + 1] for l in range(3)),
        "segment_max": 4 * 512 * (M[3] + B),
    }
    # kNN-interp2 per FP level: 4 F Mx + 12 (Mx + My) + 4 F My   (x = coarse, y = fine)
    coarse = [B, M[3], M[2], M[1]]
    fine = [M[3], M[2], M[1], M[0]]
    out["interp_concat"] = sum(4 * Fc[i] * coarse[i] + 12 * (coarse[i] + fine[i]) + 4 * Fc[i] * fine[i] for i in range(4))
    return out


KERNEL_OF = {"gemm_hoist": "gemm_kernel", "gemm_res": "gemm_kernel", "gemm_mlp": "gemm_kernel", "sa_conv": "sa_conv_kernel"}


def profile_step(net, data):
    import torch
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


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(batch0):
    """CPU oracle on the FIRST voxel of batch 0 of this benchmark (same batch definition, bounded sample)."""
    import torch
    from oracle import net as onet
    from pointstowood_amd import synthetic_weights as weights
    threads = min(os.cpu_count() or 1, 16)   # more threads only add contention on this path
    torch.set_num_threads(threads)
    sd = weights.synth_state_dict(1, C, seed=0)
    n = int(batch0["ptr"][1])
    pos, refl = batch0["pos"][:n].clone(), batch0["reflectance"][:n].clone()
    bidx, sf = torch.zeros(n, dtype=torch.long), batch0["sf"][:1].clone()
    run = lambda: onet.forward(sd, pos, bidx, refl, sf, k=K_NBR)
    run()  # warm-up
    reps, t0 = 2, time.perf_counter()
    for _ in range(reps):
        run()
    dt = (time.perf_counter() - t0) / reps
    return {"value": n / dt, "unit": "points/s", "cores": threads, "kind": "port", "cpu_model": cpu_model(),
            "host_cores": os.cpu_count(),
            "sample": f"voxel 0 of batch 0 ({n} pts, U2-16k seed 123), k={K_NBR}, C={C}, fp32, {reps} timed passes after 1 "
                      f"warm-up, {dt:.2f} s per pass, {threads} torch threads"}


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        if "RANK" in os.environ:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
        raise SystemExit(self_launch(args))     # no torch / HIP call has happened in this process
    if args.gpus == 1:
        world = 1
    rank = int(os.environ.get("RANK", "0")) if world > 1 else 0
    local = int(os.environ.get("LOCAL_RANK", "0")) if world > 1 else 0

    import torch
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
    nb = max(1, args.batches)
    host = [host_batch(rank, j) for j in range(nb)]
    resident = [Feed.to_device(b, device) for b in host]
    pinned = [{k: b[k].pin_memory() for k in Feed.FIELDS} for b in host]
    out_host = [torch.empty(BATCH * NPTS, dtype=torch.float32).pin_memory() for _ in range(nb)]
    peak = PEAK_TFLOPS[args.precision]

    def run(n, pcie=False, stamps=None):
        """n steps = n full forwards (geometry + features) of one voxel batch each, rotating over the distinct batches.
        With --pipeline the engine's two-stream software pipeline overlaps the geometry phase of step i+1 with the
        feature phase of step i.  pcie: inputs come from pinned host memory, logits go back to it, inside the region."""
        out = None

        def feed():
            for i in range(n):
                yield Feed.to_device(pinned[i % nb], device, non_blocking=True) if pcie else resident[i % nb]

        def finish(i, logits):
            o = gather_logits(logits, dist) if world > 1 else logits
            if pcie:
                out_host[i % nb].copy_(logits, non_blocking=True)
            if stamps is not None:
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                stamps.append(e)
            return o
        if args.pipeline:
            for i, logits in enumerate(net.stream(feed())):
                out = finish(i, logits)
        else:
            for i, d in enumerate(feed()):
                out = finish(i, net(d))
        return out

    def timed(n, pcie):
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        stamps = []
        s0 = torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        s0.record()
        out = run(n, pcie, stamps)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t)
        marks = [s0] + stamps
        per_step = [marks[i].elapsed_time(marks[i + 1]) for i in range(n)]   # completion-to-completion, ms
        return dt, per_step, out

    # setup, not warmup: one untimed forward per distinct batch so that the caching allocator has seen every workspace size
    # (level sizes are data-dependent; a first-time hipMalloc inside the timed region costs milliseconds); then W warmup steps
    for d in resident:
        net(d)
    run(args.warmup)
    dt, per_step, out = timed(args.steps, False)
    assert bool(torch.isfinite(out).all())
    pcie = None
    if not args.no_pcie:
        run(min(2, args.warmup), pcie=True)
        dt_p, per_p, _ = timed(args.steps, True)
        pcie = {"value": world * args.steps * BATCH * NPTS / dt_p, "unit": "points/s", "ms_per_step": dt_p / args.steps * 1e3,
                "ms_per_step_median": statistics.median(per_p),
                "what": "same steps with the inputs (pos, batch, reflectance, sf, ptr: 2.7 MB) copied from pinned host memory and "
                        "the logits (0.5 MB) copied back inside the timed region"}

    if rank == 0:
        per, geo = profile_step(net, resident[0])
        total_macs, kmacs, sizes = algorithmic_macs(geo)
        dom = max(kmacs, key=lambda kname: per[kname][0])
        dom_ms, dom_launches = per[dom]
        achieved = 2.0 * kmacs[dom] / (dom_ms * 1e-3) / 1e12
        pts = world * args.steps * BATCH * NPTS
        h = args.precision != "fp32"
        kname = {"gemm_kernel": (f"gemm_hp_kernel<{args.precision}> (persistent; 256x256 and 128x128 tile instantiations, all launches)"
                                 if h else "gemm_kernel"),
                 "sa_conv_kernel": (f"sa_conv16p_kernel<{args.precision}> (+ sa_edge_meta_kernel)" if h else "sa_conv_kernel")}
        traffic, traffic_src = None, None
        try:
            tfile = TRAFFIC_FILE.format(precision=args.precision)
            t = json.load(open(tfile))
            if t.get("precision") == args.precision and dom in t.get("kernels", {}):
                k = t["kernels"][dom]
                traffic = (k["fetch_bytes_per_step"] + k["write_bytes_per_step"]) / max(1, k["launches_per_step"])
                traffic_src = os.path.relpath(tfile, ROOT)
        except (OSError, ValueError, KeyError):
            pass
        abytes = algorithmic_bytes(geo)
        hbm = {}
        for name, nbytes in abytes.items():
            if name in per and per[name][0] > 0:
                gbps = nbytes / (per[name][0] * 1e-3) / 1e9
                hbm[name] = {"algorithmic_bytes_per_step": nbytes, "ms_per_step": round(per[name][0], 4), "launches": per[name][1],
                             "achieved_GBps": round(gbps, 1), "frac_of_8TBps": round(gbps / PEAK_HBM_GBPS, 4)}
        notes = {"f16x3": "f16x3 issues 3 fp16 MFMAs per algorithmic product: ceiling for algorithmic FLOPs is peak/3",
                 "fp16": "one fp16 MFMA per product", "bf16": "one bf16 MFMA per product", "fp32": "exact fp32 MFMA"}
        line = {
            "metric": "classified points/sec", "value": pts / dt, "unit": "points/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "ms_per_step_median": statistics.median(per_step),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: batch_size 8 x 16384-pt 2 m voxels, k=32, xyz-only, 1 batch per GPU per step, "
                                   f"{nb} distinct seeded batches in rotation, inputs resident in HBM",
                       "setup": "one untimed forward per distinct batch before the warmup steps (sizes the caching allocator)",
                       "global_batch_voxels": world * BATCH, "points_per_step": world * BATCH * NPTS, "C": C,
                       "level_sizes_batch0": sizes, "parallelism": f"voxel-batch sharding x{world}, RCCL all-gather of logits",
                       "pipeline": "2 HIP streams: geometry(i+1) || features(i) (features high priority)" if args.pipeline else "sequential"},
            "end_to_end_tflops_algorithmic": 2.0 * total_macs * args.steps / dt / 1e12,
            "pcie_inclusive": pcie,
            "roofline": {"bound": "mfma", "kernel": kname.get(dom, dom), "achieved": achieved, "peak": peak,
                         "unit": "TFLOP/s", "frac": achieved / peak, "traffic": traffic, "traffic_unit": "bytes per launch (FETCH_SIZE x2 + WRITE_SIZE)",
                         "traffic_source": traffic_src, "note": notes[args.precision],
                         "launches_per_step": dom_launches, "kernel_ms_per_step": dom_ms,
                         "algorithmic_gflop_per_step": 2.0 * kmacs[dom] / 1e9},
            "hbm_kernels": hbm,
            "kernel_ms_per_step": {kname: round(v[0], 4) for kname, v in sorted(per.items(), key=lambda kv: -kv[1][0])},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(host[0])
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
