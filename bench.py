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
(weak scaling, voxel batches are independent); the path's only collective, an RCCL all-gather of the
per-point logits, runs once behind the last step over all steps' logits (``--gather final``, the
default: north_star's "RCCL only for the final gather") or after every step (``--gather per-step``),
inside the timed region either way.  Rank 0 prints ONE JSON line.

``roofline`` is measured live: after the timed regions one extra, sequential step is run with HIP
events around every run of consecutive launches of one kernel class (on the launch stream) and the
dominant kernel's algorithmic FLOPs are divided by its measured time; ``traffic`` comes from the
committed rocprofv3 PMC summary of the same command (``profiles/r6_<precision>_hbm_traffic.json``).
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
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "r6_{precision}_hbm_traffic.json")   # written by tools/profile_round.sh


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=48)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batches", type=int, default=8, help="distinct seeded voxel batches the steps rotate over")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-voxel0", action="store_true",
                    help="time the CPU oracle on voxel 0 of batch 0 only (seconds) instead of the whole 8-voxel batch (~2 min)")
    ap.add_argument("--cpu-baseline-sweep", action="store_true", help="thread counts {16, 64, all} timed once each, the best re-timed")
    ap.add_argument("--cpu-baseline-child", default=None, metavar="THREADS:VOXELS:PASSES", help=argparse.SUPPRESS)
    ap.add_argument("--gather", default="final", choices=["final", "per-step"],
                    help="--gpus N: 'final' (default) = the path's only collective, the RCCL all-gather of the per-point logits, runs ONCE "
                         "after the last step over all steps' logits (north_star: 'RCCL only for the final gather'; still inside the "
                         "timed region); 'per-step' = one all-gather per step on the launch stream")
    ap.add_argument("--no-workloads", action="store_true",
                    help="skip the extra `workloads` object (configs[2], configs[4] in f16x3 and fp16, a surface-like batch)")
    ap.add_argument("--no-plot-workload", action="store_true", help="skip the configs[3] 10 M-point plot entry of `workloads` (~10 s)")
    ap.add_argument("--workload", default="voxels", choices=["voxels", "plot"],
                    help="voxels (default): BASELINE configs[1], one voxel batch per GPU per step (weak scaling); plot: BASELINE "
                         "configs[3], ONE synthetic plot voxelised, classified and back-projected by all ranks together (strong scaling)")
    ap.add_argument("--plot-points", type=int, default=10_000_000, help="--workload plot: points of the synthetic plot")
    ap.add_argument("--no-pcie", action="store_true", help="skip the second (H2D/D2H-inclusive) timed region")
    ap.add_argument("--no-single-call", action="store_true",
                    help="skip the `single_call` region (lone forwards, searches beside the features): profiling runs whose per-kernel "
                         "durations must be those of one kernel at a time")
    ap.add_argument("--pipeline", type=int, default=1,
                    help="1: two-stream pipeline over the step sequence (default); 0: strictly sequential forwards")
    ap.add_argument("--precision", default="f16x3", choices=["f16x3", "fp16", "bf16", "fp32"],
                    help="f16x3: split-fp16 MFMA (3 MFMAs per product, fp32-class accuracy; the parity default); fp16 / bf16: one "
                         "MFMA per product (the reference's autocast arithmetic; outside the 1e-4 bar); fp32: fp32 MFMA")
    ap.add_argument("--engine-opt", action="append", default=[], metavar="KEY=VALUE",
                    help="EngineOptions field for the benched Net (A/B runs), e.g. --engine-opt res_streams=2 --engine-opt sampler=sort")
    return ap.parse_args(argv)


def engine_options(pairs):
    """--engine-opt KEY=VALUE ... -> keyword arguments of Net (values typed like the EngineOptions defaults)."""
    from pointstowood_amd.engine import EngineOptions
    defaults, out = EngineOptions(), {}
    for kv in pairs:
        key, _, val = kv.partition("=")
        cur = getattr(defaults, key)   # AttributeError names an unknown option
        out[key] = (val.lower() not in ("0", "false", "no")) if isinstance(cur, bool) else type(cur)(val)
    return out


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


def hoist_saved_macs(geo, opts):
    """MACs the engine does NOT execute where an FP module's layer 0 runs its interpolated half on the coarse rows
    (EngineOptions.fp_hoist: relu(W [interp(y) | skip] + b) = relu(W_s skip + b + interp(W_i y))): Fc x N0 x (fine - coarse rows)
    per module that takes the route.  The ALGORITHMIC figure (algorithmic_macs) stays the reference formulation's."""
    if not getattr(opts, "fp_hoist", False):
        return 0
    rows = [geo.B] + [geo.levels[l].n for l in (3, 2, 1, 0)]          # coarse of fp4, fp3, fp2, fp1, then N
    n0 = [24 * C, 20 * C, 16 * C, 16 * C]                             # layer 0's outputs: fp4..fp1 (model.py:226-233)
    return sum(16 * C * n0[i] * (rows[i + 1] - rows[i]) for i in range(4) if 0 < rows[i] <= opts.fp_hoist_ratio * rows[i + 1])


def algorithmic_bytes(geo, opts=None):
    """SURVEY.md 8(d) algorithmic bytes of the memory-bound kernels (fp32 features, int32 indices, each datum once).  The
    interpolation class counts the FP modules that RUN the interpolation kernel; one that takes the fp_hoist route interpolates
    inside its GEMM's epilogue and leaves this class only its weight records (positions in, 16 B per fine row out)."""
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
    hoisted = [bool(opts is not None and opts.fp_hoist and 0 < coarse[i] <= opts.fp_hoist_ratio * fine[i]) for i in range(4)]
    out["interp_concat"] = sum((12 * (coarse[i] + fine[i]) + 16 * fine[i]) if hoisted[i] else
                               (4 * Fc[i] * coarse[i] + 12 * (coarse[i] + fine[i]) + 4 * Fc[i] * fine[i]) for i in range(4))
    return out


def search_pairs(geo):
    """the neighbour searches' ALGORITHMIC work = candidate-distance evaluations of the reference's brute-force
    definition, sum over voxels of queries x candidates (model.py:118 radius: M1 x N; :120 knn: M2 x M1, M3 x M2; :149
    knn_interpolate: M2 x M3, M1 x M2, N x M1).  The grid-indexed kernels evaluate far fewer (only candidates within reach of
    a query block); like the FLOP formula, the figure is the reference algorithm's, not this implementation's."""
    import torch
    cnt = [torch.diff(geo.levels[l].ptr.long()).cpu() for l in range(4)]
    dot = lambda a, b: int((cnt[a] * cnt[b]).sum())
    return {"ball_query": dot(1, 0), "knn": dot(2, 1) + dot(3, 2), "knn2": dot(2, 3) + dot(1, 2) + dot(0, 1)}


# fp32 VALU peak (MI355X_MICROARCH.md: 157.3 TFLOP/s vector fp32) in candidate-distance evaluations: 3 subtracts, 3 multiplies,
# 2 adds per pair (oracle/ops.py's ((dx*dx)+(dy*dy))+(dz*dz), no FMA) = 8 FLOP; the compare / insertion is not counted
PEAK_PAIRS_PER_S = 157.3e12 / 8.0

SEARCH_EVAL_FILE = os.path.join(ROOT, "profiles", "r6_search_evaluated.json")
MFMA_BUSY_FILE = os.path.join(ROOT, "profiles", "r6_{precision}_mfma_busy.csv")

KERNEL_OF = {"gemm_hoist": "gemm_kernel", "gemm_res": "gemm_kernel", "gemm_mlp": "gemm_kernel", "sa_conv": "sa_conv_kernel"}


def _geom_hash():
    """sha256 of the geometry kernels' source (the file the search kernels live in)."""
    import hashlib
    return hashlib.sha256(open(os.path.join(ROOT, "pointstowood_amd", "csrc", "p2w_geom.hip"), "rb").read()).hexdigest()[:16]


def profile_step(net, data, reps=3):
    """Per-kernel-class GPU time of one sequential forward: HIP-event brackets around every run of consecutive launches of one
    class, on the stream they are launched on.  One untimed pass (the pipelined region before it leaves the chip at another
    clock), then `reps` bracketed forwards back to back; a class' time is the median over them."""
    import torch
    eng = net._engine
    keep = {}
    streams, eng.res_streams = eng.res_streams, 1   # per-kernel durations: no two kernels in flight while they are timed
    net(data)
    runs = []
    for _ in range(reps):
        eng.events, eng.events_grouped = [], True   # one HIP-event pair per run of consecutive launches of one kernel class
        keep = {"geometry_only": True}   # the geometry (level sizes) without the fp32 copies a full `keep` asks the engine for
        net(data, keep=keep)
        eng.flush_events()
        torch.cuda.synchronize()
        ev, eng.events, eng.events_grouped = eng.events, None, False
        per = {}
        for name, s, e, launches in ev:
            kname = KERNEL_OF.get(name, name)
            t, n = per.get(kname, (0.0, 0))
            per[kname] = (t + s.elapsed_time(e), n + launches)
        runs.append(per)
    eng.res_streams = streams
    per = {kname: (statistics.median(r[kname][0] for r in runs), runs[0][kname][1]) for kname in runs[0]}
    return per, keep["geometry"]


MFMA_PER_PRODUCT = {"f16x3": 3, "fp16": 1, "bf16": 1, "fp32": 1}


def mfma_busy(precision):
    """SQ_VALU_MFMA_BUSY_CYCLES share per GEMM instantiation from the round's committed PMC summary (separate --pmc pass of the
    same command: tools/profile_round.sh), or None."""
    import csv
    try:
        rows = list(csv.DictReader(l for l in open(MFMA_BUSY_FILE.format(precision=precision)) if not l.lstrip('"').startswith("#")))
    except OSError:
        return None
    out = {}
    for r in rows:
        name = r.get("kernel") or r.get("Kernel_Name") or ""
        if "gemm_hp" in name:
            for key in ("mfma_busy_pct", "MFMA_busy_pct", "mfma_busy"):
                if key in r:
                    out[name[:80]] = float(r[key])
                    break
    return {"source": os.path.relpath(MFMA_BUSY_FILE.format(precision=precision), ROOT), "percent_by_instantiation": out} if out else None


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline_child(spec):
    """One timing leg of the CPU baseline in a FRESH process (bench.py --cpu-baseline-child THREADS:VOXELS:PASSES): the
    thread pools are sized before the first torch call of the process (OMP / MKL environment set by the parent, then
    torch.set_num_threads), nothing of the GPU run's host state (pinned buffers, HIP runtime threads, the allocator) is
    around.  Prints {"times": [...], "points": n}: 1 untimed warm-up pass, then PASSES timed ones."""
    threads, nvox, passes = (int(v) for v in spec.split(":"))
    import torch
    torch.set_num_threads(threads)
    from oracle import net as onet
    from pointstowood_amd import synthetic_weights as weights
    batch0 = host_batch(0, 0)
    sd = weights.synth_state_dict(1, C, seed=0)
    n = int(batch0["ptr"][nvox])
    pos, refl = batch0["pos"][:n].clone(), batch0["reflectance"][:n].clone()
    bidx, sf = batch0["batch"][:n].clone(), batch0["sf"][:nvox].clone()
    run = lambda: onet.forward(sd, pos, bidx, refl, sf, k=K_NBR)
    run()
    ts = []
    for _ in range(passes):
        t0 = time.perf_counter()
        run()
        ts.append(time.perf_counter() - t0)
    print(json.dumps({"times": ts, "points": n, "threads": torch.get_num_threads()}), flush=True)


def cpu_baseline(full=True, sweep_threads=False):
    """CPU oracle (oracle/net.py, a port pinned to the reference's own outputs) on batch 0 of this benchmark as ONE batch, like
    the GPU runs it (same batch definition, same k, C, weights) - BASELINE.md section 4.  Default: the WHOLE batch (8 voxels,
    131 072 points: ~30 s per pass on a 64-core EPYC - the oracle's [E, C] edge tensors fall out of cache) at 64 threads (the
    best count in every sweep of rounds 3 - 5), one warm-up pass + 3 timed ones in a fresh child process (~2 min); full=False:
    voxel 0 only; sweep_threads: thread counts {16, 64, all host cores} timed once each first, the best re-timed.  The child's
    thread pools are pinned before its first torch call, so the number does not depend on what the GPU part of this run left
    behind on the host (round 3: 3.9 k vs 13.6 k points/s for the same code)."""
    nvox = BATCH if full else 1
    cores = os.cpu_count() or 1
    sweep = sorted({min(cores, t) for t in (16, 64, cores)}) if sweep_threads else [min(cores, 64)]

    def leg(threads, passes):
        env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads), OMP_PROC_BIND="false",
                   OMP_WAIT_POLICY="PASSIVE", HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-child", f"{threads}:{nvox}:{passes}"],
                           env=env, capture_output=True, text=True, timeout=600 if nvox > 1 else 300)
        if r.returncode != 0:
            raise RuntimeError("cpu baseline child failed: " + r.stderr[-2000:])
        return json.loads(r.stdout.strip().splitlines()[-1])

    tried = {}
    if len(sweep) > 1:
        for threads in sweep:
            out = leg(threads, 1)
            tried[threads] = out["times"][0]
    best = min(tried, key=lambda t: (tried[t], t)) if tried else sweep[0]
    passes = 3 if full else 5
    final = leg(best, passes)
    n = final["points"]
    dt = statistics.median(final["times"])
    return {"value": n / dt, "unit": "points/s", "cores": best, "kind": "port", "cpu_model": cpu_model(), "host_cores": cores,
            "cores_used_of_present": f"{best}/{cores}", "ms_per_batch": dt * 1e3, "voxels": nvox,
            "threads_tried": {str(t): round(v, 3) for t, v in tried.items()},
            "passes_s": [round(t, 3) for t in final["times"]], "best_pass_points_per_s": round(n / min(final["times"]), 1),
            "sample": f"voxels 0..{nvox - 1} of batch 0 as one batch ({n} pts, U2-16k seeds 123..{122 + nvox}), k={K_NBR}, C={C}, fp32; "
                      f"{best} threads in a fresh process: median of {passes} passes after 1 warm-up, {dt:.2f} s per pass"}


def device_feed(vox, device):
    from pointstowood_amd import synthetic_voxels as synth
    return Feed.to_device(synth.collate(vox), device)


def hbm_kernels(per, geo, opts):
    """SURVEY.md 8(d) algorithmic bytes / measured time / 8 TB/s for the memory-bound kernels of one profiled forward."""
    out = {}
    for name, nbytes in algorithmic_bytes(geo, opts).items():
        if name in per and per[name][0] > 0:
            gbps = nbytes / (per[name][0] * 1e-3) / 1e9
            out[name] = {"algorithmic_bytes_per_step": nbytes, "ms_per_step": round(per[name][0], 4), "launches": per[name][1],
                         "achieved_GBps": round(gbps, 1), "frac_of_8TBps": round(gbps / PEAK_HBM_GBPS, 4)}
    return out


def measure_workload(net, data, reps=5, hbm=False):
    """One extra workload: setup forward (sizes the allocator), `reps` pipelined steps over the same batch, then one
    sequential profiled step for the per-kernel times and the algorithmic FLOPs of the level sizes actually produced."""
    import torch
    net(data)
    for _ in net.stream(data for _ in range(2)):   # sizes the allocator pools of both feature streams
        pass
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in net.stream(data for _ in range(reps)):
        pass
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    per, geo = profile_step(net, data)
    total_macs, kmacs, sizes = algorithmic_macs(geo)
    n = int(data.pos.shape[0])
    top = sorted(per.items(), key=lambda kv: -kv[1][0])[:3]
    out = {"points": n, "voxels": int(data.sf.numel()), "steps": reps, "ms_per_batch": round(dt * 1e3, 3),
           "range_fallbacks": int(getattr(net._engine, "range_fallbacks", 0)),
           "points_per_s": round(n / dt, 1), "end_to_end_tflops_algorithmic": round(2.0 * total_macs / dt / 1e12, 2),
           "M_over_N": [round(m / max(n, 1), 3) for m in sizes["M"]], "kernel_ms_top3": {k: round(v[0], 3) for k, v in top}}
    if hbm:   # the memory-bound kernels where they have work (B = 64: 1 M points per launch)
        out["hbm_kernels"] = hbm_kernels(per, geo, net.engine_options)
    for kname, macs in kmacs.items():
        if kname in per and per[kname][0] > 0:
            out[kname + "_tflops_algorithmic"] = round(2.0 * macs / (per[kname][0] * 1e-3) / 1e12, 1)
    return out


def plot_workload(args, device, n=10_000_000, reps=2):
    """BASELINE configs[3] at its stated size on this GPU: the 10 M-point synthetic forest plot through
    pipeline.segment_plot (voxelise 2 m + 4 m -> classify every voxel -> back-project with the k = 64 median vote;
    reference predict.py:116-156, src/predicter.py:193-234, src/preprocessing.py:79-127) on a Net of its own.  One untimed run
    (allocator), then `reps` timed ones (stage times of the last); then ONE more run with the forwards issued one after the other
    and a HIP-event bracket per kernel class: the plot's level sizes, algorithmic FLOPs and per-class times - how efficiently the
    plot's ragged forwards run compared with the bench batch."""
    import torch
    from pointstowood_amd import Net
    from pointstowood_amd import synthetic_weights as weights
    from pointstowood_amd.pipeline import segment_plot
    from pointstowood_amd.synthetic_voxels import forest_plot
    net = Net(num_classes=1, C=C, k=K_NBR, precision=args.precision, **engine_options(args.engine_opt))
    net.load_state_dict(weights.synth_state_dict(1, C, seed=0), strict=True)
    net = net.to(device).eval()
    pc = forest_plot(n, side=100.0).to(device)
    gen = lambda: torch.Generator(device=device).manual_seed(0)
    segment_plot(pc, net, generator=gen())
    times, stats = [], {}
    for _ in range(reps):
        torch.cuda.synchronize()
        stats = {}
        t0 = time.perf_counter()
        n_z, label, pwood = segment_plot(pc, net, generator=gen(), stats=stats)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    ok = bool(torch.isfinite(pwood).all()) and set(label.unique().tolist()) <= {0.0, 1.0}
    dt = min(times)
    out = {"points": n, "voxels": stats.get("voxels"), "classified_points": stats.get("classified_points"), "runs": reps,
           "ms_per_plot": round(dt * 1e3, 1), "ms_per_plot_all": [round(t * 1e3, 1) for t in times],
           "plot_points_per_s": round(n / dt, 1),
           "classified_points_per_s": round(stats.get("classified_points", 0) / max(stats.get("classify_s", 1e-9), 1e-9), 1),
           "stages_s_last_run": {k: round(v, 4) for k, v in stats.items() if k.endswith("_s")},
           "forwards": len(stats.get("batch_points", [])), "max_points_per_forward": stats.get("max_points"),
           "outputs_ok": ok, "dtype": args.precision, "range_fallbacks": int(getattr(net._engine, "range_fallbacks", 0))}
    # efficiency: sequential forwards with per-class event brackets
    eng = net._engine
    macs, kmacs_tot, fwd_sizes = 0, {}, []
    feats = eng.features

    def features(geo, keep=None):
        nonlocal macs
        t, km, sz = algorithmic_macs(geo)
        macs += t
        for k_, v in km.items():
            kmacs_tot[k_] = kmacs_tot.get(k_, 0) + v
        fwd_sizes.append(dict(sz, voxels=geo.B))
        return feats(geo, keep)
    eng.features = features
    net.stream = lambda batches: (net(b) for b in batches)     # one forward after the other on one stream: the brackets need it
    try:
        segment_plot(pc, net, generator=gen())                 # (untimed: this stream's allocator pool has not seen these forwards yet)
        macs, kmacs_tot, fwd_sizes = 0, {}, []
        eng.events, eng.events_grouped = [], True
        st2 = {}
        segment_plot(pc, net, generator=gen(), stats=st2)
        eng.flush_events()
        torch.cuda.synchronize()
        ev = eng.events
    finally:
        eng.events, eng.events_grouped = None, False
        del eng.features, net.stream
    per = {}
    for name, s_, e_, launches in ev:
        kname = KERNEL_OF.get(name, name)
        t_, n_ = per.get(kname, (0.0, 0))
        per[kname] = (t_ + s_.elapsed_time(e_), n_ + launches)
    cls_s = max(stats.get("classify_s", 0.0), 1e-9)
    out["classify_tflops_algorithmic"] = round(2.0 * macs / cls_s / 1e12, 2)
    out["algorithmic_gflop_per_plot"] = round(2.0 * macs / 1e9, 1)
    out["sequential_profile"] = {
        "classify_s": round(st2.get("classify_s", 0.0), 4),
        "kernel_ms_per_plot": {k: round(v[0], 2) for k, v in sorted(per.items(), key=lambda kv: -kv[1][0])[:8]},
        "kernel_launches_per_plot": {k: v[1] for k, v in sorted(per.items(), key=lambda kv: -kv[1][0])[:3]},
        "note": "HIP-event brackets per run of consecutive launches of one class, forwards issued one after the other: a class' time "
                "includes the gaps to the next class' first launch (the last search's includes the host's wait for the level sizes)",
    }
    for kname, m_ in kmacs_tot.items():
        if kname in per and per[kname][0] > 0:
            out["sequential_profile"][kname + "_tflops_algorithmic"] = round(2.0 * m_ / (per[kname][0] * 1e-3) / 1e12, 1)
    tot = lambda key: [sum(f[key][i] for f in fwd_sizes) for i in range(3)]
    big = max(fwd_sizes, key=lambda f: f["N"]) if fwd_sizes else None
    out["level_sizes_total"] = {"N": sum(f["N"] for f in fwd_sizes), "M": tot("M"), "E": tot("E"), "forwards": len(fwd_sizes)} if fwd_sizes else None
    out["level_sizes_largest_forward"] = big
    # The same voxels through the reference's CLI surface (predict.py --voxels = predicter.classify_voxels; the voxel list in
    # memory instead of 14 k files): host-side dataset feed + collation -> stream pipeline -> one D2H copy, against the
    # reference-shaped loop (BalancedBatchSampler at --batch_size 8, one forward and one D2H copy per batch) on every 4th voxel.
    try:
        from pointstowood_amd import DataLoader, predicter
        from pointstowood_amd.preprocessing import voxelise
        vox, _ = voxelise(pc, (2.0, 4.0), 128, 16384, generator=gen())
        host_vox = [v.cpu() for v in vox]
        del vox
        ds = predicter.VoxelDataset(host_vox)
        predicter.classify_voxels(net, predicter.VoxelDataset(host_vox[::16]), 0.5, device)      # (allocator)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rows = predicter.classify_voxels(net, ds, 0.5, device)
        dt_new = time.perf_counter() - t0
        sub = predicter.VoxelDataset(host_vox[::4])
        sampler = predicter.BalancedBatchSampler(sub, 8)
        t0 = time.perf_counter()
        n_old = sum(predicter.classify_batch(net, d, 0.5, device).shape[0] for d in DataLoader(sub, batch_sampler=sampler, num_workers=0))
        dt_old = time.perf_counter() - t0
        out["voxel_directory_cli"] = {
            "voxels": len(ds), "points": int(rows.shape[0]), "s": round(dt_new, 3), "points_per_s": round(rows.shape[0] / dt_new, 1),
            "path": "predicter.classify_voxels: PointBudgetSampler (262144 points per forward) -> Net.stream -> one D2H copy",
            "batch_size_8_loop": {"voxels": len(sub), "points": int(n_old), "s": round(dt_old, 3), "points_per_s": round(n_old / dt_old, 1),
                                  "path": "BalancedBatchSampler(8) + one classify_batch (forward + D2H) per batch: the reference's loop shape"}}
        del rows, host_vox, ds, sub
    except Exception as e:   # the headline line must not die on an auxiliary workload
        out["voxel_directory_cli"] = {"error": repr(e)[:300]}
    del pc, n_z, label, pwood, net
    torch.cuda.empty_cache()
    return out


def extra_workloads(net, args, device):
    """The other BASELINE.json workloads and a surface-like batch, measured in the same run as the bench line (a few steps
    each): driver-observable numbers for configs[2], configs[4] (f16x3 and, as the config says fp16, fp16) and for input
    that saturates the ball-query cap the way real TLS surfaces do (model.py:118; SURVEY.md 8d's second generator)."""
    import torch
    from pointstowood_amd import Net
    from pointstowood_amd import synthetic_voxels as synth
    from pointstowood_amd import synthetic_weights as weights
    out = {}
    cases = [
        ("configs[2] B=64 x 16384 xyz+reflectance", lambda: [synth.uniform_voxel(2.0, NPTS, 200 + i, True) for i in range(64)]),
        ("configs[4] B=128 mixed 512..16384", lambda: [synth.uniform_voxel(2.0, n, 400 + i, True) for i, n in enumerate(synth.mixed_sizes())]),
        ("surface B=8 x 16384 xyz-only (cylinders + blobs: ball-query cap saturated)",
         lambda: [synth.surface_voxel(2.0, NPTS, 300 + i, False) for i in range(BATCH)]),
    ]
    for name, make in cases:
        d = device_feed(make(), device)
        out[name] = dict(measure_workload(net, d, hbm=name.startswith("configs[2]")), dtype=args.precision)
        if name.startswith("configs[4]") and args.precision == "f16x3":
            net16 = Net(num_classes=1, C=C, k=K_NBR, precision="fp16", **engine_options(args.engine_opt))
            net16.load_state_dict(weights.synth_state_dict(1, C, seed=0), strict=True)
            net16 = net16.to(device).eval()
            out[name + " [fp16]"] = dict(measure_workload(net16, d), dtype="fp16",
                                         note="one fp16 MFMA per product: the reference's autocast arithmetic, outside the 1e-4 bar")
            del net16
        del d
        torch.cuda.empty_cache()
    # small batches: the reference's default --batch_size 8 on plot-sized voxels (~1355 points each): pipelined and one call at a time
    sb = [device_feed([synth.uniform_voxel(2.0, 1355, 100 * j + i, False) for i in range(BATCH)], device) for j in range(4)]
    for d in sb:
        net(d)
    for _ in net.stream(sb[i % 4] for i in range(8)):
        pass
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in net.stream(sb[i % 4] for i in range(100)):
        pass
    torch.cuda.synchronize()
    dt_p = (time.perf_counter() - t0) / 100
    t0 = time.perf_counter()
    for i in range(50):
        net(sb[i % 4])
        torch.cuda.synchronize()
    dt_s = (time.perf_counter() - t0) / 50
    out["small batches B=8 x 1355 xyz-only"] = {"points": BATCH * 1355, "pipelined_ms_per_batch": round(dt_p * 1e3, 3),
                                                "pipelined_points_per_s": round(BATCH * 1355 / dt_p, 1),
                                                "single_call_ms_per_batch": round(dt_s * 1e3, 3),
                                                "single_call_points_per_s": round(BATCH * 1355 / dt_s, 1)}
    del sb
    if not args.no_plot_workload:   # last, and on a Net of its own: its oversized voxels must not shape the other workloads' engine state
        out["configs[3] 10 M-point plot (voxelise + classify + back-project, 1 GPU)"] = plot_workload(args, device)
    return out


def plot_main(args, world, rank, device, dist):
    """--workload plot: BASELINE configs[3] - ONE synthetic plot, voxelised (2 m + 4 m grids, min 128 / max 16384 points),
    classified (LPT-sharded voxel batches) and back-projected (contiguous slices of the plot) by all ranks together:
    strong scaling.  A step = the whole plot once; value = plot points / s."""
    import torch
    from pointstowood_amd import Net
    from pointstowood_amd import synthetic_weights as weights
    from pointstowood_amd.pipeline import segment_plot
    from pointstowood_amd.synthetic_voxels import forest_plot
    net = Net(num_classes=1, C=C, k=K_NBR, precision=args.precision, **engine_options(args.engine_opt))
    net.load_state_dict(weights.synth_state_dict(1, C, seed=0), strict=True)
    net = net.to(device).eval()
    n = args.plot_points
    side = max(10.0, 100.0 * (n / 10_000_000) ** 0.5)
    pc = forest_plot(n, side=side).to(device)
    gen = lambda: torch.Generator(device=device).manual_seed(0)
    steps, warm = max(1, min(args.steps, 3)), max(1, min(args.warmup, 1))
    for _ in range(warm):
        segment_plot(pc, net, generator=gen(), dist=dist)
    times, stats = [], {}
    for _ in range(steps):
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        stats = {}
        t0 = time.perf_counter()
        n_z, label, pwood = segment_plot(pc, net, generator=gen(), stats=stats, dist=dist)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t)
        times.append(dt)
    assert bool(torch.isfinite(pwood).all())
    if rank == 0:
        dt = sum(times) / len(times)
        print(json.dumps({
            "metric": "plot points/sec (voxelise + classify + back-project)", "value": n / dt, "unit": "points/s", "n_gpus": world,
            "steps": steps, "warmup": warm, "ms_per_step": dt * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": args.precision, "data": "synthetic",
            "config": {"workload": f"BASELINE configs[3]: {n}-point synthetic forest plot ({side:.0f} m square), grid_size 2.0/4.0, min_pts 128, "
                                   "max_pts 16384, voxel batches of a fifth of a rank's share (262144 .. 2097152 points, 1 voxel per 1024 points) LPT-sharded over the ranks, one all-gather "
                                   "of the float32 probabilities (4 B per classified point), back-projection (k=64 median vote) owned by x-slabs of the plot "
                                   "against the voxels within a halo, one all-gather of the results",
                       "voxels": stats.get("voxels"), "classified_points": stats.get("classified_points"), "C": C,
                       "parallelism": f"voxel-batch sharding x{world} + spatial (x-slab) sharding of the back-projection x{world}, 2 RCCL all-gathers"},
            "stages_s_rank0_last_step": {k: round(v, 4) for k, v in stats.items() if k.endswith("_s")},
            "classified_points_per_s": stats.get("classified_points", 0) / max(stats.get("classify_s", 1e-9), 1e-9),
        }), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    args = parse_args()
    if os.environ.get("P2W_BENCH_WATCHDOG"):   # tests: a bench that does not finish dumps every thread's stack and exits instead of hanging
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["P2W_BENCH_WATCHDOG"]), exit=True)
    if args.cpu_baseline_child:
        return cpu_baseline_child(args.cpu_baseline_child)
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

    if args.workload == "plot":
        return plot_main(args, world, rank, device, dist)

    from pointstowood_amd import synthetic_weights as weights
    from pointstowood_amd import Net
    from pointstowood_amd.dist import gather_logits
    net = Net(num_classes=1, C=C, k=K_NBR, precision=args.precision, **engine_options(args.engine_opt))
    net.load_state_dict(weights.synth_state_dict(1, C, seed=0), strict=True)
    net = net.to(device).eval()
    nb = max(1, args.batches)
    host = [host_batch(rank, j) for j in range(nb)]
    resident = [Feed.to_device(b, device) for b in host]
    pinned = [{k: b[k].pin_memory() for k in Feed.FIELDS} for b in host]
    out_host = [torch.empty(BATCH * NPTS, dtype=torch.float32).pin_memory() for _ in range(nb)]
    peak = PEAK_TFLOPS[args.precision]

    gather_marks = []

    def run(n, pcie=False, stamps=None):
        """n steps = n full forwards (geometry + features) of one voxel batch each, rotating over the distinct batches.
        With --pipeline the engine's two-stream software pipeline overlaps the geometry phase of step i+1 with the
        feature phase of step i.  pcie: inputs come from pinned host memory, logits go back to it, inside the region."""
        out = None
        held = []   # --gather final: every step's logits, gathered once behind the last step

        def feed():
            for i in range(n):
                yield Feed.to_device(pinned[i % nb], device, non_blocking=True) if pcie else resident[i % nb]

        def gather(t):
            # the path's only collective, bracketed by events on the launch stream (its own time)
            g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            g0.record()
            o = gather_logits(t, dist)
            g1.record()
            gather_marks.append((g0, g1))
            return o

        def finish(i, logits):
            o = logits
            if world > 1:
                if args.gather == "per-step":
                    o = gather(logits)
                else:
                    held.append(logits)
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
        if held:   # ONE all-gather of all n steps' logits (n x 0.5 MB per rank), still inside the timed region
            out = gather(torch.cat(held))
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
        ranks = None
        if world > 1:
            # every rank's own wall time and the time its gathers took (diagnosis of a scaling curve: a slow rank, or the
            # collective, shows up here); `dt` = the maximum over ranks, as the contract says
            marks_ = gather_marks[-n:] if args.gather == "per-step" else gather_marks[-1:]
            gather_ms = sum(a.elapsed_time(b) for a, b in marks_) / n
            mine = torch.tensor([dt, gather_ms], device=device, dtype=torch.float64)
            allr = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(allr, mine)
            dts = [float(t[0]) for t in allr]
            ranks = {"ms_per_step_min": min(dts) / n * 1e3, "ms_per_step_max": max(dts) / n * 1e3,
                     "ms_per_step_by_rank": [round(v / n * 1e3, 4) for v in dts],
                     "gather_ms_per_step_by_rank": [round(float(t[1]), 4) for t in allr]}
            dt = max(dts)
        marks = [s0] + stamps
        per_step = [marks[i].elapsed_time(marks[i + 1]) for i in range(n)]   # completion-to-completion, ms
        return dt, per_step, out, ranks

    # setup, not warmup: one untimed forward per distinct batch so that the caching allocator has seen every workspace size
    # (level sizes are data-dependent; a first-time hipMalloc inside the timed region costs milliseconds); then W warmup steps
    # (the caching allocator keeps one pool per stream: the pipeline's feature streams must have seen every batch they will get -
    # one pass over the distinct batches through the pipeline itself; one lone forward for the profiled step's stream)
    net(resident[0])
    if args.pipeline:
        run(nb + (nb % 2))
    else:
        for d in resident[1:]:
            net(d)
    run(args.warmup)
    dt, per_step, out, rank_stats = timed(args.steps, False)
    assert bool(torch.isfinite(out).all())
    pcie = None
    if not args.no_pcie:
        run(min(2, args.warmup), pcie=True)
        dt_p, per_p, _, _ = timed(args.steps, True)
        pcie = {"value": world * args.steps * BATCH * NPTS / dt_p, "unit": "points/s", "ms_per_step": dt_p / args.steps * 1e3,
                "ms_per_step_median": statistics.median(per_p),
                "what": "same steps with the inputs (pos, batch, reflectance, sf, ptr: 2.7 MB) copied from pinned host memory and "
                        "the logits (0.5 MB) copied back inside the timed region"}

    # the call the reference makes (predicter.py:198: outputs = model(data)): ONE forward per step on the caller's stream, the host
    # waits for every step's logits before it issues the next (no cross-batch pipeline; inside the forward the searches run beside
    # the features - EngineOptions.overlap)
    single_call = None
    if world == 1 and not args.no_single_call:
        for d in resident:
            net(d)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            net(resident[i % nb])
            torch.cuda.synchronize()
        dt_sc = (time.perf_counter() - t0) / args.steps
        single_call = {"ms_per_step": dt_sc * 1e3, "value": BATCH * NPTS / dt_sc, "unit": "points/s", "steps": args.steps,
                       "what": "one net(data) per step + torch.cuda.synchronize() after each (the reference's loop shape, predicter.py:198): "
                               "searches on a second stream beside the features inside the forward, no cross-batch pipeline"}

    if rank == 0:
        per, geo = profile_step(net, resident[0])
        total_macs, kmacs, sizes = algorithmic_macs(geo)
        dom = max(kmacs, key=lambda kname: per[kname][0])
        dom_ms, dom_launches = per[dom]
        achieved = 2.0 * kmacs[dom] / (dom_ms * 1e-3) / 1e12
        saved = hoist_saved_macs(geo, net.engine_options) if dom == "gemm_kernel" else 0
        pts = world * args.steps * BATCH * NPTS
        h = args.precision != "fp32"
        kname = {"gemm_kernel": (f"gemm_hp_kernel<{args.precision}> (persistent; 256x256, 128x128 and 64x128 tile instantiations, split-K tails, all launches)"
                                 if h else "gemm_kernel"),
                 "sa_conv_kernel": (f"sa_conv16p_kernel<{args.precision}> (+ sa_edge_meta_kernel)" if h else "sa_conv_kernel")}
        traffic, traffic_src, traffic_step = None, None, None
        try:
            tfile = TRAFFIC_FILE.format(precision=args.precision)
            t = json.load(open(tfile))
            if t.get("precision") == args.precision and dom in t.get("kernels", {}):
                k = t["kernels"][dom]
                traffic_step = k["fetch_bytes_per_step"] + k["write_bytes_per_step"]
                traffic = traffic_step / max(1, dom_launches)    # per launch of THIS line's launch count: traffic x launches = the class' bytes per step
                traffic_src = os.path.relpath(tfile, ROOT)
        except (OSError, ValueError, KeyError):
            pass
        hbm = hbm_kernels(per, geo, net.engine_options)
        pairs = search_pairs(geo)
        # The searches against the fp32 VALU peak: the roofline figure counts the distance evaluations the grid kernels PERFORM
        # (profiles/r6_search_evaluated.json: counted in a -DP2W_SLAB_PROFILE build on this workload's batch 0); the reference's
        # brute-force definition (queries x candidates per voxel) is what the kernels save, reported as a ratio, not as a fraction.
        evaluated, ev_src = {}, None
        try:
            ev = json.load(open(SEARCH_EVAL_FILE))
            from pointstowood_amd import build as _build
            # counted on one diagnostic build of the geometry kernels: only valid for the sources it was counted on
            if ev.get("geom_srchash") in (None, _geom_hash()):
                evaluated, ev_src = ev.get("evaluated_pairs_per_step", {}), os.path.relpath(SEARCH_EVAL_FILE, ROOT)
        except (OSError, ValueError):
            pass
        search = {"unit": "candidate-distance evaluations/s", "peak": PEAK_PAIRS_PER_S,
                  "peak_note": "fp32 VALU peak 157.3 TFLOP/s / 8 FLOP per evaluation",
                  "evaluated_source": ev_src, "kernels": {}}
        for name, npairs in pairs.items():
            if name in per and per[name][0] > 0:
                sec = per[name][0] * 1e-3
                k = {"ms_per_step": round(per[name][0], 4), "launches": per[name][1], "bruteforce_pairs_per_step": npairs}
                if name in evaluated:
                    k["evaluated_pairs_per_step"] = int(evaluated[name])
                    k["achieved"] = evaluated[name] / sec
                    k["frac_of_valu_peak"] = round(evaluated[name] / sec / PEAK_PAIRS_PER_S, 5)
                    k["bruteforce_over_evaluated"] = round(npairs / max(evaluated[name], 1), 1)
                search["kernels"][name] = k
        tot_ms = sum(v["ms_per_step"] for v in search["kernels"].values())
        if tot_ms > 0:
            search["ms_per_step"] = round(tot_ms, 4)
            ev_tot = sum(v.get("evaluated_pairs_per_step", 0) for v in search["kernels"].values())
            if ev_tot:
                search["achieved"] = ev_tot / (tot_ms * 1e-3)
                search["frac_of_valu_peak"] = round(search["achieved"] / PEAK_PAIRS_PER_S, 5)
            search["bruteforce_pairs_per_s"] = sum(pairs[k] for k in search["kernels"]) / (tot_ms * 1e-3)
            search["note"] = ("latency-bound kernels (VALU active 17-26 % of wave cycles, profiles/*_valu.csv): the fraction says how "
                              "little of the chip's arithmetic the exact searches need, not how well they use it")
        notes = {"f16x3": "f16x3 issues 3 fp16 MFMAs per algorithmic product: ceiling for algorithmic FLOPs is peak/3",
                 "fp16": "one fp16 MFMA per product", "bf16": "one bf16 MFMA per product", "fp32": "exact fp32 MFMA"}
        # Order of the line: the diagnostics FIRST (workloads, per-kernel tables), the contract keys LAST - whoever keeps only the
        # tail of the line keeps metric / value / roofline / cpu_baseline / single_call / pcie_inclusive.
        line = {}
        if world == 1 and not args.no_workloads:
            try:      # the headline line must not die on an auxiliary workload
                line["workloads"] = extra_workloads(net, args, device)
            except Exception as e:
                line["workloads"] = {"error": repr(e)[:400]}
        line.update({
            "search": search,
            "hbm_kernels": hbm,
            "kernel_ms_per_step": {kname: round(v[0], 4) for kname, v in sorted(per.items(), key=lambda kv: -kv[1][0])},
            "timing_note": f"timed region {dt:.2f} s ({args.steps} steps) right after {args.warmup} warm-up steps: a cold-clock number; box-to-box spread of this "
                           "line is +-3-5 % (DVFS), same-box A/Bs are in docs/LAB_NOTES.md",
            # forwards the f16x3 range guard recomputed on the fp32 MFMA path (0 on this benchmark: a non-zero count would mean the
            # timed steps were not f16x3 steps)
            "range_fallbacks": int(getattr(net._engine, "range_fallbacks", 0)),
            "end_to_end_tflops_algorithmic": 2.0 * total_macs * args.steps / dt / 1e12,
        })
        if rank_stats is not None:
            line["ranks"] = rank_stats
        line.update({
            "metric": "classified points/sec", "value": pts / dt, "unit": "points/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "ms_per_step_median": statistics.median(per_step),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: batch_size 8 x 16384-pt 2 m voxels, k=32, xyz-only, 1 batch per GPU per step, "
                                   f"{nb} distinct seeded batches in rotation, inputs resident in HBM",
                       "global_batch_voxels": world * BATCH, "points_per_step": world * BATCH * NPTS, "C": C,
                       "level_sizes_batch0": sizes, "parallelism": f"voxel-batch sharding x{world}, RCCL all-gather of logits "
                                                                    + ("once after the last step" if args.gather == "final" else "after every step"),
                       "pipeline": ("HIP streams: geometry(i+1) || features(i), features alternating over "
                                    f"{net.engine_options.feature_streams} high-priority streams") if args.pipeline else "sequential"},
            "roofline": {"bound": "mfma", "kernel": kname.get(dom, dom), "achieved": achieved, "peak": peak,
                         "unit": "TFLOP/s", "frac": achieved / peak, "traffic": traffic, "traffic_unit": "bytes per launch (FETCH_SIZE x2 + WRITE_SIZE)",
                         "traffic_bytes_per_step": traffic_step, "traffic_source": traffic_src, "note": notes[args.precision],
                         "launches_per_step": dom_launches, "kernel_ms_per_step": dom_ms,
                         "algorithmic_gflop_per_step": 2.0 * kmacs[dom] / 1e9,
                         "executed_gflop_per_step": 2.0 * (kmacs[dom] - saved) / 1e9,
                         "executed_frac": 2.0 * (kmacs[dom] - saved) / (dom_ms * 1e-3) / 1e12 * MFMA_PER_PRODUCT[args.precision] / peak,
                         "mfma_busy": mfma_busy(args.precision)},
        })
        if world == 1 and not args.no_cpu_baseline:
            try:      # (a host without the memory or the time for the whole batch: fall back to voxel 0, then to an error entry)
                line["cpu_baseline"] = cpu_baseline(full=not args.cpu_baseline_voxel0, sweep_threads=args.cpu_baseline_sweep)
            except Exception as e:
                try:
                    line["cpu_baseline"] = dict(cpu_baseline(full=False), note="whole-batch run failed: " + repr(e)[:200])
                except Exception as e2:
                    line["cpu_baseline"] = {"error": repr(e2)[:400]}
        if single_call is not None:
            line["single_call"] = single_call
        # `value` is measured with the inputs resident in HBM, as the bench contract requires; SURVEY.md 8(d) words the metric "incl.
        # H2D/D2H": the same steps fed from pinned host memory with the logits copied back inside the region are `pcie_inclusive`
        line["pcie_inclusive"] = pcie
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
