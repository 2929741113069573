#!/usr/bin/env python3
"""Multi-GPU prediction on ONE GPU (verdict r4 item 7): every rank's share of the BASELINE configs[3] plot
(pipeline.segment_plot: 10 M points, 2 m + 4 m voxels) is run IN TURN through the real sharded code path with a stand-in for
torch.distributed that records what each rank contributes to the two exchanges and plays the other ranks' (recorded)
contributions back - so every rank back-projects against the true gathered classification.  Per world size W in {1, 2, 4, 8}:

  pass 1  each rank r: voxelise (replicated) -> classify its LPT share -> first exchange: its block is recorded, the run stops there
  pass 2  each rank r: the same again, now the first exchange returns every rank's recorded block -> back-projection of its slice

and the predicted plot time = max over ranks of (voxelise + classify + back-project) + the two all-gathers at a ring model of
the xGMI links.  Output: one JSON document (commit it as profiles/r5_scaling_prediction.json) the first real SCALE_r*.json can
be checked against; rank imbalance > 10 % of the classify stage is flagged (dist.batch_cost is the knob).

    python tools/scaling_predict.py [--points 10000000] [--worlds 1,2,4,8] [--out FILE]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from pointstowood_amd import Net, pipeline  # noqa: E402
from pointstowood_amd import synthetic_weights as weights  # noqa: E402
from pointstowood_amd.synthetic_voxels import forest_plot  # noqa: E402

XGMI_LINK_GBPS = 48.0      # one direction of one xGMI link as RCCL's ring all-gather sees it (MI355X_MICROARCH.md: 7 links x ~153 GB/s
                           # aggregate bidirectional per GPU; a ring uses two links per GPU): deliberately conservative


class _Stop(Exception):
    pass


class ReduceOp:
    MIN, MAX, SUM = "min", "max", "sum"


class FakeDist:
    """torch.distributed for ONE rank of a pretended world: collectives are served from `store` (contributions recorded per
    collective call index and rank); a rank whose peers have not been recorded yet stops at its first data exchange."""
    ReduceOp = ReduceOp

    def __init__(self, rank, world, store, log, stop=True, lengths_first=False):
        self.rank, self.world, self.store, self.log, self.calls, self.stop = rank, world, store, log, 0, stop
        self.lengths_first = lengths_first      # shard="slices": gather_rows = a lengths all-gather + the padded blocks

    def get_world_size(self, group=None):
        return self.world

    def get_rank(self, group=None):
        return self.rank

    def get_backend(self, group=None):
        return "nccl"

    def all_reduce(self, t, op=None, group=None):          # (the default budget's MIN over free memory: one GPU, one value)
        return None

    def barrier(self):
        return None

    def all_gather(self, out, t, group=None):
        c, self.calls = self.calls, self.calls + 1
        slot = self.store.setdefault(c, {})
        slot[self.rank] = t.detach().clone()
        self.log.append({"call": c, "rank": self.rank, "bytes_contributed": t.numel() * t.element_size()})
        if len(slot) < self.world:
            data = (c % 2 == 1) if self.lengths_first else True   # round 5's flow sends a lengths call in front of every data block
            if data and self.stop:         # a data block whose peers are not known yet: pass 1 ends here (a lengths call before it is
                raise _Stop()              # answered with this rank's own length for everybody: its buffer holds exactly its own rows)
            for o in out:
                o.copy_(t)
            return
        shape = out[0].shape
        for r in range(self.world):
            src = slot[r]
            if src.shape != shape:   # recorded in pass 1 under another padding: rows beyond a rank's own are padding either way
                buf = torch.zeros(shape, dtype=src.dtype, device=src.device)
                n = min(shape[0], src.shape[0])
                buf[:n] = src[:n]
                src = buf
            out[r].copy_(src)


REPEATS = 2
SHARD = "spatial"


def run_world(pc, net, world, gen):
    old = SHARD == "slices"
    store, log, ranks = {}, [], []
    cls_stats = []
    for r in range(world):      # pass 1: classify shares
        st = {}
        try:
            pipeline.segment_plot(pc, net, generator=gen(), stats=st, shard=SHARD,
                                  dist=FakeDist(r, world, store, log, lengths_first=old) if world > 1 else None)
        except _Stop:
            torch.cuda.synchronize()
        cls_stats.append(st)
    if world == 1:
        st = cls_stats[0]
        for _ in range(REPEATS - 1):
            s1 = {}
            pipeline.segment_plot(pc, net, generator=gen(), stats=s1)
            torch.cuda.synchronize()
            for key in ("voxelise_s", "classify_s", "backproject_s"):
                st[key] = min(st[key], s1[key])
        return {"world": 1, "ranks": [{"rank": 0, "voxelise_s": st["voxelise_s"], "classify_s": st["classify_s"],
                                       "backproject_s": st["backproject_s"], "forwards": len(st["batch_points"]),
                                       "points": sum(st["batch_points"])}],
                "exchange_bytes": [0, 0], "predicted_s": st["voxelise_s"] + st["classify_s"] + st["backproject_s"]}
    for r in range(world):      # pass 2: everything, with the peers' recorded blocks; REPEATS times, every stage's fastest run counts
        st = None                # (a single run of a 0.05 - 0.3 s stage on a box that has just changed its clock is +-20 %)
        for _ in range(REPEATS):
            s1 = {}
            d = FakeDist(r, world, store, [], stop=False, lengths_first=old)     # (the LAST exchange - the per-point results - is answered with the rank's own slice)
            pipeline.segment_plot(pc, net, generator=gen(), stats=s1, dist=d, shard=SHARD)
            torch.cuda.synchronize()
            if st is None:
                st = s1
            else:
                for key in ("voxelise_s", "classify_s", "backproject_s"):
                    st[key] = min(st[key], s1[key])
        # (classify_s includes the stand-in gather: a device copy)
        ranks.append({"rank": r, "voxelise_s": round(st["voxelise_s"], 4), "classify_s": round(st["classify_s"], 4),
                      "backproject_s": round(st["backproject_s"], 4), "forwards": len(st["batch_points"]), "points": sum(st["batch_points"]),
                      "backproject_tiers": st.get("backproject_tiers")})
    if old:
        blocks = store[1]
        max_rows = max(int(b.shape[0]) for b in blocks.values())
        ex1 = world * max_rows * blocks[0].shape[1] * blocks[0].element_size()            # padded all-gather of the classified points
        ex2 = 0
        if 3 in store:
            b2 = store[3]
            ex2 = world * max(int(b.shape[0]) for b in b2.values()) * b2[0].shape[1] * b2[0].element_size()
    else:   # two all-gathers of known sizes: the float32 probabilities, the (label, pwood) pairs
        ex1 = world * max(int(b.numel()) for b in store[0].values()) * 4
        ex2 = world * max(int(b.numel()) for b in store[1].values()) * 4 if 1 in store else 0
    ring = lambda nbytes: nbytes * (world - 1) / world / (XGMI_LINK_GBPS * 1e9)
    crit = max(x["voxelise_s"] + x["classify_s"] + x["backproject_s"] for x in ranks)
    cls = [x["classify_s"] for x in ranks]
    return {"world": world, "ranks": ranks, "exchange_bytes": [ex1, ex2],
            "exchange_s_ring_model": [round(ring(ex1), 4), round(ring(ex2), 4)],
            "classify_imbalance": round(max(cls) / (sum(cls) / len(cls)) - 1.0, 3),
            "predicted_s": round(crit + ring(ex1) + ring(ex2), 4)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=10_000_000)
    ap.add_argument("--worlds", default="1,2,4,8")
    ap.add_argument("--out", default=None)
    ap.add_argument("--repeats", type=int, default=2, help="runs per rank; every stage's fastest one counts")
    ap.add_argument("--shard", default="spatial", choices=["spatial", "slices"], help="segment_plot's sharding of the back-projection")
    args = ap.parse_args()
    global REPEATS, SHARD
    REPEATS = max(1, args.repeats)
    SHARD = args.shard
    dev = torch.device("cuda", 0)
    net = Net(num_classes=1, C=bench.C, k=bench.K_NBR)
    net.load_state_dict(weights.synth_state_dict(1, bench.C, seed=0), strict=True)
    net = net.to(dev).eval()
    side = 100.0 * (args.points / 10_000_000) ** 0.5
    pc = forest_plot(args.points, side=max(side, 10.0)).to(dev)
    gen = lambda: torch.Generator(device=dev).manual_seed(0)
    pipeline.segment_plot(pc, net, generator=gen())        # allocator
    res = [run_world(pc, net, w, gen) for w in (int(x) for x in args.worlds.split(","))]
    base = res[0]["predicted_s"]
    for r in res:
        r["speedup_vs_1"] = round(base / r["predicted_s"], 3)
        r["efficiency"] = round(base / r["predicted_s"] / r["world"], 3)
        r["plot_points_per_s"] = round(args.points / r["predicted_s"], 1)
    doc = {"what": "predicted strong-scaling curve of BASELINE configs[3] (one plot, all ranks together), every rank's share measured in "
                   "turn on ONE MI355X through pipeline.segment_plot with a recording stand-in for torch.distributed",
           "points": args.points, "xgmi_ring_GBps_assumed": XGMI_LINK_GBPS, "runs_per_rank": REPEATS,
           "shard": SHARD,
           "replicated_per_rank": ("voxelise, the search grid of ALL classified points (inside backproject_s)" if SHARD == "slices" else
                                   "voxelise, the x-order of the plot points and the voxels' x-ranges (inside backproject_s)"),
           "worlds": res}
    txt = json.dumps(doc, indent=1)
    print(txt)
    if args.out:
        open(args.out, "w").write(txt + "\n")


if __name__ == "__main__":
    main()
