#!/usr/bin/env python3
"""In-kernel phase profile of the grid searches (diagnostic build: P2W_EXTRA_CFLAGS=-DP2W_SLAB_PROFILE).
Prints, per search launch of one forward, the mean cycles a workgroup spent in each phase."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pointstowood_amd import synthetic_weights as weights
from pointstowood_amd import Net
from pointstowood_amd import engine as E
from pointstowood_amd import _lib as _libmod
from pointstowood_amd._lib import lib
import json
JSON_OUT = None
if "--json" in sys.argv:   # also write {kernel class: distance evaluations per forward} (bench.py reads profiles/r6_search_evaluated.json)
    i = sys.argv.index("--json")
    JSON_OUT = sys.argv[i + 1]
    del sys.argv[i:i + 2]
if len(sys.argv) > 1:      # a library built with -DP2W_SLAB_PROFILE (tools/build_variant.sh prof "-DP2W_SLAB_PROFILE")
    _libmod.LIB_PATH = os.path.abspath(sys.argv[1])
evaluated = {}

dev = torch.device("cuda", 0)
net = Net(num_classes=1, C=bench.C, k=bench.K_NBR).to(dev).eval()
net.load_state_dict(weights.synth_state_dict(1, bench.C, seed=0), strict=True)
net = net.to(dev)
data = bench.make_batch(0, dev)
for _ in range(2):
    net(data)
torch.cuda.synchronize()
L = lib()
L.p2w_debug_slab_prof.argtypes = [C.c_void_p, C.c_int]
buf = (C.c_ulonglong * 16)()
L.p2w_debug_slab_prof(buf, 1)
orig = E.Engine._call
names = ["setup", "probe", "plan", "stage", "scan", "check", "ladder", "flush"]

def call(self, name, fn, *args):
    r = orig(self, name, fn, *args)
    if name in ("knn", "knn2", "ball_query"):
        torch.cuda.synchronize()
        L.p2w_debug_slab_prof(buf, 1)
        blocks = max(buf[11], 1)
        evaluated[name] = evaluated.get(name, 0) + 4 * int(buf[13])   # slot 13 counts wave 0's active queries: x 4 waves per workgroup
        print(f"{name:10s} blocks={buf[11]:5d} passes/blk={buf[8]/blocks:.2f} cand/pass={buf[9]/max(buf[8],1):7.1f} "
              f"active/pass={buf[10]/max(buf[8],1):5.1f} flushes/blk={buf[12]/blocks:.2f} | " + " ".join(f"{n}={buf[i]/blocks:7.0f}" for i, n in enumerate(names)))
    return r

E.Engine._call = call
net(data)
if JSON_OUT:
    json.dump({"geom_srchash": bench._geom_hash(), "workload": "BASELINE configs[1] batch 0 (B = 8 x 16384, k = 32)", "what": "candidate-distance evaluations the grid searches "
               "actually perform per forward (sum over passes of staged candidates x active queries of the workgroup: wave 0's count x 4 waves), counted in a "
               "-DP2W_SLAB_PROFILE build", "evaluated_pairs_per_step": evaluated}, open(JSON_OUT, "w"), indent=1)
    print("wrote", JSON_OUT)
