#!/usr/bin/env python3
"""The FP modules' launches of one sequential bench forward under a variant library (tools/build_variant.sh NAME
"-DP2W_INTERP_DEPTH=n" feat) and an fp_hoist threshold: per-launch HIP-event times (median of 7) of the gemm_mlp class and the
interpolation class.   python tools/interp_epi_ab.py [lib.so] [key=value ...]"""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointstowood_amd import _lib as _libmod
args = sys.argv[1:]
if args and args[0].endswith(".so"):
    _libmod.LIB_PATH = os.path.abspath(args.pop(0))
    from pointstowood_amd import build as _b
    _b._stale = lambda: False
import torch
import bench
from pointstowood_amd import Net, synthetic_weights as weights
from pointstowood_amd import engine as eng_mod
dev = torch.device("cuda")
data = bench.make_batch(0, dev, 0)
E = eng_mod.Engine
orig = E._gemm_h2
shapes = []


def spy(self, name, A, ldh_a, M, lin, *a, **kw):
    if self.events is not None:
        shapes.append((name, int(M), int(lin.K), int(lin.N), "interp" if kw.get("interp") is not None else ""))
    return orig(self, name, A, ldh_a, M, lin, *a, **kw)


E._gemm_h2 = spy
net = Net(num_classes=1, C=32, k=32, **bench.engine_options(args))
net.load_state_dict(weights.synth_state_dict(1, 32, seed=0), strict=True)
net = net.to(dev).eval()
net(data); net(data)
eng = net._engine
runs, iruns = [], []
for rep in range(7):
    shapes.clear()
    eng.events, eng.events_grouped = [], False
    net(data)
    torch.cuda.synchronize()
    ev, eng.events = eng.events, None
    runs.append([s.elapsed_time(e) * 1e3 for n, s, e in ev if n in ("gemm_hoist", "gemm_res", "gemm_mlp")])
    iruns.append(sum(s.elapsed_time(e) * 1e3 for n, s, e in ev if n == "interp_concat"))
med = [statistics.median(r[i] for r in runs) for i in range(len(runs[0]))]
tot = 0.0
for i, t in enumerate(med):
    name, M, K, N, tag = shapes[i] if i < len(shapes) else ("head", 0, 512, 512, "")
    if name == "gemm_mlp" or i >= len(shapes):
        tot += t
        print(f"{name:9s} {M:7d} {K:5d} {N:5d} {tag:7s} {t:8.1f} us")
print(f"{os.path.basename(_libmod.LIB_PATH)} {' '.join(args)}: gemm_mlp + head {tot:.1f} us, interpolation class {statistics.median(iruns):.1f} us, "
      f"all GEMMs {sum(med):.1f} us")
