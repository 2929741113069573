#!/usr/bin/env python3
"""The GEMM launches of one forward on the bench batch, in launch order: (name, M, K, N, output kind, epilogue reads) and the
bytes one pass over each launch's operands moves.  Written to gpurun_out/gemm_list.json for tools/gemm_traffic.py, which sets
the per-dispatch FETCH_SIZE / WRITE_SIZE counters of a profiled bench run beside them (diagnostic)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from pointstowood_amd import Net, synthetic_weights as weights  # noqa: E402
from pointstowood_amd import engine as eng_mod  # noqa: E402

dev = torch.device("cuda", 0)
net = Net(num_classes=1, C=bench.C, k=bench.K_NBR, precision="f16x3").to(dev).eval()
net.load_state_dict(weights.synth_state_dict(1, bench.C, seed=0), strict=True)
net = net.to(dev)
data = bench.make_batch(0, dev)
net(data)
log = []
E = eng_mod.Engine
orig, orig_dot = E._gemm_h2, getattr(E, "_gemm_h2_rowdot", None)


def spy(self, name, A, ldh_a, M, lin, out_f32=None, ldo=0, out_h2=None, ldh_o=0, residual=None, ldr=0, residual_h=False):
    rd = M * lin.K * 4 + lin.N * lin.K * 4                       # A (hi + lo) + W (hi + lo)
    if residual is not None:
        rd += M * lin.N * 4
    wr = (M * lin.N * 4 if out_f32 is not None else 0) + (M * lin.N * 4 if out_h2 is not None else 0)
    log.append({"name": name, "M": int(M), "K": int(lin.K), "N": int(lin.N), "f32": out_f32 is not None, "h2": out_h2 is not None,
                "residual": residual is not None, "read_bytes": int(rd), "write_bytes": int(wr)})
    return orig(self, name, A, ldh_a, M, lin, out_f32, ldo, out_h2, ldh_o, residual, ldr, residual_h)


E._gemm_h2 = spy
net(data)
torch.cuda.synchronize()
E._gemm_h2 = orig
os.makedirs("gpurun_out", exist_ok=True)
json.dump(log, open("gpurun_out/gemm_list.json", "w"), indent=1)
for r in log:
    print(r)
print(len(log), "launches through _gemm_h2 (the head operator launches its own GEMM)")
