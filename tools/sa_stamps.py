#!/usr/bin/env python3
"""In-kernel phase stamps of the fused PointNetConv kernel (diagnostic build -DP2W_SA_STAMP) on the bench forward:
per level, for waves 0 and 4 of every workgroup: cycles of the whole persistent loop, of the per-slab barrier waits, of the
work between barrier and epilogue (DMA issue + gather + fragment reads + MFMAs + producer) and of the epilogues."""
import os
import statistics
import sys
os.environ.setdefault("P2W_EXTRA_CFLAGS", "-DP2W_SA_STAMP")

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from pointstowood_amd import synthetic_weights as weights  # noqa: E402
from pointstowood_amd import Net  # noqa: E402

dev = torch.device("cuda", 0)
net = Net(1, C=bench.C, k=bench.K_NBR, precision=os.environ.get("P2W_PRECISION", "f16x3"),
          sa_pack=False)   # one class of targets per level: the stamp buffer sits behind its descriptors
net.load_state_dict(weights.synth_state_dict(1, bench.C, seed=0))
net = net.to(dev).eval()
d = bench.make_batch(0, dev)
net(d)
keep = {}
net(d, keep=keep)
torch.cuda.synchronize()
geo = keep["geometry"]
for l in (1, 2, 3):
    M = geo.levels[l].n
    ws = keep[f"sa{l}_module.ws"]
    off = M * 32 * 20 + M * 4 + 64
    st = ws[off: off + 256 * 2 * 8 * 8].clone().view(torch.int64).view(256, 2, 8).cpu()   # (clone: the offset need not be 8-byte aligned)
    used = st[:, 0, 0] > 0
    med = lambda t: statistics.median(t.tolist())
    for w in (0, 1):
        s_ = st[used, w]
        tot, wait, epi, mma, slabs, items, ld = (med(s_[:, i]) for i in range(7))
        print(f"level {l} (M={M}) wave{4*w}: {int(used.sum())} WGs, {items:.0f} items x {slabs/items:.0f} slabs: loop {tot:.0f} cyc "
              f"({tot/slabs:.0f}/slab), barrier wait {100*wait/tot:.0f} %, body {100*mma/tot:.0f} % ({mma/slabs:.0f}/slab), "
              f"end-of-slab load/DMA wait {100*ld/tot:.0f} % ({ld/slabs:.0f}/slab), epilogue {100*(epi-ld)/tot:.0f} % ({(epi-ld)/items:.0f}/item)", flush=True)
