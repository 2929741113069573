#!/usr/bin/env python3
"""Emulation (CPU, PyTorch) of cheaper forms of f16x3's two cross terms — the round-4 verdict's "experiment before any kernel work".

f16x3 computes  a·w = a_hi·w_hi + (a_lo·w_hi + a_hi·w_lo)  with three fp16 MFMAs per product.  The candidates compute the
parenthesis from narrower operands on the block-scaled `v_mfma_scale_f32_16x16x128_f8f6f4` (fp8: 2x, fp6: 4x the fp16 rate) or on
the int8 MFMA (2x), i.e. 2 / 1.5 MFMA-equivalents per product instead of 3:

  fp8   : cross = q8(a_lo)·q8(w_hi) + q8(a_hi)·q8(w_lo),   q8 = e4m3fn with one power-of-two scale per 32 K-elements
  fp6   : the same with e2m3 (3 mantissa bits, 2 exponent bits), scale per 32 K-elements
  int8  : the same with int8 and one scale per row (of a) / per output column (of w)
  f16x2 : cross dropped (what plain fp16 with fp16 hi/lo activations but hi-only products would give)

Every `F.linear` of `oracle/net.py` is replaced by the emulated product (fp64 accumulation, so what is measured is the operand
rounding alone); the six golden cases and 12 fuzz batches are run, max |logit - fp32 logit| is printed per candidate.
Runs without a GPU:   python tools/cross_term_emulate.py
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn.functional as F
from oracle import net as onet
from pointstowood_amd import synthetic_weights as weights, synthetic_voxels as synth
from tests import golden_util as G

torch.set_num_threads(8)
real_linear = F.linear


def split16(x):
    hi = x.to(torch.float16).to(torch.float32)
    lo = (x - hi).to(torch.float16).to(torch.float32)
    return hi, lo


def blocks(x, blk=32):
    """[R, K] -> [R, K/blk, blk] (K zero padded)."""
    R, K = x.shape
    pad = (-K) % blk
    if pad:
        x = F.pad(x, (0, pad))
    return x.reshape(R, -1, blk)


def q_block(x, fmt):
    """Quantise along K with one power-of-two scale per 32 elements (MX style): returns the de-quantised fp32 values."""
    R, K = x.shape
    b = blocks(x)
    amax = b.abs().amax(dim=2, keepdim=True).clamp_min(1e-38)
    if fmt == "e4m3":
        top = 448.0
    elif fmt == "e2m3":
        top = 7.5
    e = torch.ceil(torch.log2(amax / top))          # smallest power of two with amax / 2^e <= top
    s = torch.exp2(e)
    v = b / s
    if fmt == "e4m3":
        q = v.to(torch.float8_e4m3fn).to(torch.float32)
    else:   # e2m3: sign, 2 exponent bits (bias 1), 3 mantissa bits: normals 1.0 .. 7.5 in steps of 2^(E-3), subnormals k/8
        a = v.abs()
        ex = torch.floor(torch.log2(a.clamp_min(1e-30))).clamp(0, 2)
        step = torch.exp2(ex - 3)
        q = torch.round(a / step) * step
        q = torch.sign(v) * q.clamp_max(7.5)
    return (q * s).reshape(R, -1)[:, :K]


def q_int8(x):
    amax = x.abs().amax(dim=1, keepdim=True).clamp_min(1e-38)
    s = amax / 127.0
    return torch.round(x / s).clamp(-127, 127) * s


def mm(a, w):
    return (a.double() @ w.double().t())


def make_linear(mode):
    def lin(x, w, b=None):
        if x.shape[1] < 8:     # the 3-wide stem / relative-position columns run on the VALU in fp32 in the product
            return real_linear(x, w, b)
        ah, al = split16(x)
        wh, wl = split16(w)
        if mode == "fp32":
            out = mm(x, w)
        elif mode == "fp16":
            out = mm(ah, wh)
        elif mode == "f16x3":
            out = mm(ah, wh) + mm(al, wh) + mm(ah, wl)
        elif mode == "f16x2a":   # activations carried as hi/lo, weights hi only
            out = mm(ah, wh) + mm(al, wh)
        elif mode in ("fp8", "fp6"):
            f = "e4m3" if mode == "fp8" else "e2m3"
            out = mm(ah, wh) + mm(q_block(al, f), q_block(wh, f)) + mm(q_block(ah, f), q_block(wl, f))
        elif mode == "int8":
            out = mm(ah, wh) + mm(q_int8(al), q_int8(wh)) + mm(q_int8(ah), q_int8(wl))
        else:
            raise ValueError(mode)
        out = out.float()
        return out if b is None else out + b
    return lin


def cases():
    for name in G.CASES[:5]:
        g, inp, meta = G.load(name)
        sd = weights.synth_state_dict(1, meta["C"], seed=meta["wseed"])
        yield name, sd, inp["pos"], inp["batch"], inp["reflectance"], inp["sf"], meta["k"]
    sd = weights.synth_state_dict(1, 32, seed=0)
    for s in range(6):
        vox = [synth.uniform_voxel(2.0, 1500 + 300 * i, 700 + 10 * s + i, bool(s & 1)) for i in range(2)]
        pos = torch.cat([torch.as_tensor(v["pos"]) for v in vox]).float()
        batch = torch.cat([torch.full((len(v["pos"]),), i, dtype=torch.long) for i, v in enumerate(vox)])
        refl = torch.cat([torch.as_tensor(v["reflectance"]) for v in vox]).float()
        sf = torch.stack([torch.as_tensor(v["sf"]).reshape(()) for v in vox]).float()
        yield f"fuzz{s}", sd, pos, batch, refl, sf, 32


def main():
    modes = ["fp32", "f16x3", "fp8", "fp6", "int8", "f16x2a", "fp16"]
    worst = {m: 0.0 for m in modes}
    for name, sd, pos, batch, refl, sf, k in cases():
        F.linear = real_linear
        ref = onet.forward(sd, pos.clone(), batch, refl, sf, k=k).double()
        row = []
        for m in modes:
            F.linear = make_linear(m)
            try:
                out = onet.forward(sd, pos.clone(), batch, refl, sf, k=k).double()
            finally:
                F.linear = real_linear
            err = float((out - ref).abs().max())
            worst[m] = max(worst[m], err)
            row.append(f"{m} {err:.2e}")
        print(f"{name:22s} " + "  ".join(row), flush=True)
    print("worst max|dlogit| vs the fp32 oracle (limit of the parity tests: 4e-4; verdict's build threshold: 2e-4):")
    for m in modes:
        print(f"  {m:7s} {worst[m]:.3e}")


if __name__ == "__main__":
    main()
