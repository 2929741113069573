"""Checkpoint layout of the reference ``Net`` and a recipe for synthetic checkpoints.

Data generator for bench.py, the profiling tools and the tests; no
reference arithmetic lives here.

``key_table`` lists the state-dict keys/shapes that ``Net(num_classes, C)``
(``pointstowood/src/model.py:204-224``) registers - 257 keys for C=32 - derived from
the module tree described there (``MLP()`` :198-202, ``SAModule`` :87-95,
``InvertedResidualBlock`` :46-73, ``DepthwiseSeparableConv1d`` :18-35,
``ReflectanceYesNo`` :155-161).  ``tests/golden/make_golden.py`` asserts the table is
identical (names, order, shapes, dtypes) to the real module's ``state_dict()``.

The trained ``global.pth`` is not available (``.MISSING_LARGE_BLOBS``), so parity is
checked with ``synth_state_dict``: every tensor is a deterministic function of
(key, shape, seed) drawn from numpy's PCG64 stream (stable across machines), with
non-trivial BatchNorm statistics, some negative BN/depthwise scales (exercises
BN-after-ReLU-before-max), and a head gain that gives logits a std of a few units so
that a probability comparison has teeth.
"""
from __future__ import annotations

import zlib
from collections import OrderedDict

import numpy as np
import torch

BN_EPS = 1e-5


def _bn(prefix, c):
    return [(f"{prefix}.weight", (c,), "bn_w"), (f"{prefix}.bias", (c,), "bn_b"),
            (f"{prefix}.running_mean", (c,), "bn_m"), (f"{prefix}.running_var", (c,), "bn_v"),
            (f"{prefix}.num_batches_tracked", (), "bn_n")]


def _mlp(prefix, ch):
    out = []
    for i in range(1, len(ch)):
        out += [(f"{prefix}.{i-1}.0.weight", (ch[i], ch[i - 1]), "lin_w"),
                (f"{prefix}.{i-1}.0.bias", (ch[i],), "lin_b")]
        if i != 1:
            out += _bn(f"{prefix}.{i-1}.2", ch[i])
    return out


def _dsc(prefix, c):
    return ([(f"{prefix}.depthwise_conv.weight", (c, 1, 1), "dw_w"),
             (f"{prefix}.depthwise_conv.bias", (c,), "lin_b")] + _bn(f"{prefix}.depthwise_bn", c) +
            [(f"{prefix}.pointwise_conv.weight", (c, c, 1), "lin_w"),
             (f"{prefix}.pointwise_conv.bias", (c,), "lin_b")] + _bn(f"{prefix}.pointwise_bn", c))


def _resblock(prefix, f):
    e = 4 * f
    return ([(f"{prefix}.expand.0.weight", (e, f, 1), "lin_w"), (f"{prefix}.expand.0.bias", (e,), "lin_b")]
            + _bn(f"{prefix}.expand.1", e)
            + _dsc(f"{prefix}.conv.0", e) + _bn(f"{prefix}.conv.1", e)
            + _dsc(f"{prefix}.conv.3", e) + _bn(f"{prefix}.conv.4", e)
            + [(f"{prefix}.project.0.weight", (f, e, 1), "lin_w"), (f"{prefix}.project.0.bias", (f,), "lin_b")]
            + _bn(f"{prefix}.project.1", f))


def _yesno(prefix, h=32):
    return [(f"{prefix}.fc1.weight", (h, 1), "lin_w"), (f"{prefix}.fc1.bias", (h,), "lin_b"),
            (f"{prefix}.fc2.weight", (h, h), "lin_w"), (f"{prefix}.fc2.bias", (h,), "lin_b"),
            (f"{prefix}.fc3.weight", (1, h), "lin_w"), (f"{prefix}.fc3.bias", (1,), "lin_b")]


def sa_channels(C):
    """local_nn channel lists and residual widths of the three SA levels (model.py:210-212)."""
    return [([C + 4, 2 * C, 4 * C], 4 * C), ([4 * C + 4, 6 * C, 8 * C], 8 * C),
            ([8 * C + 4, 12 * C, 16 * C], 16 * C)]


def fp_channels(C):
    """NN channel lists of fp4..fp1 (model.py:215-218)."""
    return {4: [32 * C, 24 * C, 16 * C], 3: [24 * C, 20 * C, 16 * C],
            2: [20 * C, 16 * C, 16 * C], 1: [17 * C, 16 * C, 16 * C]}


def key_table(num_classes: int = 1, C: int = 32):
    t = _mlp("stem_mlp", [3, C])
    for l, (nn_ch, f) in enumerate(sa_channels(C), start=1):
        t += _mlp(f"sa{l}_module.conv.local_nn", nn_ch)
        t += _resblock(f"sa{l}_module.residual_block", f)
        t += _yesno(f"sa{l}_module.reflectanceyesno")
    t += _mlp("sa4_module.NN", [16 * C + 3, 16 * C, 16 * C])
    for l in (4, 3, 2, 1):
        t += _mlp(f"fp{l}_module.NN", fp_channels(C)[l])
    t += [("conv1.weight", (16 * C, 16 * C, 1), "lin_w"), ("conv1.bias", (16 * C,), "lin_b"),
          ("conv2.weight", (num_classes, 16 * C, 1), "head_w"), ("conv2.bias", (num_classes,), "lin_b")]
    t += _bn("norm", 16 * C)
    return t


def synth_state_dict(num_classes: int = 1, C: int = 32, seed: int = 0, head_gain: float = 25.0,
                     lin_gain: float = 4.0):
    sd = OrderedDict()
    for key, shape, kind in key_table(num_classes, C):
        rng = np.random.Generator(np.random.PCG64((zlib.crc32(key.encode()) + 1000003 * seed) & 0xFFFFFFFF))
        n = int(np.prod(shape)) if shape else 1
        u = rng.random(n)  # float64 in [0,1)
        if kind in ("lin_w", "head_w"):
            fan_in = int(np.prod(shape[1:]))
            bound = np.sqrt(lin_gain / fan_in) * (head_gain if kind == "head_w" else 1.0)
            v = (2 * u - 1) * bound
            if kind == "head_w":  # zero-sum head: cancels the common positive offset of post-ReLU features
                v = v - v.mean()
        elif kind == "lin_b":
            v = (2 * u - 1) * 0.1
        elif kind in ("dw_w", "bn_w"):
            sign = np.where(rng.random(n) < 0.2, -1.0, 1.0)
            v = (0.6 + 0.9 * u) * sign
        elif kind in ("bn_b", "bn_m"):
            v = (2 * u - 1) * 0.2
        elif kind == "bn_v":
            v = 0.5 + u
        elif kind == "bn_n":
            sd[key] = torch.zeros((), dtype=torch.long)
            continue
        else:
            raise KeyError(kind)
        sd[key] = torch.from_numpy(v.astype(np.float32).reshape(shape))
    return sd
