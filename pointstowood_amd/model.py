"""``Net`` - drop-in for the reference's ``src.model.Net`` on MI355X.

Same constructor (``Net(num_classes, C=32)``), same 257-key ``state_dict`` (names, order,
shapes - so ``global.pth`` loads unchanged through ``load_state_dict(..., strict=False)``,
reference ``pointstowood/src/predicter.py:97-105``), same ``forward(data) -> [sum N] fp32
logits`` over a duck-typed batch with ``pos``, ``batch``, ``reflectance``, ``sf``
(``pointstowood/src/model.py:226-245``).  The module tree is generated from a key table
instead of hand-written layer classes because no PyTorch layer is ever executed: the
forward runs ``pointstowood_amd.engine.Engine`` over weights folded and packed for the
HIP kernels.  Inference only (eval semantics; no autograd graph is built).
"""
from __future__ import annotations

import math

import torch

from . import _lib
from .engine import Engine, EngineOptions, PackedWeights


def _bn(prefix, c):
    return [(f"{prefix}.weight", (c,), "ones"), (f"{prefix}.bias", (c,), "zeros"),
            (f"{prefix}.running_mean", (c,), "buf0"), (f"{prefix}.running_var", (c,), "buf1"),
            (f"{prefix}.num_batches_tracked", (), "bufn")]


def _lin(prefix, o, i, conv=False):
    return [(f"{prefix}.weight", (o, i, 1) if conv else (o, i), "conv" if conv else "lin"),
            (f"{prefix}.bias", (o,), "zeros")]


def _mlp(prefix, ch):
    keys = []
    for i in range(1, len(ch)):
        keys += _lin(f"{prefix}.{i-1}.0", ch[i], ch[i - 1])
        if i != 1:
            keys += _bn(f"{prefix}.{i-1}.2", ch[i])
    return keys


def _dsc(prefix, c):
    return ([(f"{prefix}.depthwise_conv.weight", (c, 1, 1), "conv"), (f"{prefix}.depthwise_conv.bias", (c,), "zeros")]
            + _bn(f"{prefix}.depthwise_bn", c) + _lin(f"{prefix}.pointwise_conv", c, c, conv=True)
            + _bn(f"{prefix}.pointwise_bn", c))


def checkpoint_layout(num_classes: int = 1, C: int = 32):
    """(key, shape, kind) for every entry of the reference ``Net.state_dict()``, in its order."""
    keys = _mlp("stem_mlp", [3, C])
    f_in = C
    for l, f in ((1, 4 * C), (2, 8 * C), (3, 16 * C)):
        p = f"sa{l}_module"
        keys += _mlp(p + ".conv.local_nn", [f_in + 4, {1: 2, 2: 6, 3: 12}[l] * C, f])
        e = 4 * f
        r = p + ".residual_block"
        keys += _lin(r + ".expand.0", e, f, conv=True) + _bn(r + ".expand.1", e)
        keys += _dsc(r + ".conv.0", e) + _bn(r + ".conv.1", e) + _dsc(r + ".conv.3", e) + _bn(r + ".conv.4", e)
        keys += _lin(r + ".project.0", f, e, conv=True) + _bn(r + ".project.1", f)
        y = p + ".reflectanceyesno"
        keys += _lin(y + ".fc1", 32, 1) + _lin(y + ".fc2", 32, 32) + _lin(y + ".fc3", 1, 32)
        f_in = f
    keys += _mlp("sa4_module.NN", [16 * C + 3, 16 * C, 16 * C])
    for l, ch in ((4, [32, 24, 16]), (3, [24, 20, 16]), (2, [20, 16, 16]), (1, [17, 16, 16])):
        keys += _mlp(f"fp{l}_module.NN", [c * C for c in ch])
    keys += _lin("conv1", 16 * C, 16 * C, conv=True) + _lin("conv2", num_classes, 16 * C, conv=True)
    keys += _bn("norm", 16 * C)
    return keys


class _Node(torch.nn.Module):
    """Anonymous container: only carries parameters/buffers under the reference's names."""


class Net(torch.nn.Module):
    def __init__(self, num_classes: int, C: int = 32, k: int = 32, precision: str = "f16x3", **engine_options):
        """``num_classes`` / ``C`` as the reference's ``Net``; ``k`` = neighbours per SA level (the reference hard-codes 32);
        ``precision`` see below; ``engine_options``: fields of ``engine.EngineOptions`` (sampler=, search=, sa_pack=, ...),
        the behaviour switches of the forward - keyword arguments, not environment variables."""
        super().__init__()
        self.num_classes, self.C, self.k = int(num_classes), int(C), int(k)
        self.engine_options = EngineOptions(**engine_options)
        # "f16x3": split-fp16 MFMA, fp32 accumulate (parity default); "fp32": fp32 MFMA; "fp16" / "bf16": one MFMA per
        # product like the reference's torch.cuda.amp.autocast path (predicter.py:197) - faster, NOT within 1e-4
        self.precision = precision
        for key, shape, kind in checkpoint_layout(self.num_classes, self.C):
            *path, leaf = key.split(".")
            node = self
            for part in path:
                if part not in node._modules:
                    node.add_module(part, _Node())
                node = node._modules[part]
            if kind.startswith("buf"):
                t = (torch.zeros(shape) if kind == "buf0" else torch.ones(shape) if kind == "buf1"
                     else torch.zeros((), dtype=torch.long))
                node.register_buffer(leaf, t)
            else:
                node.register_parameter(leaf, torch.nn.Parameter(self._init(shape, kind), requires_grad=False))
        self._packed = None
        self._engine = None
        self._fb_engine = None   # fp32 engine of the range guard's fallback (built on first use)
        self.eval()

    @staticmethod
    def _init(shape, kind):
        """Same distributions as the reference's ``initialize_weights`` (model.py:9-16)."""
        if kind == "ones":
            return torch.ones(shape)
        if kind == "zeros":
            return torch.zeros(shape)
        t = torch.empty(shape)
        fan_in = int(shape[1] * (shape[2] if len(shape) == 3 else 1))
        if kind == "lin":   # xavier uniform
            bound = math.sqrt(6.0 / (fan_in + shape[0]))
        else:               # Conv1d: kaiming uniform, fan_in, relu
            bound = math.sqrt(2.0) * math.sqrt(3.0 / fan_in)
        return t.uniform_(-bound, bound)

    # -- weight packing -------------------------------------------------------------------------
    def _apply(self, fn, *a, **kw):
        self._packed = self._fb_engine = None
        return super()._apply(fn, *a, **kw)

    def load_state_dict(self, *a, **kw):
        self._packed = self._fb_engine = None
        return super().load_state_dict(*a, **kw)

    def repack(self):
        """Call after mutating parameters in place."""
        self._packed = self._fb_engine = None

    def _range_fallback(self, geo, keep):
        """The feature phase of `geo` on the fp32 MFMA path: what the range guard (EngineOptions.range_guard) runs instead when a
        layer's activations do not fit the 16-bit planes of f16x3 / fp16 / bf16 (a checkpoint whose BatchNorm scales push them
        beyond +-6e4 or under fp16's subnormal floor).  The geometry is precision-independent and is reused."""
        if getattr(geo, "rows0_sorted", False):
            raise RuntimeError("range fallback: not available with fp1_cell_order=True (the fp32 path keeps level 0 in input order)")
        dev = geo.sf.device
        if self._fb_engine is None or self._fb_engine.w.device != dev:
            packed = PackedWeights(self.state_dict(), self.C, self.num_classes, dev, "fp32")
            self._fb_engine = Engine(packed, k=self.k, precision="fp32", options=self.engine_options)
        return self._fb_engine.features(geo, keep)

    def _ensure_packed(self, device):
        if self._packed is None or self._packed.device != device or self._packed.precision != self.precision:
            self._packed = PackedWeights(self.state_dict(), self.C, self.num_classes, device, self.precision)
            self._engine = Engine(self._packed, k=self.k, precision=self.precision, options=self.engine_options)
        if self._engine.k != self.k or self._engine.precision != self.precision or self._engine.options is not self.engine_options:
            self._engine = Engine(self._packed, k=self.k, precision=self.precision, options=self.engine_options)
        self._engine.fallback = self._range_fallback if self.precision != "fp32" else None
        return self._engine

    # -- forward --------------------------------------------------------------------------------
    def _inputs(self, data):
        pos, batch, refl, sf = data.pos, data.batch, data.reflectance, data.sf
        _lib.require_cuda(pos, batch, refl, sf)
        _lib.lib()
        dev = pos.device
        B = int(sf.numel())
        pos = pos.to(torch.float32)
        if pos.stride(1) != 1:
            pos = pos.contiguous()
        refl = refl.to(torch.float32).contiguous()
        sf = sf.to(torch.float32).reshape(-1).contiguous()
        p = getattr(data, "ptr", None)
        if p is not None and p.numel() == B + 1:
            ptr0 = p.to(device=dev, dtype=torch.int32).contiguous()
        else:  # sorted batch vector -> CSR, on the device, no sync
            ptr0 = torch.searchsorted(batch.to(torch.int64).contiguous(),
                                      torch.arange(B + 1, device=dev, dtype=torch.int64)).to(torch.int32)
        return dev, (pos, refl, ptr0, sf)

    @torch.no_grad()
    def forward(self, data, keep: dict | None = None):
        if self.training:
            raise RuntimeError("pointstowood_amd.Net is inference-only: call .eval()")
        dev, args = self._inputs(data)
        eng = self._ensure_packed(dev)
        logits = eng.forward(*args, keep=keep)
        data.x = eng.stem_out  # the reference stores the stem features on the batch (model.py:228)
        return logits

    @torch.no_grad()
    def stream(self, batches):
        """Pipelined inference over an iterable of batches (already on the GPU): yields one logits tensor per batch,
        in order, with the geometry phase of batch i+1 overlapped with the feature phase of batch i (two HIP streams).
        Same results as calling the module once per batch."""
        if self.training:
            raise RuntimeError("pointstowood_amd.Net is inference-only: call .eval()")
        it = iter(batches)
        first = next(it, None)
        if first is None:
            return
        dev, args0 = self._inputs(first)
        eng = self._ensure_packed(dev)

        def inputs():
            yield args0
            for d in it:
                yield self._inputs(d)[1]
        yield from eng.forward_stream(inputs())
