"""Functional fp32 CPU restatement of the reference forward.  TEST INFRASTRUCTURE.

``forward(sd, pos, batch, reflectance, sf)`` follows ``Net.forward``
(``pointstowood/src/model.py:226-245``) in eval mode, statement by statement, over a
plain state dict (layout: ``pointstowood_amd/synthetic_weights.py``) and ``oracle.ops``.  It is checked
against the reference's own modules (imported over ``oracle/stubs``) by
``tests/test_oracle_golden.py`` via the vectors in ``tests/golden``.

Two reference behaviours are restated by their *effect*:
* ``ReflectanceYesNo`` (model.py:155-175, called at :110-112) multiplies the
  reflectance column by a hard gumbel-softmax over a size-1 axis, which is exactly
  1.0 for every input, so the multiplication is omitted (x * 1.0 == x bit for bit);
* ``torch.cuda.amp.autocast`` (predicter.py:197) is a no-op on the CPU path.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import ops as _oracle_ops
BN_EPS = 1e-5   # torch.nn.BatchNorm1d default (model.py builds every BN with it)

ops = _oracle_ops   # the operator module the functions below call; ``forward(..., ops=module)`` swaps it for one call


def _bn(sd, p, x):
    """BatchNorm1d in eval mode on [rows, C]."""
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"],
                        sd[p + ".weight"], sd[p + ".bias"], False, 0.0, BN_EPS)


def _mlp(sd, p, x, n_layers):
    """``MLP()`` (model.py:198-202): layer 0 = Lin+ReLU, later layers = Lin+ReLU+BN."""
    for i in range(n_layers):
        x = F.relu(F.linear(x, sd[f"{p}.{i}.0.weight"], sd[f"{p}.{i}.0.bias"]))
        if i != 0:
            x = _bn(sd, f"{p}.{i}.2", x)
    return x


def _conv1x1(sd, p, x):
    """Conv1d(kernel_size=1) on a [1,C,M] view of [M,C] == a Linear over channels."""
    return F.linear(x, sd[p + ".weight"][:, :, 0], sd[p + ".bias"])


def _dsc(sd, p, x):
    """``DepthwiseSeparableConv1d.forward`` (model.py:37-44) with kernel_size=1."""
    x = x * sd[p + ".depthwise_conv.weight"][:, 0, 0] + sd[p + ".depthwise_conv.bias"]
    x = F.relu(_bn(sd, p + ".depthwise_bn", x))
    x = _conv1x1(sd, p + ".pointwise_conv", x)
    return F.relu(_bn(sd, p + ".pointwise_bn", x))


def _resblock(sd, p, x):
    """``InvertedResidualBlock.forward`` (model.py:75-85); shortcut is the identity."""
    out = F.relu(_bn(sd, p + ".expand.1", _conv1x1(sd, p + ".expand.0", x)))
    out = _dsc(sd, p + ".conv.0", out)
    out = F.relu(_bn(sd, p + ".conv.1", out))
    out = _dsc(sd, p + ".conv.3", out)
    out = _bn(sd, p + ".conv.4", out)
    out = _bn(sd, p + ".project.1", _conv1x1(sd, p + ".project.0", out))
    return F.relu(out + x)


def _pointnet_conv(sd, p, x, pos_src, pos_dst, src, dst):
    """``PointNetConv.message`` + max aggregation (pointnet.py:86-132)."""
    pos_j, pos_i = pos_src[src], pos_dst[dst]
    rel = pos_j[:, :3] - pos_i[:, :3]
    nrm = torch.norm(rel, dim=1, keepdim=True)
    dmax, _ = ops.scatter_max(nrm, dst, dim=0, dim_size=pos_dst.shape[0])
    msg = torch.zeros((pos_j.shape[0], pos_j.shape[1]), device=pos_j.device)
    msg[:, :3] = rel / (dmax[dst] + 1e-8)
    msg[:, 3] = pos_j[:, 3]
    msg = torch.cat([x[src], msg], dim=1)
    msg = _mlp(sd, p + ".local_nn", msg, 2)
    if hasattr(ops, "segment_max_rows"):
        return ops.segment_max_rows(msg, dst, pos_dst.shape[0])
    return ops.scatter_max(msg, dst, dim=0, dim_size=pos_dst.shape[0])[0]   # = MessagePassing.propagate's aggr='max'


def _sa(sd, p, resolution, k, x, pos, batch, reflectance, sf, cap):
    """``SAModule.forward`` (model.py:108-127), eval branch."""
    pos = torch.cat([pos[:, :3], reflectance.unsqueeze(-1)], dim=-1)
    idx = ops.consecutive_cluster(ops.voxel_grid(pos[:, :3], resolution, batch))[1]
    if resolution == 0.04:
        row, col = ops.radius(pos[:, :3], pos[idx, :3], resolution * 2, batch, batch[idx], max_num_neighbors=k)
    else:
        row, col = ops.knn(pos[:, :3], pos[idx, :3], k, batch_x=batch, batch_y=batch[idx])
    pos[:, :3] = pos[:, :3] / sf[batch].unsqueeze(-1)
    x = _pointnet_conv(sd, p + ".conv", x, pos, pos[idx], col, row)
    pos[:, :3] = pos[:, :3] * sf[batch].unsqueeze(-1)
    if cap is not None:
        cap[p + ".idx"] = idx
        cap[p + ".batch"] = batch[idx]
        cap[p + ".edge_q"], cap[p + ".edge_c"] = row, col
        cap[p + ".conv"] = x
    x = _resblock(sd, p + ".residual_block", x)
    if cap is not None:
        cap[p + ".out"] = x
    return x, pos[idx, :3], batch[idx], reflectance[idx], sf


def _fp(sd, p, k, x, pos, batch, x_skip, pos_skip, batch_skip, cap):
    """``FPModule.forward`` (model.py:148-153)."""
    x = ops.knn_interpolate(x, pos, pos_skip, batch, batch_skip, k=k)
    x = torch.cat([x, x_skip], dim=1)
    x = _mlp(sd, p + ".NN", x, 2)
    if cap is not None:
        cap[p + ".out"] = x
    return x, pos_skip, batch_skip


@torch.no_grad()
def forward(sd, pos, batch, reflectance, sf, k: int = 32, capture: dict | None = None, ops=None):
    """logits [sum N] fp32.  ``k`` overrides ``SAModule.k`` (hard-coded 32 at model.py:210-212).
    ``ops``: another module with the operators' call signatures (``pointstowood_amd.ops`` on GPU tensors: the composition
    test of INTEGRATION.md's route B); default = ``oracle.ops``."""
    if ops is not None:
        global_ops = globals()["ops"]
        globals()["ops"] = ops
        try:
            return forward(sd, pos, batch, reflectance, sf, k=k, capture=capture)
        finally:
            globals()["ops"] = global_ops
    cap = capture
    x0 = _mlp(sd, "stem_mlp", pos[:, :3], 1)
    sa0 = (x0, pos, batch, reflectance, sf)
    sa1 = _sa(sd, "sa1_module", 0.04, k, *sa0, cap)
    sa2 = _sa(sd, "sa2_module", 0.08, k, *sa1, cap)
    sa3 = _sa(sd, "sa3_module", 0.16, k, *sa2, cap)
    # GlobalSAModule.forward (model.py:134-140)
    x4 = globals()["ops"].global_max_pool(_mlp(sd, "sa4_module.NN", torch.cat([sa3[0], sa3[1]], dim=1), 2), sa3[2])
    nb = x4.shape[0]
    sa4 = (x4, torch.zeros((nb, 3), device=pos.device), torch.arange(nb, device=pos.device))
    if cap is not None:
        cap["stem"] = x0
        cap["sa4_module.out"] = x4
    fp4 = _fp(sd, "fp4_module", 2, *sa4, *sa3[:3], cap)
    fp3 = _fp(sd, "fp3_module", 2, *fp4, *sa2[:3], cap)
    fp2 = _fp(sd, "fp2_module", 2, *fp3, *sa1[:3], cap)
    x, _, _ = _fp(sd, "fp1_module", 2, *fp2, *sa0[:3], cap)
    # head (model.py:241-243)
    x = F.relu(_bn(sd, "norm", _conv1x1(sd, "conv1", x)))
    x = _conv1x1(sd, "conv2", x)
    return torch.squeeze(x.t()).to(torch.float)
