"""Shim for ``MessagePassing`` (imported at pointstowood/src/pointnet.py:6).

Only what PointNetConv uses: ``propagate(edge_index, x=(X, None), pos=(Psrc, Pdst))``
gathers ``x_j``, ``pos_j`` by the source row and ``pos_i`` by the target row of
``edge_index``, calls ``self.message`` and max-aggregates onto ``Pdst.shape[0]`` rows.
"""
import torch
from oracle.ops import segment_max_rows


class MessagePassing(torch.nn.Module):
    def __init__(self, aggr="max", **kwargs):
        super().__init__()
        assert aggr == "max"
        self.aggr = aggr

    def reset_parameters(self):
        pass

    def propagate(self, edge_index, x, pos):
        src, dst = edge_index[0], edge_index[1]
        x_src = x[0] if isinstance(x, tuple) else x
        x_j = None if x_src is None else x_src[src]
        msg = self.message(x_j=x_j, pos_i=pos[1][dst], pos_j=pos[0][src], edge_index_i=dst)
        return segment_max_rows(msg, dst, pos[1].shape[0])
