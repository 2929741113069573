"""``Data`` / ``Batch`` / ``DataLoader`` with the attribute contract the reference relies on.

The reference gets these from torch-geometric (``pointstowood/src/predicter.py:10-11,93,177``).
Only the behaviour that path uses is provided:

* ``Data(**tensors)`` stores tensors as attributes; attribute assignment works
  (``data.x = ...``, ``model.py:228``); ``.to(device)`` moves every tensor attribute.
* collation concatenates every tensor attribute along dim 0 after promoting 0-dim tensors
  to shape ``[1]`` (``sf`` -> ``[B]``, ``local_shift [3]`` -> ``[3B]``) and adds
  ``batch`` (``[sum N]`` int64) and ``ptr`` (``[B+1]`` int64) from the ``pos`` row counts.
"""
from __future__ import annotations

import torch


class Data:
    def __init__(self, **kwargs):
        for k, v in kwargs.items():
            setattr(self, k, v)

    def keys(self):
        return [k for k, v in self.__dict__.items() if not k.startswith("_")]

    def tensors(self):
        return {k: v for k, v in self.__dict__.items() if isinstance(v, torch.Tensor)}

    @property
    def num_nodes(self):
        return self.pos.shape[0]

    def to(self, device, non_blocking: bool = False):
        for k, v in self.tensors().items():
            setattr(self, k, v.to(device, non_blocking=non_blocking))
        return self

    def pin_memory(self):
        for k, v in self.tensors().items():
            setattr(self, k, v.pin_memory())
        return self

    def __repr__(self):
        body = ", ".join(f"{k}={list(v.shape)}" for k, v in self.tensors().items())
        return f"{type(self).__name__}({body})"


class Batch(Data):
    @classmethod
    def from_data_list(cls, data_list):
        if not data_list:
            raise ValueError("empty batch")
        out = cls()
        for k in data_list[0].tensors():
            parts = [getattr(d, k) for d in data_list]
            parts = [p.reshape(1) if p.dim() == 0 else p for p in parts]
            setattr(out, k, torch.cat(parts, dim=0))
        n = torch.tensor([d.pos.shape[0] for d in data_list], dtype=torch.long)
        out.batch = torch.repeat_interleave(torch.arange(len(data_list), dtype=torch.long), n)
        out.ptr = torch.cat([torch.zeros(1, dtype=torch.long), torch.cumsum(n, 0)])
        out.num_graphs = len(data_list)
        return out


class DataLoader(torch.utils.data.DataLoader):
    """``torch.utils.data.DataLoader`` that collates ``Data`` objects into a ``Batch``."""

    def __init__(self, dataset, batch_size: int = 1, shuffle: bool = False, **kwargs):
        kwargs.pop("collate_fn", None)
        if "batch_sampler" in kwargs:
            super().__init__(dataset, collate_fn=Batch.from_data_list, **kwargs)
        else:
            super().__init__(dataset, batch_size=batch_size, shuffle=shuffle, collate_fn=Batch.from_data_list, **kwargs)
