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

import contextlib

import numpy as np
import torch


@contextlib.contextmanager
def serial_host_ops():
    """Runs the enclosed CPU tensor operations with ONE intra-op thread and restores the caller's count afterwards.

    The feed side of the path (``TestingDataset.__getitem__``, collation: predicter.py:78-94,177) is a dozen tensor operations on a
    few hundred to 16 384 points.  With the process-wide thread count torch picks on the GPU hosts (128 - 256) each of them that
    enters a parallel region - MKL's vector sqrt, ``repeat_interleave``, reductions above the 32 768-element grain - wakes the
    whole OpenMP team: measured on the MI355X host 2.0 ms per ``torch.sqrt`` of 600 values and 4.3 ms per ``repeat_interleave``
    of 8, i.e. 25 ms per batch of 8 voxels (tools/loop_profile.py), and the spinning team slows the launching thread 4 x.  Results
    do not depend on the thread count (element-wise operations; the reductions here keep one column / row per thread)."""
    n = torch.get_num_threads()
    if n > 1:
        torch.set_num_threads(1)
    try:
        yield
    finally:
        if n > 1:
            torch.set_num_threads(n)


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
        with serial_host_ops():
            for k in data_list[0].tensors():
                parts = [getattr(d, k) for d in data_list]
                parts = [p.reshape(1) if p.dim() == 0 else p for p in parts]
                setattr(out, k, torch.cat(parts, dim=0))
            n = np.array([d.pos.shape[0] for d in data_list], dtype=np.int64)
            # (numpy: integer results, and torch.repeat_interleave runs a parallel region over the B counts)
            out.batch = torch.from_numpy(np.repeat(np.arange(len(data_list), dtype=np.int64), n))
            out.ptr = torch.from_numpy(np.concatenate([np.zeros(1, dtype=np.int64), np.cumsum(n)]))
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
