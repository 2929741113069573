"""Shim for ``torch_geometric.data`` (pointstowood/src/predicter.py:10): ``Data`` / ``Batch`` / ``Dataset`` with the
collation contract SURVEY.md Appendix A states for PyG (concat every tensor attribute along dim 0 after promoting
0-dim tensors to [1]; add ``batch`` and ``ptr`` from the ``pos`` row counts).  Written independently of
``pointstowood_amd.data`` so that fixtures generated through it cross-check the product's collation.
See oracle/stubs/README.md."""
import torch


class Data:
    def __init__(self, **kw):
        self.__dict__.update(kw)

    def _tensors(self):
        return [(k, v) for k, v in self.__dict__.items() if torch.is_tensor(v)]

    def to(self, device, *a, **kw):
        for k, v in self._tensors():
            self.__dict__[k] = v.to(device)
        return self

    def pin_memory(self):
        return self


class Batch(Data):
    @staticmethod
    def from_data_list(items):
        out = Batch()
        for key, _ in items[0]._tensors():
            vals = [getattr(d, key) for d in items]
            out.__dict__[key] = torch.cat([v.reshape(1) if v.dim() == 0 else v for v in vals], dim=0)
        counts = torch.tensor([d.pos.shape[0] for d in items], dtype=torch.long)
        out.batch = torch.arange(len(items), dtype=torch.long).repeat_interleave(counts)
        out.ptr = torch.cat([counts.new_zeros(1), counts.cumsum(0)])
        return out


class Dataset(torch.utils.data.Dataset):
    pass
