"""Shim for ``torch_geometric.loader.DataLoader`` (pointstowood/src/predicter.py:11,177-180): a torch DataLoader whose
collate function is ``Batch.from_data_list``.  See oracle/stubs/README.md."""
import torch

from .data import Batch


class DataLoader(torch.utils.data.DataLoader):
    def __init__(self, dataset, batch_size=1, shuffle=False, **kw):
        kw.pop("collate_fn", None)
        if "batch_sampler" in kw:
            super().__init__(dataset, collate_fn=Batch.from_data_list, **kw)
        else:
            super().__init__(dataset, batch_size=batch_size, shuffle=shuffle, collate_fn=Batch.from_data_list, **kw)
