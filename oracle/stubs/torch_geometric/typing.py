"""Shim for the type aliases imported at pointstowood/src/pointnet.py:8-15."""
from typing import Optional, Tuple, Union
from torch import Tensor


class SparseTensor:  # never instantiated on this path
    pass


torch_sparse = None
Adj = Union[Tensor, SparseTensor]
OptTensor = Optional[Tensor]
PairTensor = Tuple[Tensor, Tensor]
PairOptTensor = Tuple[Optional[Tensor], Optional[Tensor]]
