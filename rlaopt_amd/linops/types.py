"""``LinOpType`` alias and the operator-or-tensor check (rlaopt/linops/types.py)."""
from typing import Any, Union

import torch

from .base import _BaseLinOp

__all__ = ["LinOpType", "_is_linop_or_torch_tensor"]

LinOpType = Union[
    "LinOp", "TwoSidedLinOp", "SymmetricLinOp",
    "DistributedLinOp", "DistributedTwoSidedLinOp", "DistributedSymmetricLinOp",
]


def _is_linop_or_torch_tensor(param: Any, param_name: str):
    if not isinstance(param, (_BaseLinOp, torch.Tensor)):
        raise TypeError(
            f"{param_name} is of type {type(param).__name__}, "
            "but expected type LinOpType or torch.Tensor"
        )
