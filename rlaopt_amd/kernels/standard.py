"""The five kernel families as operator classes (rlaopt/kernels/standard.py:86-111,
factory.py:9-79): ``RBFLinOp(A1, A2, kernel_config)`` and
``DistributedRBFLinOp(A1, A2, kernel_config, devices, use_full_kernel=True)`` etc.  The
formulas themselves live in ``csrc/rk_kernelmat.hip`` (selected by ``kind``)."""

import torch

from .base import _DistributedKernelLinOp, _KernelLinOp
from .configs import KernelConfig

__all__ = []


def _make(kind: str, display: str):
    class Single(_KernelLinOp):
        def __init__(self, A1: torch.Tensor, A2: torch.Tensor, kernel_config: KernelConfig):
            super().__init__(A1, A2, kernel_config, _kernel_kind=kind)

    class Sharded(_DistributedKernelLinOp):
        def __init__(self, A1, A2, kernel_config, devices, use_full_kernel: bool = True, group=None):
            super().__init__(A1, A2, kernel_config, devices, use_full_kernel, _kernel_kind=kind,
                             group=group)

    Single.__name__ = Single.__qualname__ = f"{display}LinOp"
    Sharded.__name__ = Sharded.__qualname__ = f"Distributed{display}LinOp"
    Single.__doc__ = f"{display} kernel linear operator."
    Sharded.__doc__ = f"Row-sharded {display} kernel linear operator with row and block oracles."
    return Single, Sharded


for _kind, _display in (
    ("rbf", "RBF"), ("laplace", "Laplace"), ("matern12", "Matern12"),
    ("matern32", "Matern32"), ("matern52", "Matern52"),
):
    _single, _sharded = _make(_kind, _display)
    globals()[_single.__name__] = _single
    globals()[_sharded.__name__] = _sharded
    __all__ += [_single.__name__, _sharded.__name__]
