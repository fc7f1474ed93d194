"""``KernelConfig`` (rlaopt/kernels/configs.py:11-60): lengthscale (float, or a 1-D tensor
with one entry per feature) and a constant multiplier applied to every product."""

from dataclasses import asdict, dataclass
from typing import Any

import torch

from ..utils import _is_float

__all__ = ["KernelConfig"]


@dataclass(kw_only=True, frozen=False)
class KernelConfig:
    const_scaling: float = 1.0
    lengthscale: float | torch.Tensor

    def __post_init__(self):
        _is_float(self.const_scaling, "const_scaling")
        if not isinstance(self.lengthscale, (float, torch.Tensor)):
            raise TypeError(
                f"lengthscale is of type {type(self.lengthscale).__name__}, "
                "but expected type float or torch.Tensor"
            )
        if isinstance(self.lengthscale, torch.Tensor) and self.lengthscale.ndim != 1:
            raise ValueError(
                f"lengthscale has {self.lengthscale.ndim} dimensions, but expected 1 dimension"
            )

    def to_dict(self) -> dict:
        return asdict(self)

    def to(self, device: torch.device) -> "KernelConfig":
        """Copy with a tensor lengthscale moved to ``device`` (self when it is a float)."""
        if isinstance(self.lengthscale, torch.Tensor):
            return KernelConfig(
                const_scaling=self.const_scaling, lengthscale=self.lengthscale.to(device)
            )
        return self


def _is_kernel_config(param: Any, param_name: str):
    if not isinstance(param, KernelConfig):
        raise TypeError(
            f"{param_name} is of type {type(param).__name__}, but expected type KernelConfig"
        )
