"""Argument validators with the reference's names, messages and exception types
(rlaopt/utils/input_checkers.py:27-147): TypeError for a wrong type, ValueError for a
wrong value.  Built from two generic helpers instead of one hand-written body each."""

from typing import Any

import torch

__all__ = [
    "_is_bool", "_is_callable", "_is_dict", "_is_float", "_is_list", "_is_set", "_is_str",
    "_is_torch_device", "_is_torch_f32_f64", "_is_torch_size", "_is_torch_tensor",
    "_is_torch_tensor_1d_2d", "_is_nonneg_float", "_is_pos_float", "_is_pos_int",
]


def _expect_type(param: Any, name: str, typ, label: str) -> None:
    if not isinstance(param, typ):
        raise TypeError(f"{name} is of type {type(param).__name__}, but expected type {label}")


def _type_checker(typ, label):
    def check(param: Any, param_name: str) -> None:
        _expect_type(param, param_name, typ, label)

    check.__doc__ = f"Raise TypeError unless the argument is a {label}."
    return check


_is_bool = _type_checker(bool, "bool")
_is_dict = _type_checker(dict, "dict")
_is_float = _type_checker(float, "float")
_is_int = _type_checker(int, "int")
_is_list = _type_checker(list, "list")
_is_set = _type_checker(set, "set")
_is_str = _type_checker(str, "str")
_is_torch_device = _type_checker(torch.device, "torch.device")
_is_torch_dtype = _type_checker(torch.dtype, "torch.dtype")
_is_torch_size = _type_checker(torch.Size, "torch.Size")
_is_torch_tensor = _type_checker(torch.Tensor, "torch.Tensor")


def _is_callable(param: Any, param_name: str) -> None:
    if not callable(param):
        raise TypeError(
            f"{param_name} is of type {type(param).__name__}, but expected type callable"
        )


def _is_torch_f32_f64(param: Any, param_name: str) -> None:
    _is_torch_dtype(param, param_name)
    if param not in (torch.float32, torch.float64):
        raise ValueError(f"{param_name} is {param}, but expected torch.float32 or torch.float64")


def _is_torch_tensor_1d_2d(param: Any, param_name: str) -> None:
    _is_torch_tensor(param, param_name)
    if param.ndim not in (1, 2):
        raise ValueError(f"{param_name} must be a 1D or 2D tensor. Received {param.ndim}D tensor.")


def _is_nonneg_float(param: Any, param_name: str) -> None:
    _is_float(param, param_name)
    if param < 0:
        raise ValueError(f"{param_name} must be non-negative, but received {param}")


def _is_pos_float(param: Any, param_name: str) -> None:
    _is_float(param, param_name)
    if param <= 0:
        raise ValueError(f"{param_name} must be positive, but received {param}")


def _is_pos_int(param: Any, param_name: str) -> None:
    _is_int(param, param_name)
    if param <= 0:
        raise ValueError(f"{param_name} must be positive, but received {param}")
