"""Small tensor helpers (reference: src/fastforward/common.py:36-75)."""

from __future__ import annotations

from typing import Any, Callable, Sequence

import torch


def ensure_tensor(value: torch.Tensor | float | Sequence[float], device: torch.device | str | None = None) -> torch.Tensor:
    """Return `value` unchanged if it is a tensor, else ``torch.tensor(value, device=device)``."""
    return value if isinstance(value, torch.Tensor) else torch.tensor(value, device=device)


def tensor_or_none(value: Any, dtype: torch.dtype, device: torch.device | str) -> torch.Tensor | None:
    """None and tensors pass through untouched; scalars become tensors of `dtype` on `device`."""
    if value is None or isinstance(value, torch.Tensor):
        return value
    return torch.tensor(value, dtype=dtype, device=device)


def maybe_tensor_apply(obj: Any, fn: Callable[[torch.Tensor], Any]) -> Any:
    """Apply `fn` when `obj` is a tensor; otherwise hand `obj` back."""
    return fn(obj) if isinstance(obj, torch.Tensor) else obj
