"""Seeded input generation shared by gen_golden.py (which runs the reference) and the tests.

Inputs of the larger fixtures are not stored: they are regenerated with torch's CPU generator,
which is deterministic for a given torch build (the GPU box runs the same image as the build box).
"""

from __future__ import annotations

import torch


def make_data(seed: int, shape: tuple[int, ...], dtype: torch.dtype, kind: str) -> torch.Tensor:
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(*shape, generator=g)
    if kind == "outlier":
        mask = torch.rand(*shape, generator=g) < 0.01
        x = torch.where(mask, x * 25.0, x)
    elif kind == "positive":
        x = x.abs() + 0.25
    return x.to(dtype)


def dtype_from_name(name: str | None) -> torch.dtype | None:
    return None if name is None else getattr(torch, name.removeprefix("torch."))
