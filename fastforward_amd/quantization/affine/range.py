"""Range <-> parameter math (reference: src/fastforward/quantization/affine/range.py).

``parameters_for_range`` here keeps the reference's calling convention, including returning
``offset=None`` for the symmetric two-sided case — which, like the reference (:100), needs one
host-visible decision when ``symmetric and allow_one_sided``. The calibration hot path does not
call this function: it uses :func:`fastforward_amd.ops.parameters_for_range`, which takes the same
decision on the device and writes straight into the quantizer's parameters.
"""

from __future__ import annotations

import torch

from fastforward_amd import ops
from fastforward_amd.common import ensure_tensor


def integer_minimum(num_bits: float) -> float:
    """Smallest code of a signed `num_bits` grid: -(2 ** (num_bits - 1)) (reference :9-17)."""
    return -(2 ** (num_bits - 1))


def integer_maximum(num_bits: float) -> float:
    """Largest code of a signed `num_bits` grid (reference :20-28)."""
    return -integer_minimum(num_bits) - 1


def quantization_range(
    scale: torch.Tensor | float, offset: torch.Tensor | float | None, num_bits: float
) -> tuple[torch.Tensor | float, torch.Tensor | float]:
    """(min, max) real values representable with (scale, offset) (reference :31-51)."""
    offset = 0.0 if offset is None else offset
    return (integer_minimum(num_bits) + offset) * scale, (integer_maximum(num_bits) + offset) * scale


def parameters_for_range(
    min_range: torch.Tensor,
    max_range: torch.Tensor,
    num_bits: float,
    symmetric: bool,
    allow_one_sided: bool,
) -> tuple[torch.Tensor, torch.Tensor | None]:
    """(scale, offset) that best cover [min_range, max_range] (reference :54-122)."""
    min_range, max_range = ensure_tensor(min_range), ensure_tensor(max_range)
    if not isinstance(max_range, torch.Tensor) or max_range.device != min_range.device:
        max_range = ensure_tensor(max_range).to(min_range.device)
    shape = min_range.shape
    one_sided = bool(allow_one_sided and symmetric and (min_range.to(torch.float32).min() >= 0))
    two_sided_symmetric = symmetric and not one_sided
    scale, offset = ops.parameters_for_range(
        min_range, max_range, num_bits, symmetric, allow_one_sided, want_offset=not two_sided_symmetric
    )
    scale = scale.reshape(shape)
    if two_sided_symmetric or offset is None:
        return scale, None
    return scale, offset.reshape(shape)
