"""Autograd wrappers around the hot-path ops (reference: src/fastforward/quantization/affine/_autograd.py).

Gradient semantics as in the reference (:1-16): all gradient approximation lives in the quantize
functions; dequantize and dynamic-quantize pass gradients straight through.
These three Functions are the only callers of the four ops (reference call sites :86,99,121,148), and they call
them where the reference does: through the torch operator registry (``torch.ops.fastforward_amd.*``, registered in
fastforward_amd/ops.py with the reference's schemas), so a kernel registered for another dispatch key, a FakeTensor
trace or ``torch.library.opcheck`` sees exactly the calls the product makes.
"""

from __future__ import annotations

from typing import Any, Literal

import torch

from fastforward_amd import ops  # noqa: F401  (defines torch.ops.fastforward_amd.*)
from fastforward_amd.common import tensor_or_none

_OPS = torch.ops.fastforward_amd


def _float_dtype_of(data: torch.Tensor) -> torch.dtype:
    return data.dtype if data.dtype.is_floating_point else torch.get_default_dtype()


def quantize_affine(
    data: torch.Tensor,
    scale: float | torch.Tensor,
    offset: float | torch.Tensor | None,
    tile_size: torch.Size | Literal["data_shape"],
    num_bits: int,
    quantized_dtype: torch.dtype | None,
) -> torch.Tensor:
    dtype = _float_dtype_of(data)
    scale = tensor_or_none(scale, dtype=dtype, device=data.device)
    offset = tensor_or_none(offset, dtype=dtype, device=data.device)
    return QuantizeStaticAffine.apply(data, scale, offset, tile_size, num_bits, quantized_dtype)


def dequantize_affine(
    data: torch.Tensor,
    scale: float | torch.Tensor,
    offset: float | torch.Tensor | None,
    tile_size: torch.Size | Literal["data_shape"],
    dtype: torch.dtype | None,
) -> torch.Tensor:
    if dtype is None:
        dtype = _float_dtype_of(data)
    scale = tensor_or_none(scale, dtype=dtype, device=data.device)
    offset = tensor_or_none(offset, dtype=dtype, device=data.device)
    return DequantizeAffine.apply(data, scale, offset, tile_size, dtype)


def quantize_dynamic_affine(
    data: torch.Tensor,
    tile_size: torch.Size | Literal["data_shape"],
    num_bits: int,
    symmetric: bool,
    allow_one_sided: bool,
    quantized_dtype: torch.dtype | None,
) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor | None]:
    return QuantizeDynamicAffine.apply(data, tile_size, num_bits, symmetric, allow_one_sided, quantized_dtype)


def _resolve(data: torch.Tensor, tile_size: Any) -> torch.Size:
    return data.shape if isinstance(tile_size, str) else tile_size


class QuantizeStaticAffine(torch.autograd.Function):
    @staticmethod
    def forward(ctx: Any, data, scale, offset, tile_size, num_bits, quantized_dtype):  # type: ignore[no-untyped-def]
        tile_size = _resolve(data, tile_size)
        ctx.save_for_backward(data, scale, offset)
        ctx.tile_size, ctx.num_bits = tile_size, num_bits
        return _OPS.quantize_by_tile(data, scale, list(tile_size), float(num_bits), quantized_dtype or data.dtype, offset)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx: Any, output_grad):  # type: ignore[no-untyped-def]
        data, scale, offset = ctx.saved_tensors
        dinput, dscale, doffset = _OPS.quantize_by_tile_backward(
            data, output_grad, scale, list(ctx.tile_size), float(ctx.num_bits), offset
        )
        return dinput, dscale, (doffset if offset is not None else None), None, None, None


class QuantizeDynamicAffine(torch.autograd.Function):
    @staticmethod
    def forward(ctx: Any, data, tile_size, num_bits, symmetric, allow_one_sided, quantized_dtype):  # type: ignore[no-untyped-def]
        return _OPS.quantize_dynamic_by_tile(
            data, list(_resolve(data, tile_size)), float(num_bits), symmetric, allow_one_sided, quantized_dtype or data.dtype
        )

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx: Any, output_grad, scale_grad, offset_grad):  # type: ignore[no-untyped-def]
        return output_grad, None, None, None, None, None


class DequantizeAffine(torch.autograd.Function):
    @staticmethod
    def forward(ctx: Any, data, scale, offset, tile_size, dtype):  # type: ignore[no-untyped-def]
        return _OPS.dequantize_by_tile(data, scale, list(_resolve(data, tile_size)), offset, dtype)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx: Any, output_grad):  # type: ignore[no-untyped-def]
        return output_grad, None, None, None, None
