"""Convenience entry points for dynamic affine quantization (reference: affine/dynamic.py)."""

from __future__ import annotations

from typing import TYPE_CHECKING

import torch

from fastforward_amd.quantization import granularity as granularities
from fastforward_amd.quantization.affine.function import AffineQuantizationFunction, DynamicAffineQuantParams
from fastforward_amd.quantization.function import QuantizationContext

if TYPE_CHECKING:
    from fastforward_amd.quantized_tensor import QuantizedTensor


def quantization_context(
    granularity: granularities.Granularity,
    num_bits: int,
    symmetric: bool = False,
    allow_one_sided: bool = True,
    quantized_dtype: torch.dtype | None = None,
    dequantize_dtype: torch.dtype | None = None,
) -> QuantizationContext[DynamicAffineQuantParams]:
    params = DynamicAffineQuantParams(
        num_bits=num_bits,
        granularity=granularity,
        symmetric=symmetric,
        allow_one_sided=allow_one_sided,
        quantized_dtype=quantized_dtype,
        dequantize_dtype=dequantize_dtype,
    )
    return QuantizationContext(AffineQuantizationFunction, params)


def quantize_per_granularity(input: torch.Tensor, granularity: granularities.Granularity, num_bits: int = 8, symmetric: bool = False, allow_one_sided: bool = True, output_dtype: torch.dtype | None = None) -> "QuantizedTensor":
    params = DynamicAffineQuantParams(
        num_bits=num_bits,
        granularity=granularity,
        symmetric=symmetric,
        allow_one_sided=allow_one_sided,
        quantized_dtype=output_dtype,
    )
    return AffineQuantizationFunction.quantize(input, params)


def quantize_by_tile(input: torch.Tensor, tile_size: torch.Size, num_bits: int = 8, symmetric: bool = False, allow_one_sided: bool = True, output_dtype: torch.dtype | None = None) -> "QuantizedTensor":
    return quantize_per_granularity(input, granularities.PerTile(tile_size), num_bits, symmetric, allow_one_sided, output_dtype)


def quantize_per_tensor(input: torch.Tensor, num_bits: int = 8, symmetric: bool = False, allow_one_sided: bool = True, output_dtype: torch.dtype | None = None) -> "QuantizedTensor":
    return quantize_per_granularity(input, granularities.PerTensor(), num_bits, symmetric, allow_one_sided, output_dtype)


def quantize_per_channel(input: torch.Tensor, axis: int | tuple[int, ...] = -1, num_bits: int = 8, symmetric: bool = False, allow_one_sided: bool = True, output_dtype: torch.dtype | None = None) -> "QuantizedTensor":
    return quantize_per_granularity(input, granularities.PerChannel(axis), num_bits, symmetric, allow_one_sided, output_dtype)


def quantize_per_block(input: torch.Tensor, channel_axis: int, block_axis: int, block_size: int, num_bits: int = 8, symmetric: bool = False, allow_one_sided: bool = True, output_dtype: torch.dtype | None = None) -> "QuantizedTensor":
    tile = list(input.shape)
    tile[channel_axis] = 1
    tile[block_axis] = block_size
    return quantize_by_tile(input, torch.Size(tile), num_bits, symmetric, allow_one_sided, output_dtype)
