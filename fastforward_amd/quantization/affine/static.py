"""Convenience entry points for static affine quantization (reference: affine/static.py:19-212)."""

from __future__ import annotations

from typing import TYPE_CHECKING

import torch

from fastforward_amd.quantization import granularity as granularities
from fastforward_amd.quantization.affine.function import AffineQuantizationFunction, StaticAffineQuantParams
from fastforward_amd.quantization.function import QuantizationContext

if TYPE_CHECKING:
    from fastforward_amd.quantized_tensor import QuantizedTensor

_Param = torch.Tensor | float


def quantization_context(
    scale: _Param,
    offset: _Param | None,
    granularity: granularities.Granularity | None = None,
    num_bits: int = 8,
    output_dtype: torch.dtype | None = None,
    dequantize_dtype: torch.dtype | None = None,
) -> QuantizationContext[StaticAffineQuantParams]:
    params = StaticAffineQuantParams(
        scale=scale,
        offset=offset,
        num_bits=num_bits,
        granularity=granularity or granularities.PerTensor(),
        quantized_dtype=output_dtype,
        dequantize_dtype=dequantize_dtype,
    )
    return QuantizationContext(AffineQuantizationFunction, params)


def _quantize(input: torch.Tensor, scale: _Param, offset: _Param | None, granularity: granularities.Granularity, num_bits: int, output_dtype: torch.dtype | None) -> "QuantizedTensor":
    params = StaticAffineQuantParams(
        scale=scale, offset=offset, num_bits=num_bits, granularity=granularity, quantized_dtype=output_dtype
    )
    return AffineQuantizationFunction.quantize(input, params)


def quantize_by_tile(input: torch.Tensor, scale: _Param, offset: _Param | None, tile_size: torch.Size, num_bits: int = 8, output_dtype: torch.dtype | None = None) -> "QuantizedTensor":
    return _quantize(input, scale, offset, granularities.PerTile(tile_shape=tile_size), num_bits, output_dtype)


def quantize_per_tensor(input: torch.Tensor, scale: _Param, offset: _Param | None = None, num_bits: int = 8, output_dtype: torch.dtype | None = None) -> "QuantizedTensor":
    return _quantize(input, scale, offset, granularities.PerTensor(), num_bits, output_dtype)


def quantize_per_channel(input: torch.Tensor, scale: _Param, offset: _Param | None = None, axis: int | tuple[int, ...] = -1, num_bits: int = 8, output_dtype: torch.dtype | None = None) -> "QuantizedTensor":
    return _quantize(input, scale, offset, granularities.PerChannel(axis), num_bits, output_dtype)


def quantize_per_block(input: torch.Tensor, scale: torch.Tensor, offset: torch.Tensor, channel_axis: int, block_axis: int, block_size: int, num_bits: int = 8, output_dtype: torch.dtype | None = None) -> "QuantizedTensor":
    tile = list(input.shape)
    tile[channel_axis] = 1
    tile[block_axis] = block_size
    return quantize_by_tile(input, scale, offset, torch.Size(tile), num_bits, output_dtype)


def quantize_per_granularity(input: torch.Tensor, scale: _Param, offset: _Param | None, granularity: granularities.Granularity, num_bits: int = 8, output_dtype: torch.dtype | None = None) -> "QuantizedTensor":
    if isinstance(granularity, granularities.PerTensor):
        return quantize_per_tensor(input, scale, offset, num_bits, output_dtype)
    if isinstance(granularity, granularities.PerChannel):
        return quantize_per_channel(input, scale, offset, granularity.channel_dims, num_bits, output_dtype)
    tile = granularity.tile_size(input.shape)
    assert not isinstance(tile, str)
    return quantize_by_tile(input, scale, offset, tile, num_bits, output_dtype)
