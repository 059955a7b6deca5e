"""Affine quantization function and its parameter records.

Reference: src/fastforward/quantization/affine/function.py — ``StaticAffineQuantParams`` (:31-40),
``DynamicAffineQuantParams`` (:49-59), ``AffineQuantizationFunction`` (:66-188): static / dynamic /
export routing on quantize, and dequantize refusing dynamic parameters (:182-183).
"""

from __future__ import annotations

import dataclasses

from typing import TYPE_CHECKING, Any, Callable, TypeAlias

import torch

from fastforward_amd import flags
from fastforward_amd.exceptions import QuantizationError
from fastforward_amd.quantization import granularity as granularities
from fastforward_amd.quantization.affine._memo import RECENT
from fastforward_amd.quantization.affine._autograd import (
    dequantize_affine,
    quantize_affine,
    quantize_dynamic_affine,
)
from fastforward_amd.quantization.function import (
    QuantizationContext,
    QuantizationFunction,
    QuantizationParameters,
    _fields_nocopy,
)

if TYPE_CHECKING:
    from fastforward_amd.quantized_tensor import QuantizedTensor


class ExportError(QuantizationError):
    """Export related error."""


@dataclasses.dataclass
class StaticAffineQuantParams(QuantizationParameters):
    """Parameters of static affine quantization."""

    scale: float | torch.Tensor
    offset: float | torch.Tensor | None
    num_bits: int
    granularity: granularities.Granularity
    quantized_dtype: torch.dtype | None = None
    dequantize_dtype: torch.dtype | None = None


DynamicParamInferenceFn: TypeAlias = Callable[["DynamicAffineQuantParams", torch.Tensor], tuple[torch.Tensor, "torch.Tensor | None"]]


@dataclasses.dataclass
class DynamicAffineQuantParams(QuantizationParameters):
    """Parameters of dynamic (per-call min/max) affine quantization."""

    num_bits: int
    granularity: granularities.Granularity
    symmetric: bool = False
    allow_one_sided: bool = True
    quantized_dtype: torch.dtype | None = None
    dequantize_dtype: torch.dtype | None = None
    parameter_inference_fn: DynamicParamInferenceFn | None = None


def _static_from_dynamic(params: DynamicAffineQuantParams, scale: torch.Tensor, offset: torch.Tensor | None, **changes: Any) -> StaticAffineQuantParams:
    names = {f.name for f in dataclasses.fields(StaticAffineQuantParams)}
    values = {k: v for k, v in _fields_nocopy(params).items() if k in names}
    values.update(scale=scale, offset=offset, **changes)
    return StaticAffineQuantParams(**values)


class AffineQuantizationFunction(QuantizationFunction[Any]):
    """Standard affine quantization; arithmetic runs in the HIP kernels via fastforward_amd.ops."""

    @classmethod
    def quantize(cls, data: torch.Tensor, params: Any) -> "QuantizedTensor":
        if flags.get_export_mode():
            return cls._export_quantize(data, params)  # type: ignore[return-value]
        if isinstance(params, StaticAffineQuantParams):
            return cls._static_quantize(data, params)
        if isinstance(params, DynamicAffineQuantParams):
            return cls._dynamic_quantize(data, params)
        raise TypeError(f"Unsupported type for argument 'params': '{type(params)}'")

    @classmethod
    def _export_quantize(cls, data: torch.Tensor, params: Any) -> torch.Tensor:
        """Quantize immediately followed by dequantize, returning a plain tensor (reference :93-122)."""
        if not isinstance(params, StaticAffineQuantParams):
            raise ExportError("Export supports only static affine quantization.")
        tile = params.granularity.tile_size(data.shape)
        store = params.quantized_dtype or data.dtype
        q = quantize_affine(data, params.scale, params.offset, tile, params.num_bits, store)
        return dequantize_affine(q, params.scale, params.offset, tile, store)

    @classmethod
    def _static_quantize(cls, data: torch.Tensor, params: StaticAffineQuantParams) -> "QuantizedTensor":
        from fastforward_amd.quantized_tensor import QuantizedTensor

        tile = params.granularity.tile_size(data.shape)
        container = params.quantized_dtype or data.dtype
        # the same activation quantized again by a quantizer with equal parameters (q / k / v, gate / up): the earlier codes
        raw = RECENT.lookup(data, params, tile, container)
        if raw is None:
            # ... or parameters an estimator has just rewritten, which only the device can compare with an earlier sibling's
            # (inside ``sibling_quantizers(undecided=True)``): A1 that runs unless they are the same, the result marked as such
            earlier = RECENT.earlier_for(data, params, tile, container)
            if earlier is not None:
                from fastforward_amd import ops

                raw = ops.quantize_by_tile_unless_same(data, params.scale, params.offset, params.num_bits, earlier[1], earlier[2])
                if raw is not None:
                    stamped = params.with_changes(dequantize_dtype=params.dequantize_dtype or data.dtype)
                    quantized = QuantizedTensor(raw, QuantizationContext(cls, stamped))
                    RECENT.mark_undecided(quantized, earlier, params.scale, params.offset)
                    return quantized
        if raw is None:
            raw = quantize_affine(data, params.scale, params.offset, tile, params.num_bits, container)
            RECENT.remember(data, params, tile, container, raw)
        # the dequantize dtype is stamped at quantize time                      (reference :137)
        stamped = params.with_changes(dequantize_dtype=params.dequantize_dtype or data.dtype)
        return QuantizedTensor(raw, QuantizationContext(cls, stamped))

    @classmethod
    def _dynamic_quantize(cls, data: torch.Tensor, params: DynamicAffineQuantParams) -> "QuantizedTensor":
        from fastforward_amd.quantized_tensor import QuantizedTensor

        deq = params.dequantize_dtype or data.dtype
        if params.parameter_inference_fn is not None:
            scale, offset = params.parameter_inference_fn(params, data)
            return cls._static_quantize(data, _static_from_dynamic(params, scale, offset, dequantize_dtype=deq))
        tile = params.granularity.tile_size(data.shape)
        tile = data.shape if isinstance(tile, str) else tile
        raw, scale, offset = quantize_dynamic_affine(
            data, tile, params.num_bits, params.symmetric, params.allow_one_sided, params.quantized_dtype or data.dtype
        )
        static = _static_from_dynamic(params, scale, offset, dequantize_dtype=deq)
        return QuantizedTensor(raw, QuantizationContext(AffineQuantizationFunction, static))

    @classmethod
    def dequantize(cls, data: torch.Tensor, params: Any) -> torch.Tensor:
        if isinstance(params, DynamicAffineQuantParams):
            raise TypeError("Cannot dequantize a QuantizedTensor with dynamic parameters.")
        tile = params.granularity.tile_size(data.shape)
        return dequantize_affine(data, params.scale, params.offset, tile, params.dequantize_dtype)
