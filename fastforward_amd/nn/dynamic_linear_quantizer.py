"""Per-call min/max affine quantizer (reference: src/fastforward/nn/dynamic_linear_quantizer.py).

Each call runs A3 (``quantize_dynamic_by_tile``): one fused device sequence of the min/max
reduction, the range -> parameter kernel and the quantize pass.
"""

from __future__ import annotations

from typing import Any

import torch

from fastforward_amd.nn.linear_quantizer import AbstractAffineQuantizer
from fastforward_amd.quantization import granularity as granularities
from fastforward_amd.quantization.affine import (
    AffineQuantizationFunction,
    DynamicAffineQuantParams,
    DynamicParamInferenceFn,
)
from fastforward_amd.quantization.function import QuantizationFunction


class DynamicLinearQuantizer(AbstractAffineQuantizer):
    def __init__(
        self,
        num_bits: int,
        *,
        granularity: granularities.Granularity | None = None,
        quantized_dtype: torch.dtype | None = None,
        parameter_inference_fn: DynamicParamInferenceFn | None = None,
        allow_one_sided: bool = True,
        symmetric: bool = False,
    ) -> None:
        super().__init__(num_bits=num_bits, granularity=granularity, quantized_dtype=quantized_dtype)
        self.parameter_inference_fn = parameter_inference_fn
        self.symmetric = symmetric
        self.allow_one_sided = allow_one_sided

    def quantization_parameters(self) -> DynamicAffineQuantParams:
        return DynamicAffineQuantParams(
            granularity=self.granularity,
            num_bits=self.num_bits,
            quantized_dtype=self.quantized_dtype,
            parameter_inference_fn=self.parameter_inference_fn,
            symmetric=self.symmetric,
            allow_one_sided=self.allow_one_sided,
        )

    @property
    def quantization_function(self) -> type[QuantizationFunction[Any]]:
        return AffineQuantizationFunction

    def reset_parameters(self) -> None:
        pass
