"""View-like operators on affine-quantized tensors that act on the raw codes.

Subset of src/fastforward/quantization/_linear_quantized_ops.py needed on the Llama linear path
(SURVEY §2): ``contiguous`` (:94-96) for any quantized tensor and ``view`` / ``view_as`` /
``reshape`` / ``transpose`` for per-tensor affine tensors (:99-123). They only move metadata.
"""

from __future__ import annotations

from typing import Any

import torch

from fastforward_amd.dispatcher import Predicate, register
from fastforward_amd.quantization import granularity as granularities
from fastforward_amd.quantized_tensor import QuantizedTensor, apply_and_reattach


def _static_affine(tensor: Any) -> bool:
    from fastforward_amd.quantization.affine import AffineQuantizationFunction, StaticAffineQuantParams

    if not isinstance(tensor, QuantizedTensor):
        return False
    context = tensor.quantization_context
    return issubclass(context.quantization_fn, AffineQuantizationFunction) and isinstance(
        context.quantization_params, StaticAffineQuantParams
    )


def _granularity_of(tensor: QuantizedTensor) -> Any:
    return getattr(tensor.quantization_context.quantization_params, "granularity", None)


affine_predicate = Predicate(lambda input, *a, **k: _static_affine(input))
affine_per_tensor_predicate = Predicate(
    lambda input, *a, **k: _static_affine(input) and isinstance(_granularity_of(input), granularities.PerTensor)
)
affine_per_channel_predicate = Predicate(
    lambda input, *a, **k: _static_affine(input) and isinstance(_granularity_of(input), granularities.PerChannel)
)


@register("contiguous")
def contiguous(input: QuantizedTensor) -> QuantizedTensor:
    return apply_and_reattach(lambda x: x.contiguous(), input)


def _no_dtype_view(name: str, args: tuple[Any, ...]) -> None:
    if args and isinstance(args[0], torch.dtype):
        raise TypeError(f"QuantizedTensor.{name}(dtype) is not supported")


@register("view", affine_per_tensor_predicate)
def view(input: QuantizedTensor, *args: Any) -> QuantizedTensor:
    _no_dtype_view("view", args)
    return apply_and_reattach(lambda x: x.view(*args), input)


@register("view_as", affine_per_tensor_predicate)
def view_as(input: QuantizedTensor, *args: Any) -> QuantizedTensor:
    _no_dtype_view("view_as", args)
    return apply_and_reattach(lambda x: x.view_as(*args), input)


@register("reshape", affine_per_tensor_predicate)
def reshape(input: QuantizedTensor, *args: Any) -> QuantizedTensor:
    return apply_and_reattach(lambda x: x.reshape(*args), input)


@register("transpose", affine_per_tensor_predicate)
def transpose(input: QuantizedTensor, *args: Any) -> QuantizedTensor:
    return apply_and_reattach(lambda x: x.transpose(*args), input)
