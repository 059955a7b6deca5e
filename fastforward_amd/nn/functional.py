"""Quantized functional operators on the linear path: ``linear``, ``matmul``, ``mm``, ``bmm``.

Reference: the generated ``ff.nn.functional.*`` (src/fastforward/_gen/operators.py:79-106 for
``linear``; matmul/mm/bmm follow the same template) and their fallbacks
(src/fastforward/_gen/fallback.py:77-112, 699-798). Each operator is
``dispatch(name, **kwargs) or fallback`` — the dispatcher lookup is plug-in seam #2, where
``fastforward_amd.fused_linear`` registers the int8-MFMA kernel. The other 49 generated operators of
the reference are pure float fallbacks and are out of scope.
"""

from __future__ import annotations

from typing import TYPE_CHECKING, Any, Callable, Optional

import torch

from fastforward_amd import flags
from fastforward_amd.dispatcher import dispatch
from fastforward_amd.exceptions import QuantizationError
from fastforward_amd.quantized_tensor import QuantizedTensor

if TYPE_CHECKING:
    from fastforward_amd.nn.quantizer import Quantizer

__all__ = ["linear", "matmul", "mm", "bmm"]


def _dequantized(name: str, value: Any, strict: bool, required: bool = True) -> Any:
    if strict and required and not isinstance(value, QuantizedTensor):
        raise QuantizationError(
            f"Expected '{name}' to be an instance of 'QuantizedTensor' because strict_quantization=True."
        )
    return value.dequantize() if isinstance(value, QuantizedTensor) else value


def _check_output_quantizer(output_quantizer: Any, strict: bool) -> None:
    if strict and output_quantizer is None:
        raise QuantizationError("'output_quantizer' must be provided if strict_quantization=True")


def _fallback_linear(input: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor | None = None, *, output_quantizer: Optional["Quantizer"] = None, strict_quantization: bool = True) -> torch.Tensor:
    """Dequantize operands, float linear, optional output quantizer (reference fallback.py:77-112)."""
    _check_output_quantizer(output_quantizer, strict_quantization)
    input = _dequantized("input", input, strict_quantization)
    weight = _dequantized("weight", weight, strict_quantization)
    bias = _dequantized("bias", bias, strict_quantization, required=False)
    output = torch.nn.functional.linear(input=input, weight=weight, bias=bias)
    return output_quantizer(output) if output_quantizer is not None else output


def _binary_fallback(torch_op: Callable[..., torch.Tensor], second: str) -> Callable[..., torch.Tensor]:
    def fallback(input: torch.Tensor, other: torch.Tensor, *, output_quantizer: Optional["Quantizer"] = None, strict_quantization: bool = True) -> torch.Tensor:
        _check_output_quantizer(output_quantizer, strict_quantization)
        input = _dequantized("input", input, strict_quantization)
        other = _dequantized(second, other, strict_quantization)
        output = torch_op(input, other)
        return output_quantizer(output) if output_quantizer is not None else output

    return fallback


_fallback_matmul = _binary_fallback(torch.matmul, "other")
_fallback_mm = _binary_fallback(torch.mm, "mat2")
_fallback_bmm = _binary_fallback(torch.bmm, "mat2")


def _strict(value: bool | None) -> bool:
    return flags.get_strict_quantization() if value is None else value


def linear(input: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor | None = None, *, output_quantizer: Optional["Quantizer"] = None, strict_quantization: bool | None = None) -> torch.Tensor:
    kwargs = dict(input=input, weight=weight, bias=bias, output_quantizer=output_quantizer, strict_quantization=_strict(strict_quantization))
    kernel = dispatch("linear", **kwargs) or _fallback_linear
    # codes of a sibling quantizer whose A1 launch the device may have skipped (quantization/affine/_memo.py): a kernel that has
    # not declared that it reads the codes in force gets them written first
    if getattr(input, "_ffq_earlier", None) is not None and not getattr(kernel, "reads_undecided_codes", False):
        from fastforward_amd.quantization.affine._memo import RECENT

        RECENT.settle(input)
    return kernel(**kwargs)


def matmul(input: torch.Tensor, other: torch.Tensor, *, output_quantizer: Optional["Quantizer"] = None, strict_quantization: bool | None = None) -> torch.Tensor:
    kwargs = dict(input=input, other=other, output_quantizer=output_quantizer, strict_quantization=_strict(strict_quantization))
    kernel = dispatch("matmul", **kwargs)
    if kernel:
        return kernel(**kwargs)
    return _fallback_matmul(input, other, output_quantizer=output_quantizer, strict_quantization=kwargs["strict_quantization"])


def mm(input: torch.Tensor, mat2: torch.Tensor, *, output_quantizer: Optional["Quantizer"] = None, strict_quantization: bool | None = None) -> torch.Tensor:
    kwargs = dict(input=input, mat2=mat2, output_quantizer=output_quantizer, strict_quantization=_strict(strict_quantization))
    kernel = dispatch("mm", **kwargs)
    if kernel:
        return kernel(**kwargs)
    return _fallback_mm(input, mat2, output_quantizer=output_quantizer, strict_quantization=kwargs["strict_quantization"])


def bmm(input: torch.Tensor, mat2: torch.Tensor, *, output_quantizer: Optional["Quantizer"] = None, strict_quantization: bool | None = None) -> torch.Tensor:
    kwargs = dict(input=input, mat2=mat2, output_quantizer=output_quantizer, strict_quantization=_strict(strict_quantization))
    kernel = dispatch("bmm", **kwargs)
    if kernel:
        return kernel(**kwargs)
    return _fallback_bmm(input, mat2, output_quantizer=output_quantizer, strict_quantization=kwargs["strict_quantization"])
