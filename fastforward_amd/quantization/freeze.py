"""``freeze_parameters`` — write every parameter's quantize -> dequantize value back into the parameter during ONE forward
pass and retire its quantizer (reference: src/fastforward/quantization/freeze.py:13-125).

Where :func:`fuse_qdq_weights` finds weight quantizers by convention, this follows the data: for the duration of the
context every quantizer of the given modules carries an override that runs the quantizer's own forward (all other active
overrides and hooks included), dequantizes the result (A1 then A2, both on the device), and — if the tensor that came in is
an ``nn.Parameter`` — copies the value into it in place. The quantizer is then replaced by a ``QuantizerStub`` with the same
metadata (unless ``remove_quantizers=False``). A quantizer that hands its input back unchanged (e.g. under
``ff.disable_quantization``) is left alone, and so is its parameter. This is SURVEY 8(f) row 1: after freezing, a forward
never re-quantizes the 6.98 G weight elements.

Unlike the reference (whose list of handles is never filled, :112-125) the overrides are removed again when the context
exits, as its docstring promises.
"""

from __future__ import annotations

import contextlib

from typing import Any, Callable, Iterator, Sequence

import torch

import fastforward_amd as ff

from fastforward_amd.nn.quantized_module import named_quantizers
from fastforward_amd.nn.quantizer import Quantizer, QuantizerStub


class _FreezeOnCall:
    """Quantizer override: freeze what flows through `owner`'s quantizer."""

    def __init__(self, owner: torch.nn.Module, retire: bool) -> None:
        self.owner, self.retire = owner, retire

    def __call__(self, quantizer: Quantizer, forward: Callable[..., Any], args: tuple[Any, ...], kwargs: dict[str, Any]) -> torch.Tensor:
        data = args[0] if args else kwargs["data"]
        produced = forward(data)
        value = produced.dequantize() if isinstance(produced, ff.QuantizedTensor) else produced
        if value is data:  # nothing was quantized (disabled quantizer): neither the parameter nor the quantizer changes
            return data
        if isinstance(data, torch.nn.Parameter):
            with torch.no_grad():
                data.copy_(value)
        if self.retire:
            for name, candidate in named_quantizers(self.owner, recurse=False, skip_stubs=False):
                if candidate is quantizer:
                    setattr(self.owner, name, QuantizerStub(_metadata=quantizer.quant_metadata))
                    break
        return data  # as the reference (:60): the caller sees the tensor it passed in (a frozen parameter now holds `value`)


@contextlib.contextmanager
def freeze_parameters(modules: torch.nn.Module | Sequence[torch.nn.Module], remove_quantizers: bool = True) -> Iterator[None]:
    """Within the context, ONE forward pass freezes every ``nn.Parameter`` that reaches a quantizer of `modules` to its
    quantized value (same dtype, grid-snapped contents) and, by default, swaps those quantizers for stubs. Transformed
    parameters (no longer ``nn.Parameter`` objects) are not written back. Strict quantization is off inside the context:
    the overrides return plain tensors (reference :74-125)."""
    roots = [modules] if isinstance(modules, torch.nn.Module) else list(modules)
    handles = []
    for root in roots:
        for owner in root.modules():
            for _, quantizer in named_quantizers(owner, recurse=False):
                handles.append(quantizer.register_override(_FreezeOnCall(owner, retire=remove_quantizers)))
    try:
        with ff.strict_quantization(False):
            yield
    finally:
        for handle in handles:
            handle.remove()
