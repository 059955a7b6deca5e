"""Quantization function / parameters / context containers.

Restates the subset of src/fastforward/quantization/function.py the hot path needs:
``QuantizationParameters`` (:22-45), the ``QuantizationFunction`` ABC (:52-76) and
``QuantizationContext`` (:78-166): a frozen pair (function class, parameters) that a
``QuantizedTensor`` carries so it can be dequantized, moved, cloned or re-attached to new raw data.
"""

from __future__ import annotations

import abc
import dataclasses
import functools

from typing import TYPE_CHECKING, Any, Callable, Generic, TypeVar

import torch

from typing_extensions import Self

from fastforward_amd import flags
from fastforward_amd.common import maybe_tensor_apply

if TYPE_CHECKING:
    from fastforward_amd.quantized_tensor import QuantizedTensor


def _fields_nocopy(obj: Any) -> dict[str, Any]:
    """Field dict of a dataclass WITHOUT deep-copying values (tensors must stay shared)."""
    return {f.name: getattr(obj, f.name) for f in dataclasses.fields(obj)}


@dataclasses.dataclass
class QuantizationParameters:
    """Base class of per-function parameter records."""

    def with_changes(self, **changes: Any) -> Self:
        return dataclasses.replace(self, **changes)

    def _apply(self, fn: Callable[[Any], Any]) -> Self:
        return type(self)(**{k: fn(v) for k, v in _fields_nocopy(self).items()})

    def __format__(self, format_spec: str, /) -> str:
        return repr(self)


QuantParams = TypeVar("QuantParams", bound=QuantizationParameters)


class QuantizationFunction(Generic[QuantParams], abc.ABC):
    """``quantize(data, params) -> QuantizedTensor`` / ``dequantize(raw, params) -> Tensor``."""

    @classmethod
    @abc.abstractmethod
    def quantize(cls, data: torch.Tensor, params: QuantParams) -> "QuantizedTensor": ...

    @classmethod
    @abc.abstractmethod
    def dequantize(cls, data: torch.Tensor, params: QuantParams) -> torch.Tensor: ...


@dataclasses.dataclass(frozen=True)
class QuantizationContext(Generic[QuantParams]):
    """Everything needed to (de)quantize: the function class and its parameters."""

    quantization_fn: type[QuantizationFunction[QuantParams]]
    quantization_params: QuantParams

    def with_changes(self, quantization_fn: type[QuantizationFunction[QuantParams]] | None = None, **changes: Any) -> Self:
        replaced: dict[str, Any] = {"quantization_params": self.quantization_params.with_changes(**changes)}
        if quantization_fn is not None:
            replaced["quantization_fn"] = quantization_fn
        return dataclasses.replace(self, **replaced)

    def _apply(self, fn: Callable[[Any], Any]) -> Self:
        return dataclasses.replace(self, quantization_params=self.quantization_params._apply(fn))

    def clone_parameters(self) -> Self:
        return self._apply(functools.partial(maybe_tensor_apply, fn=torch.clone))

    def detach_parameters(self) -> Self:
        return self._apply(functools.partial(maybe_tensor_apply, fn=torch.detach))

    def contiguous_parameters(self) -> Self:
        """Same object when every tensor parameter already is contiguous (reference :136-147)."""
        candidate = self._apply(functools.partial(maybe_tensor_apply, fn=torch.Tensor.contiguous))
        before = _fields_nocopy(self.quantization_params)
        after = _fields_nocopy(candidate.quantization_params)
        return candidate if any(before[k] is not after[k] for k in before) else self

    def to(self, device: torch.device | str) -> Self:
        return self._apply(functools.partial(maybe_tensor_apply, fn=lambda t: t.to(device=device)))

    def attach(self, data: torch.Tensor) -> "QuantizedTensor":
        """Wrap `data` (raw codes) as a QuantizedTensor carrying this context (reference :158-166)."""
        from fastforward_amd.quantized_tensor import QuantizedTensor

        if flags.get_export_mode():
            return self.quantization_fn.dequantize(data, self.quantization_params)  # type: ignore[return-value]
        return QuantizedTensor(data, self)
