"""Range-estimation scaffolding (reference: src/fastforward/range_setting/common.py).

``estimate_ranges(model, estimator)`` installs an estimator override on every non-stub quantizer
(:241-289); inside the context each forward pass runs ``estimate_step`` before (or, with
``disable_quantization=True``, instead of) quantizing (:218-238).
"""

from __future__ import annotations

import abc
import contextlib

from typing import Any, Callable, Generator, Generic, Iterator, Protocol, Sequence, TypeVar, runtime_checkable

import torch

from fastforward_amd.quantization import granularity
from fastforward_amd.quantized_tensor import QuantizedTensor


@runtime_checkable
class RangeSettable(Protocol):
    """Quantizers whose parameters can be given as a (min, max) range (reference :29-65)."""

    @property
    def granularity(self) -> granularity.Granularity: ...

    @property
    def quantization_range(self) -> tuple[torch.Tensor | None, torch.Tensor | None]: ...

    @quantization_range.setter
    def quantization_range(self, __range: tuple[torch.Tensor, torch.Tensor]) -> None: ...


@runtime_checkable
class SupportsRangeBasedOperator(RangeSettable, Protocol):
    """Quantizers that can build an operator for an arbitrary range (reference :68-110)."""

    @property
    def symmetric(self) -> bool: ...

    def operator_for_range(self, __min: torch.Tensor, __max: torch.Tensor, __data_shape: torch.Size) -> Callable[[torch.Tensor], QuantizedTensor]: ...


_T = TypeVar("_T")
_Module = TypeVar("_Module", bound=torch.nn.Module)


class RangeEstimator(abc.ABC, Generic[_T, _Module]):
    """prepare / cleanup / split_module hooks used by :func:`estimate_ranges` (reference :117-172)."""

    @abc.abstractmethod
    def prepare(self, module: _Module) -> _T: ...

    @abc.abstractmethod
    def cleanup(self, module: _Module, metadata: _T) -> None: ...

    @abc.abstractmethod
    def split_module(self, module: torch.nn.Module) -> Iterator[_Module]: ...


class SimpleEstimatorStep(abc.ABC):
    """Override body: ``estimate_step`` then quantize (or pass data through) — reference :178-238."""

    def __init__(self, *args: Any, disable_quantization: bool = False, **kwargs: Any) -> None:
        self._initialized = False
        self._disable_quantization = disable_quantization
        super().__init__(*args, **kwargs)

    def setup_estimator(self, data: torch.Tensor) -> None:
        """Called once with the first batch."""

    @abc.abstractmethod
    def estimate_step(self, quantizer: Any, data: torch.Tensor) -> None: ...

    def forward(self, quantizer: Any, callback: Callable[[torch.Tensor], torch.Tensor], args: tuple[Any, ...], kwargs: dict[str, Any]) -> torch.Tensor:
        data = (lambda d, *a, **k: d)(*args, **kwargs)
        if not self._initialized:
            self.setup_estimator(data)
            self._initialized = True
        self.estimate_step(quantizer, data)
        return data if self._disable_quantization else callback(data)


@contextlib.contextmanager
def estimate_ranges(
    model_or_layers: torch.nn.Module | Sequence[torch.nn.Module],
    estimator: RangeEstimator[Any, Any] | type[RangeEstimator[Any, Any]],
    *args: Any,
    **kwargs: Any,
) -> Generator[None, None, None]:
    """Within the context every forward through `model_or_layers` is a range-estimation step."""
    layers = [model_or_layers] if isinstance(model_or_layers, torch.nn.Module) else list(model_or_layers)
    if isinstance(estimator, type):
        estimator = estimator(*args, **kwargs)
    elif args or kwargs:
        raise ValueError("`estimator` is already initialized so no `args` or `kwargs` can be given.")
    prepared = []
    for layer in layers:
        for part in estimator.split_module(layer):
            prepared.append((part, estimator.prepare(part)))
    try:
        yield
    finally:
        for part, metadata in prepared:
            estimator.cleanup(part, metadata)
