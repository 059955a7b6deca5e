"""``disable_quantization`` / ``enable_quantization`` (reference: src/fastforward/overrides.py:23-59)."""

from __future__ import annotations

import contextlib

from typing import Any, Callable, Generator, Iterable

import torch

from fastforward_amd import flags
from fastforward_amd import forward_override as override
from fastforward_amd.nn.quantized_module import named_quantizers
from fastforward_amd.nn.quantizer import Quantizer


class DisableQuantizationOverride:
    """Quantizer override returning its input untouched while quantization is disabled (reference :62-153)."""

    def __init__(self) -> None:
        self._quantization_enabled = False
        self._handles: list[override.OverrideHandle] = []

    @property
    def quantization_enabled(self) -> bool:
        return self._quantization_enabled

    @contextlib.contextmanager
    def _restore(self, previous: bool) -> Generator[None, None, None]:
        try:
            yield
        finally:
            self._quantization_enabled = previous

    def enable_quantization(self, enabled: bool = True) -> contextlib.AbstractContextManager[None]:
        previous, self._quantization_enabled = self._quantization_enabled, enabled
        return self._restore(previous)

    def disable_quantization(self) -> contextlib.AbstractContextManager[None]:
        return self.enable_quantization(False)

    def __call__(self, _context: Any, callback: Callable[..., torch.Tensor], args: tuple[Any, ...], kwargs: dict[str, Any]) -> torch.Tensor:
        if self._quantization_enabled:
            return callback(*args, **kwargs)
        return (lambda data, *a, **k: data)(*args, **kwargs)

    def __repr__(self) -> str:
        return f"{type(self).__name__}(quantization_enabled={self._quantization_enabled})"

    def attach_to(self, quantizers: Quantizer | Iterable[Quantizer]) -> None:
        if isinstance(quantizers, Quantizer):
            self._handles.append(quantizers.register_override(self))
            return
        for quantizer in quantizers:
            self.attach_to(quantizer)

    def detach(self) -> None:
        for handle in self._handles:
            handle.remove()
        self._handles = []


@contextlib.contextmanager
def disable_quantization(model: torch.nn.Module) -> Generator[None, None, None]:
    """All quantizers of `model` become identities inside the context; strict mode is switched off."""
    handles = [q.register_override(DisableQuantizationOverride()) for _, q in named_quantizers(model)]
    try:
        with flags.strict_quantization(False):
            yield
    finally:
        for handle in handles:
            handle.remove()


@contextlib.contextmanager
def enable_quantization(model: torch.nn.Module) -> Generator[None, None, None]:
    """Re-enable quantizers that were disabled by a ``DisableQuantizationOverride`` inside the context."""
    with contextlib.ExitStack() as stack:
        for _, quantizer in named_quantizers(model):
            for fn in quantizer.overrides:
                if isinstance(fn, DisableQuantizationOverride):
                    stack.enter_context(fn.enable_quantization())
        yield
