"""Per-quantizer override stack (reference: src/fastforward/forward_override.py).

An override is a callable ``(quantizer, next_fn, args, kwargs) -> Tensor`` registered on a
quantizer; calling the quantizer runs the overrides newest-first, each deciding whether to call
``next_fn`` (the next override, finally ``quantizer.quantize``). Range estimators and
``disable_quantization`` hook into the hot path through this mechanism (reference :76-126).
"""

from __future__ import annotations

import weakref

from typing import TYPE_CHECKING, Any, Callable, Mapping, Protocol, TypeVar

if TYPE_CHECKING:
    from fastforward_amd.nn.quantizer import Quantizer

T = TypeVar("T")


class OverrideFn(Protocol[T]):
    def __call__(self, __context: Any, __overridden_fn: Callable[..., T], __args: tuple[Any, ...], __kwargs: dict[str, Any]) -> T: ...


class OverrideHandle:
    """Returned by ``Quantizer.register_override``; ``remove()`` (or leaving a ``with``) unregisters."""

    global_handles: int = 0

    def __init__(self, quantizer: "Quantizer") -> None:
        self._quantizer = weakref.ref(quantizer)
        self.handle_id = OverrideHandle.global_handles
        OverrideHandle.global_handles += 1

    def remove(self) -> OverrideFn[Any] | None:
        quantizer = self._quantizer()
        return None if quantizer is None else quantizer.remove_override(self.handle_id)

    def __enter__(self) -> "OverrideHandle":
        return self

    def __exit__(self, *exc: object) -> None:
        self.remove()


class _Chain:
    """Callable that peels one override per call, innermost being the overridden function."""

    def __init__(self, context: Any, innermost: Callable[..., Any], overrides: Mapping[int, OverrideFn[Any]]) -> None:
        self.context = context
        self.innermost = innermost
        self.pending = [fn for _, fn in sorted(overrides.items())]  # oldest ... newest

    def __call__(self, *args: Any, **kwargs: Any) -> Any:
        if not self.pending:
            return self.innermost(*args, **kwargs)
        newest = self.pending.pop()
        return newest(self.context, self, args, kwargs)


def apply_overrides(context: Any, overridden_fn: Callable[..., T], override_map: Mapping[int, OverrideFn[T]]) -> Callable[..., T]:
    """`overridden_fn` wrapped by the overrides in `override_map` (highest id outermost)."""
    if not override_map:
        return overridden_fn
    return _Chain(context, overridden_fn, override_map)
