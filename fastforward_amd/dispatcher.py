"""Predicate-based operator dispatcher — plug-in seam #2 of the reference.

Behavioural contract restated from src/fastforward/dispatcher.py:
  * ``register(op, predicate, kernel, priority)`` works as a plain call, as a decorator
    (``kernel=None``) and as a ``with`` block that unregisters on exit (:233-265, :120-139).
  * entries are ordered by priority; inside one priority the newest registration is tried first
    (``bisect_left`` insert, :200-202; order test tests/test_dispatcher.py:90-114).
  * ``dispatch(op, *args, **kwargs)`` returns the first kernel whose predicate accepts the
    arguments, or None so that the caller falls back (:268-283).
  * predicates compose with ``~``, ``&`` and ``|`` (:19-87).
"""

from __future__ import annotations

import bisect
import dataclasses
import enum

from collections import defaultdict
from typing import Any, Callable


class _Composable:
    def __call__(self, *args: Any, **kwargs: Any) -> bool:
        raise NotImplementedError

    def __invert__(self) -> "_Composable":
        return _Combined(lambda results: not next(results), (self,))

    def __and__(self, other: "_Composable") -> "_Composable":
        return _Combined(all, (self, other))

    def __or__(self, other: "_Composable") -> "_Composable":
        return _Combined(any, (self, other))


class _Combined(_Composable):
    def __init__(self, reducer: Callable[[Any], bool], parts: tuple[_Composable, ...]) -> None:
        self._reducer = reducer
        self._parts = parts

    def __call__(self, *args: Any, **kwargs: Any) -> bool:
        # generator => `all` / `any` short-circuit like the reference's implementation
        return bool(self._reducer(part(*args, **kwargs) for part in self._parts))


class Predicate(_Composable):
    """Wraps a function ``(*args, **kwargs) -> bool`` so it can be combined with ~, & and |."""

    def __init__(self, fn: Callable[..., bool]) -> None:
        self._fn = fn

    def __call__(self, *args: Any, **kwargs: Any) -> bool:
        return self._fn(*args, **kwargs)

    def __repr__(self) -> str:
        return f"Predicate({getattr(self._fn, '__name__', self._fn)!r})"


class DispatcherPriority(enum.IntEnum):
    """Evaluation order of registered kernels: DEFAULT first, then the fallbacks."""

    DEFAULT = 0
    FALLBACK = 1
    NOT_IMPLEMENTED_FALLBACK = 2


@dataclasses.dataclass
class DispatcherItem:
    predicate: _Composable
    fn: Callable[..., Any]
    priority: DispatcherPriority = DispatcherPriority.DEFAULT


_DISPATCHER: dict[str, list[DispatcherItem]] = defaultdict(list)

_ALWAYS = Predicate(lambda *args, **kwargs: True)


class DispatcherRegistrationHook:
    """Returned by the functional form of :func:`register`; leaving its ``with`` block removes the kernel."""

    def __init__(self, op_name: str, item: DispatcherItem) -> None:
        self._op_name = op_name
        self._item = item

    def __enter__(self) -> None:
        return None

    def __exit__(self, *exc: object) -> None:
        self.remove()

    def remove(self) -> None:
        items = _DISPATCHER[self._op_name]
        if self._item in items:
            items.remove(self._item)


def _insert(op_name: str, predicate: _Composable | None, kernel: Callable[..., Any], priority: DispatcherPriority) -> DispatcherRegistrationHook:
    item = DispatcherItem(predicate or _ALWAYS, kernel, priority)
    items = _DISPATCHER[op_name]
    items.insert(bisect.bisect_left(items, priority, key=lambda it: it.priority), item)
    return DispatcherRegistrationHook(op_name, item)


def register(
    op_name: str,
    predicate: _Composable | None = None,
    kernel: Callable[..., Any] | None = None,
    priority: DispatcherPriority = DispatcherPriority.DEFAULT,
) -> Any:
    """Register `kernel` for `op_name`, used whenever `predicate` accepts the call's arguments."""
    if kernel is not None:
        return _insert(op_name, predicate, kernel, priority)

    def decorate(fn: Callable[..., Any]) -> Callable[..., Any]:
        _insert(op_name, predicate, fn, priority)
        return fn

    return decorate


def dispatch(op_name: str, *args: Any, **kwargs: Any) -> Callable[..., Any] | None:
    """First registered kernel for `op_name` whose predicate is truthy for the arguments."""
    for item in _DISPATCHER.get(op_name, ()):
        if item.predicate(*args, **kwargs):
            return item.fn
    return None
