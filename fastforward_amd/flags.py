"""Process-global boolean switches with setter / getter / context-manager access.

Mirrors the behaviour of the reference's ``flags.py`` (src/fastforward/flags.py:25-58,84-102) for
the two flags the hot path consults: ``strict_quantization`` (default True, :84) and
``export_mode`` (default False, :90). ``set_<flag>(v)`` changes the value immediately *and* returns
a context manager that restores the previous value on exit, exactly like the reference.
"""

from __future__ import annotations

import functools

from typing import Callable, ParamSpec, TypeVar

_P = ParamSpec("_P")
_T = TypeVar("_T")

_VALUES: dict[str, bool] = {}


class _Restore:
    """Sets a flag on construction; puts the old value back when used as a context manager."""

    def __init__(self, name: str, value: bool) -> None:
        self._name = name
        self._previous = _VALUES[name]
        _VALUES[name] = bool(value)

    def __enter__(self) -> None:
        return None

    def __exit__(self, *exc: object) -> None:
        _VALUES[self._name] = self._previous


def _define(name: str, default: bool) -> tuple[Callable[[bool], _Restore], Callable[[], bool]]:
    if name in _VALUES:
        raise ValueError(f"Flag '{name}' already exists")
    _VALUES[name] = default

    def setter(value: bool) -> _Restore:
        return _Restore(name, value)

    def getter() -> bool:
        return _VALUES[name]

    setter.__name__ = setter.__qualname__ = f"set_{name}"
    getter.__name__ = getter.__qualname__ = f"get_{name}"
    return setter, getter


set_strict_quantization, get_strict_quantization = _define("strict_quantization", True)
strict_quantization = set_strict_quantization

set_export_mode, get_export_mode = _define("export_mode", False)
export_mode = set_export_mode


def context(flag: Callable[[bool], _Restore], value: bool) -> Callable[[Callable[_P, _T]], Callable[_P, _T]]:
    """Decorator: run the function with `flag` set to `value` (reference flags.py:61-81)."""

    def decorate(func: Callable[_P, _T]) -> Callable[_P, _T]:
        @functools.wraps(func)
        def inner(*args: _P.args, **kwargs: _P.kwargs) -> _T:
            with flag(value):
                return func(*args, **kwargs)

        return inner

    return decorate
