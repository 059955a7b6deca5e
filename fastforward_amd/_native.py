"""Loader of the HIP backend library (``csrc/libffq_hip.so``).

There is deliberately no CPU fallback: if the library is missing or cannot be loaded every
operator of this package raises :class:`~fastforward_amd.exceptions.BackendError`. Build it with
``python -c "import __graft_entry__ as g; g.build()"`` or ``make -C fastforward_amd/csrc``.
"""

from __future__ import annotations

import pathlib

from fastforward_amd._cabi import FFQLibrary
from fastforward_amd.exceptions import BackendError

LIBRARY_PATH = pathlib.Path(__file__).resolve().parent / "csrc" / "libffq_hip.so"

_LIB: FFQLibrary | None = None


def library() -> FFQLibrary:
    """The loaded backend; raises BackendError when it is unavailable."""
    global _LIB
    if _LIB is None:
        if not LIBRARY_PATH.exists():
            raise BackendError(
                f"HIP backend library not found at {LIBRARY_PATH}. It must be built for gfx950 "
                "(make -C fastforward_amd/csrc); fastforward_amd has no CPU fallback."
            )
        try:
            _LIB = FFQLibrary(LIBRARY_PATH)
        except OSError as e:
            raise BackendError(f"cannot load {LIBRARY_PATH}: {e}") from e
    return _LIB


def is_available() -> bool:
    """True if the backend library can be loaded (says nothing about a GPU being present)."""
    try:
        library()
    except BackendError:
        return False
    return True
