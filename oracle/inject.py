"""TEST INFRASTRUCTURE — lets the host logic of fastforward_amd run on CPU tensors with the C oracle as the checker.

The product has no CPU path: ``fastforward_amd.ops._base._prepare`` (the device check every entry point of the ``ops`` package looks up at call
time) refuses host tensors and the dispatcher predicates of
``fastforward_amd.fused_linear`` accept HIP tensors only. ``use_oracle`` replaces those two seams for the duration of a
``with`` block so that tests (tests/conftest.py), ``__graft_entry__.smoke()`` and nothing else can drive the same Python
code with ``oracle/_build/libffq_oracle.so`` on host pointers. Nothing under ``fastforward_amd/`` imports this module.
"""

from __future__ import annotations

import contextlib

import torch


_PRODUCT = None


@contextlib.contextmanager
def use_library(lib):
    """Make `lib` (an FFQLibrary) the library fastforward_amd.ops calls into. A HIP library needs nothing else; the
    oracle (host pointers, no stream) additionally swaps the device checks of ops / fused_linear for host ones."""
    from fastforward_amd import _native, fused_linear, ops
    from fastforward_amd.exceptions import BackendError

    global _PRODUCT
    if _PRODUCT is None:  # the product's own functions, captured before the first injection
        _PRODUCT = (ops._base._prepare, fused_linear._on_backend)
    previous = (_native._LIB, ops._base._prepare, fused_linear._on_backend)
    _native._LIB = lib
    ops._base._prepare, fused_linear._on_backend = _PRODUCT  # a HIP library runs the product code as shipped (also when nested)
    if not lib.backend_name.startswith("hip"):

        def prepare_host(*tensors):
            device = None
            for t in tensors:
                if t is None:
                    continue
                if device is None:
                    device = t.device
                elif t.device != device:
                    raise RuntimeError(f"Expected all tensors to be on the same device, but found at least two devices, {device} and {t.device}!")
            assert device is not None
            if device.type != "cpu":
                raise BackendError("the injected oracle computes on host memory only")
            return lib, None

        ops._base._prepare = prepare_host
        fused_linear._on_backend = lambda *tensors: all(t.device.type == "cpu" for t in tensors)
    try:
        yield lib
    finally:
        _native._LIB, ops._base._prepare, fused_linear._on_backend = previous
