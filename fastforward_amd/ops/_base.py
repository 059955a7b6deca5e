"""What every family of entry points shares: dtype tags, the device check that hands out (library, stream), pointers / tilings, the
caller-side scratch (workspaces, ticket words, extrema words). ``_prepare`` is the ONE seam tests substitute (oracle/inject.py) to drive
the same Python code with the oracle on host memory; everything else in the package looks it up through this module at call time."""

from __future__ import annotations

from typing import Sequence

import torch

from fastforward_amd import _native
from fastforward_amd._cabi import DType, Tiling
from fastforward_amd.exceptions import BackendError


_TAGS: dict[torch.dtype, int] = {
    torch.float32: DType.F32,
    torch.bfloat16: DType.BF16,
    torch.float16: DType.F16,
    torch.float64: DType.F64,
    torch.int8: DType.I8,
    torch.int16: DType.I16,
    torch.int32: DType.I32,
    torch.int64: DType.I64,
    torch.uint8: DType.U8,
}


_DTYPES = {int(tag): dtype for dtype, tag in _TAGS.items()}


def _tag(dtype: torch.dtype) -> int:
    try:
        return int(_TAGS[dtype])
    except KeyError:
        raise NotImplementedError(f"fastforward_amd: dtype {dtype} is not supported by the HIP backend") from None


class _OnDevice:
    """The backend library bound to a HIP device that is not the thread's current one: every C-ABI call runs under
    ``torch.cuda.device(index)`` (kernels launch on the device of the stream they are given; ``hipFuncSetAttribute`` and
    the launch itself act on the CURRENT device)."""

    def __init__(self, lib, index: int) -> None:
        self._lib, self._index = lib, index

    def __getattr__(self, name: str):
        attr = getattr(self._lib, name)
        if not name.startswith("ffq_"):
            return attr

        def call(*args):
            with torch.cuda.device(self._index):
                return attr(*args)

        return call


def _prepare(*tensors: torch.Tensor | None):
    """Check that all tensors live on one HIP device; return (library, stream handle of torch's current stream there).
    There is no CPU implementation: host tensors raise BackendError."""
    lib = _native.library()
    device = None
    for t in tensors:
        if t is None:
            continue
        if device is None:
            device = t.device
        elif t.device != device:
            raise RuntimeError(
                f"Expected all tensors to be on the same device, but found at least two devices, {device} and {t.device}!"
            )
    assert device is not None
    if device.type != "cuda":
        raise BackendError(
            f"fastforward_amd's kernels run on the HIP device only (tensor on '{device}'); there is no CPU "
            "implementation of this entry point. Move the tensors to 'cuda'."
        )
    stream = torch.cuda.current_stream(device).cuda_stream
    if device.index is not None and device.index != torch.cuda.current_device():
        lib = _OnDevice(lib, device.index)
    return lib, stream


_PRODUCT_PREPARE = _prepare  # (tests substitute `_base._prepare` to drive the package with the oracle on host memory: oracle/inject.py)

# True once csrc/libffq_torch.so (the C++ dispatch-key kernels) has been loaded: set by fastforward_amd.ops.registry at import
NATIVE_DISPATCH: bool = False


def _host_route(t: torch.Tensor) -> bool:
    """True for a tensor in HOST memory when the product's own library is in use: the operator then runs the reference's
    device-agnostic ATen chain (``fastforward_amd/_host.py`` — BASELINE configs[0], the reference's default ``device="cpu"``).
    Decided by the tensor's device alone: a HIP tensor never takes it, and a HIP tensor without the library still raises."""
    return t.device.type == "cpu" and _prepare is _PRODUCT_PREPARE


def _native_route(t: torch.Tensor) -> bool:
    """True when the C++ dispatch-key kernels of libffq_torch.so serve this tensor: a HIP tensor, the extension loaded, and
    the library in use the shipped one the extension is linked against (tools/ and tests may select another build or the oracle
    through ``_native._LIB``: those go through ctypes, i.e. through whatever library that is)."""
    return NATIVE_DISPATCH and t.is_cuda and (_native._LIB is None or _native._LIB.path == str(_native.LIBRARY_PATH))


def _ptr(t: torch.Tensor | None) -> int | None:
    return None if t is None else t.data_ptr()


def _tile_of(data: torch.Tensor, tile_size: Sequence[int]) -> Tiling:
    return Tiling.make(tuple(data.shape), tuple(int(v) for v in tile_size))


def _flat(t: torch.Tensor | None) -> torch.Tensor | None:
    if t is None:
        return None
    return t.detach().reshape(-1).contiguous()


def _workspace(nbytes: int, device: torch.device) -> torch.Tensor | None:
    if nbytes <= 0:
        return None
    return torch.empty(nbytes, dtype=torch.uint8, device=device)


# Arrival counters of the split-K launches (ffq_linear_wq / ffq_mlp_gate_up_wq, include/ffq.h): zero before the first launch,
# left zero by every launch, so ONE buffer per (device, stream) serves every EAGER call enqueued on that stream — valid only for
# launches serialised on that stream. Launches captured into a hipGraph get a buffer owned by that graph (below).
_TICKETS: dict[tuple[str, int, int], torch.Tensor] = {}


def _tickets(count: int, device: torch.device, stream: int, kind: str = "wq") -> torch.Tensor | None:
    if count <= 0:
        return None
    if device.type == "cuda" and torch.cuda.is_current_stream_capturing():
        # a buffer of the graph's own (allocated from its pool, zeroed by a memset node of the capture): a graph replayed on another
        # stream, or two graphs captured on one stream and replayed concurrently, must not share counters with eager launches
        return torch.zeros(count, dtype=torch.int32, device=device)
    key = (kind, device.index if device.index is not None else torch.cuda.current_device(), int(stream or 0))
    have = _TICKETS.get(key)
    if have is None or have.numel() < count:
        if have is not None and torch.cuda.is_current_stream_capturing():
            raise RuntimeError("the split-K ticket buffer would have to grow inside a hipGraph capture: run the shape once before capturing")
        have = torch.zeros(max(count, 4096), dtype=torch.int32, device=device)
        _TICKETS[key] = have
    return have


# The three accumulator words + the arrival counter of a producer launch that also leaves [min, max] of its output
# (csrc/ffq_extrema.h): {0xFFFFFFFF, 0, 0, 0} before the first launch, put back by every launch — one buffer per (device, stream)
# for eager launches, a fresh one inside a hipGraph capture.
_EXTREMA_WORDS: dict[tuple[int, int], torch.Tensor] = {}


def _extrema_words(device: torch.device, stream: int) -> torch.Tensor:
    capturing = device.type == "cuda" and torch.cuda.is_current_stream_capturing()
    index = device.index if device.index is not None else (torch.cuda.current_device() if device.type == "cuda" else -1)
    key = (index, int(stream or 0))
    have = None if capturing else _EXTREMA_WORDS.get(key)
    if have is None:
        have = torch.zeros(4, dtype=torch.int32, device=device)
        have[:1].fill_(-1)  # (a fill kernel: capturable, unlike an assignment from a host scalar)
        if not capturing:
            _EXTREMA_WORDS[key] = have
    return have
