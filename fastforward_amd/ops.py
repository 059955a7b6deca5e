"""Tensor-level entry points of the hot path: torch tensors in, C-ABI calls out.

These functions are the bodies of the torch custom ops ``fastforward_amd::quantize_by_tile``,
``dequantize_by_tile``, ``quantize_dynamic_by_tile`` and ``quantize_by_tile_backward`` — the same
four schemas the reference registers under ``fastforward::`` (reference:
src/fastforward/quantization/_quantizer_impl.py:144-285) — plus the reduction / parameter / packing /
linear kernels the reference expresses as ATen chains. Every call is one enqueue on torch's current
HIP stream through the C ABI of ``include/ffq.h``: no host synchronisation, no hidden allocation in
the library (outputs and scratch are torch allocations), legal under hipGraph capture.
"""

from __future__ import annotations

import ctypes

from typing import Any, Sequence

import torch

from fastforward_amd import _host, _native
from fastforward_amd._cabi import FFQ_MAX_BATCH, FLAG_INF, FLAG_NAN, DType, FanOut, RowsBatch, Tiling
from fastforward_amd.exceptions import BackendError

__all__ = [
    "quantize_by_tile",
    "dequantize_by_tile",
    "quantize_dynamic_by_tile",
    "quantize_by_tile_backward",
    "minmax_by_tile",
    "running_minmax_step",
    "parameters_for_range",
    "pack_int4",
    "unpack_int4",
    "pack_q4_0_blocks",
    "pack_q8_0_blocks",
    "quantize_pack_int4",
    "unpack_dequantize_int4",
    "gptq_block",
    "grid_sqerror_by_tile",
    "linear_w8a8",
    "bmm_w8a8",
    "linear_wq",
    "linear_wq_multi",
    "mlp_gate_up_w8a8",
    "add_rmsnorm_quantize",
    "silu_mul_quantize",
    "rope_",
    "quantize_rows_rowsum",
    "quantize_rows_batch",
    "attention",
    "FLAG_INF",
    "FLAG_NAN",
]

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


_PRODUCT_PREPARE = _prepare  # (tests substitute `_prepare` to drive this module with the oracle on host memory: oracle/inject.py)


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


def quantize_by_tile(
    data: torch.Tensor,
    scale: torch.Tensor,
    tile_size: Sequence[int],
    num_bits: float,
    output_dtype: torch.dtype | None,
    offset: torch.Tensor | None = None,
) -> torch.Tensor:
    """A1 — ``fastforward::quantize_by_tile`` (reference _quantizer_impl.py:144-169)."""
    if _host_route(data):
        return _host.quantize_by_tile(data.detach(), scale.detach(), tile_size, num_bits, output_dtype, None if offset is None else offset.detach())
    data_c = data.detach().contiguous()
    scale_c, offset_c = _flat(scale), _flat(offset)
    lib, stream = _prepare(data_c, scale_c, offset_c)
    tiling = _tile_of(data_c, tile_size)
    if output_dtype is None:
        # `output_dtype or result.dtype`: the dtype the eager chain ends in             (:164)
        div = lib.ffq_promote_types(_tag(data_c.dtype), _tag(scale_c.dtype))
        if div not in (DType.F32, DType.BF16, DType.F16, DType.F64):
            div = DType.F32
        sub = lib.ffq_promote_types(div, _tag((offset_c if offset_c is not None else scale_c).dtype))
        output_dtype = _DTYPES[sub]
    out = torch.empty(data_c.shape, dtype=output_dtype, device=data_c.device)
    lib.check(
        lib.ffq_quantize_by_tile(
            _ptr(data_c), _tag(data_c.dtype), _ptr(scale_c), _tag(scale_c.dtype), scale_c.numel(),
            _ptr(offset_c), _tag(offset_c.dtype) if offset_c is not None else 0,
            offset_c.numel() if offset_c is not None else 0,
            ctypes.byref(tiling), float(num_bits), _ptr(out), _tag(output_dtype), stream,
        )
    )
    return out


def dequantize_by_tile(
    data: torch.Tensor,
    scale: torch.Tensor,
    tile_size: Sequence[int],
    offset: torch.Tensor | None = None,
    output_dtype: torch.dtype | None = None,
) -> torch.Tensor:
    """A2 — ``fastforward::dequantize_by_tile`` (reference _quantizer_impl.py:172-190)."""
    if _host_route(data):
        return _host.dequantize_by_tile(data.detach(), scale.detach(), tile_size, None if offset is None else offset.detach(), output_dtype)
    data_c = data.detach().contiguous()
    scale_c, offset_c = _flat(scale), _flat(offset)
    lib, stream = _prepare(data_c, scale_c, offset_c)
    tiling = _tile_of(data_c, tile_size)
    if output_dtype is None:
        tag = lib.ffq_dequantize_result_dtype(
            _tag(data_c.dtype), _tag(scale_c.dtype),
            _tag(offset_c.dtype) if offset_c is not None else 0, int(offset_c is not None),
        )
        output_dtype = _DTYPES[tag]
    out = torch.empty(data_c.shape, dtype=output_dtype, device=data_c.device)
    lib.check(
        lib.ffq_dequantize_by_tile(
            _ptr(data_c), _tag(data_c.dtype), _ptr(scale_c), _tag(scale_c.dtype), scale_c.numel(),
            _ptr(offset_c), _tag(offset_c.dtype) if offset_c is not None else 0,
            offset_c.numel() if offset_c is not None else 0,
            ctypes.byref(tiling), _ptr(out), _tag(output_dtype), stream,
        )
    )
    return out


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


def quantize_dynamic_by_tile(
    data: torch.Tensor,
    tile_size: Sequence[int],
    num_bits: float,
    symmetric: bool,
    allow_one_sided: bool,
    output_dtype: torch.dtype | None,
) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """A3 — ``fastforward::quantize_dynamic_by_tile`` (reference _quantizer_impl.py:243-285)."""
    if _host_route(data):
        return _host.quantize_dynamic_by_tile(data.detach(), tile_size, num_bits, symmetric, allow_one_sided, output_dtype)
    data_c = data.detach().contiguous()
    lib, stream = _prepare(data_c)
    tiling = _tile_of(data_c, tile_size)
    ntiles = lib.ffq_num_tiles(ctypes.byref(tiling))
    if ntiles < 0:
        lib.check(-ntiles)
    if output_dtype is None:
        output_dtype = data_c.dtype if data_c.dtype in (torch.float32, torch.float64) else torch.float32
    out = torch.empty(data_c.shape, dtype=output_dtype, device=data_c.device)
    scale = torch.empty(ntiles, dtype=torch.float32, device=data_c.device)
    offset = torch.empty(ntiles, dtype=torch.float32, device=data_c.device)
    nbytes = lib.ffq_quantize_dynamic_workspace_bytes(ctypes.byref(tiling), _tag(data_c.dtype))
    ws = _workspace(nbytes, data_c.device)
    # per-tensor: A5 in the reduction's last block; symmetric with the one-sided fallback: the two words of the guess / settle launches
    ticket = _tickets(2, data_c.device, stream, kind="minmax") if data_c.is_cuda and (ntiles == 1 or (symmetric and allow_one_sided)) else None
    lib.check(
        lib.ffq_quantize_dynamic_by_tile(
            _ptr(data_c), _tag(data_c.dtype), ctypes.byref(tiling), float(num_bits), int(symmetric),
            int(allow_one_sided), _ptr(out), _tag(output_dtype), _ptr(scale), _ptr(offset), _ptr(ws), nbytes, _ptr(ticket), stream,
        )
    )
    return out, scale, offset


def quantize_by_tile_backward(
    data: torch.Tensor,
    output_grad: torch.Tensor,
    scale: torch.Tensor,
    tile_size: Sequence[int],
    num_bits: float,
    offset: torch.Tensor | None = None,
) -> list[torch.Tensor]:
    """A8 — ``fastforward::quantize_by_tile_backward`` (reference _quantizer_impl.py:193-237).

    Gradients of quantize -> dequantize: clipped elements pass no data gradient; d/dscale is
    ``round(u) - u`` inside the grid and the clip bound plus the rounded offset outside; d/doffset
    is ``scale * grad`` on clipped elements only. One HIP pass + a deterministic finalize for
    per-tensor and contiguous-run tilings with fp32 parameters; other tilings / dtypes take
    :func:`_quantize_by_tile_backward_composite` (device tensor ops, same formulas).
    """
    fast = (
        not _host_route(data)  # host tensors: the composite below IS the reference's chain
        and data.dtype == output_grad.dtype
        and data.dtype in (torch.float32, torch.bfloat16, torch.float16)
        and scale.dtype == torch.float32
        and (offset is None or offset.dtype == torch.float32)
        and data.shape == output_grad.shape
    )
    if fast:
        data_c, grad_c = data.detach().contiguous(), output_grad.detach().contiguous()
        scale_c, offset_c = _flat(scale), _flat(offset)
        lib, stream = _prepare(data_c, grad_c, scale_c, offset_c)
        tiling = _tile_of(data_c, tile_size)
        ntiles = lib.ffq_num_tiles(ctypes.byref(tiling))
        if ntiles < 0:
            lib.check(-ntiles)
        dinput = torch.empty_like(data_c)
        dscale = torch.empty(ntiles, dtype=torch.float32, device=data_c.device)
        doffset = None if offset_c is None else torch.empty(ntiles, dtype=torch.float32, device=data_c.device)
        nbytes = lib.ffq_quantize_backward_workspace_bytes(ctypes.byref(tiling))
        ws = _workspace(nbytes, data_c.device)
        status = lib.ffq_quantize_by_tile_backward(
            _ptr(data_c), _ptr(grad_c), _tag(data_c.dtype), _ptr(scale_c), scale_c.numel(), _ptr(offset_c),
            offset_c.numel() if offset_c is not None else 0, ctypes.byref(tiling), float(num_bits), _ptr(dinput), _ptr(dscale),
            _ptr(doffset), _ptr(ws), nbytes, stream,
        )
        if status == 0:
            # no offset: an empty placeholder on the inputs' device (the reference returns a bare torch.Tensor(), :221-222)
            return [dinput, dscale.reshape(scale.shape), scale.new_empty(0) if doffset is None else doffset.reshape(scale.shape)]
        if status != 6:  # FFQ_ERR_DTYPE: a tiling the kernel does not cover
            lib.check(status)
    return _quantize_by_tile_backward_composite(data, output_grad, scale, tile_size, num_bits, offset)


def _quantize_by_tile_backward_composite(
    data: torch.Tensor,
    output_grad: torch.Tensor,
    scale: torch.Tensor,
    tile_size: Sequence[int],
    num_bits: float,
    offset: torch.Tensor | None = None,
) -> list[torch.Tensor]:
    """The same gradients as a composition of tensor ops on the tensors' own device, op for op as the
    reference writes them (_quantizer_impl.py:203-237): strided channels, N-d tiles, half-precision
    parameters."""
    from fastforward_amd.quantization.tiled_tensor import rows_to_tiles, tiles_to_rows

    param_shape = scale.shape
    s = scale.reshape(-1)
    o = torch.round(offset.reshape(-1)) if offset is not None else torch.zeros_like(s)  # _infer_offset (:140-141)
    tile = torch.Size(tile_size)
    lo = -(2 ** (num_bits - 1))
    hi = -lo - 1
    rows = tiles_to_rows(data, tile)
    grows = tiles_to_rows(output_grad, tile)
    u = rows / s[:, None] - o[:, None]
    q = torch.round(u)
    below, above = q < lo, q > hi
    clipped = below | above
    dinput = rows_to_tiles(torch.where(clipped, torch.zeros_like(grows), grows), data.shape, tile)
    if offset is None:
        doffset = scale.new_empty(0)
    else:
        doffset = torch.where(clipped, s[:, None] * grows, torch.zeros_like(s[:, None] * grows)).sum(1).reshape(param_shape)
    bound = torch.where(below, s.new_tensor([lo]), s.new_tensor([hi])) + o[:, None].to(s.dtype)
    dscale = torch.where(clipped, bound, (q - u).to(s.dtype)) * grows
    return [dinput, dscale.sum(1).reshape(param_shape), doffset]


def minmax_by_tile(
    data: torch.Tensor,
    tile_size: Sequence[int],
    running_min: torch.Tensor | None = None,
    running_max: torch.Tensor | None = None,
    status_flags: torch.Tensor | None = None,
    into: tuple[torch.Tensor, torch.Tensor] | None = None,
) -> tuple[torch.Tensor, torch.Tensor]:
    """A4 — per-tile (min, max) of `data` in the data dtype (reference minmax.py:227-237).

    With `running_min` / `running_max` given they are updated IN PLACE (running min / running max)
    and returned. `status_flags` (int32[1] on the data's device) is OR-ed with FLAG_INF / FLAG_NAN
    for this batch so the caller can decide when to look at it; nothing here waits for the device.
    `into` = (min, max) buffers for THIS batch's extrema (overwritten, not merged).
    """
    if _host_route(data):
        lo, hi = _host.minmax_by_tile(data.detach(), tile_size)
        if status_flags is not None:
            flag = (FLAG_INF if bool(lo.isinf().any() or hi.isinf().any()) else 0) | (FLAG_NAN if bool(lo.isnan().any() or hi.isnan().any()) else 0)
            status_flags.bitwise_or_(torch.tensor([flag], dtype=status_flags.dtype))
        if running_min is not None:
            assert running_max is not None
            running_min.copy_(torch.min(running_min, lo.to(running_min.dtype)))  # torch.min / torch.max propagate NaN (minmax.py:236-237)
            running_max.copy_(torch.max(running_max, hi.to(running_max.dtype)))
            return running_min, running_max
        if into is not None:
            into[0].copy_(lo)
            into[1].copy_(hi)
            return into
        return lo, hi
    data_c = data.detach().contiguous()
    lib, stream = _prepare(data_c, running_min, running_max, status_flags)
    tiling = _tile_of(data_c, tile_size)
    ntiles = lib.ffq_num_tiles(ctypes.byref(tiling))
    if ntiles < 0:
        lib.check(-ntiles)
    accumulate = running_min is not None
    if accumulate:
        assert running_max is not None
        mn, mx = running_min, running_max
        if mn.numel() != ntiles or mx.numel() != ntiles or mn.dtype != data_c.dtype or mx.dtype != data_c.dtype:
            raise RuntimeError(
                f"running min/max must hold {ntiles} values of dtype {data_c.dtype}, got "
                f"{mn.numel()} x {mn.dtype}"
            )
        if not (mn.is_contiguous() and mx.is_contiguous()):
            raise RuntimeError("running min/max must be contiguous")
    elif into is not None:
        mn, mx = into
        if not (mn.numel() == mx.numel() == ntiles and mn.dtype == mx.dtype == data_c.dtype and mn.is_contiguous() and mx.is_contiguous()):
            raise RuntimeError(f"`into` must be two contiguous buffers of {ntiles} values of dtype {data_c.dtype}")
    else:
        mn = torch.empty(ntiles, dtype=data_c.dtype, device=data_c.device)
        mx = torch.empty(ntiles, dtype=data_c.dtype, device=data_c.device)
    nbytes = lib.ffq_minmax_workspace_bytes(ctypes.byref(tiling), _tag(data_c.dtype))
    ws = _workspace(nbytes, data_c.device)
    ticket = _tickets(1, data_c.device, stream, kind="minmax") if data_c.is_cuda and ntiles == 1 else None  # per-tensor: one launch
    lib.check(
        lib.ffq_minmax_by_tile(
            _ptr(data_c), _tag(data_c.dtype), ctypes.byref(tiling), _ptr(mn), _ptr(mx), int(accumulate),
            _ptr(status_flags), _ptr(ws), nbytes, _ptr(ticket), stream,
        )
    )
    return mn, mx


def running_minmax_step(
    data: torch.Tensor,
    tile_size: Sequence[int],
    running_min: torch.Tensor,
    running_max: torch.Tensor,
    status_flags: torch.Tensor | None,
    num_bits: float,
    symmetric: bool,
    allow_one_sided: bool,
    scale_out: torch.Tensor,
    offset_out: torch.Tensor | None,
) -> None:
    """One ``RunningMinMaxEstimator.estimate_step`` (reference range_setting/minmax.py:215-239) without leaving the device:
    A4 merged into `running_min` / `running_max` in place, then A5 of the merged range (the quantization_range setter,
    nn/linear_quantizer.py:350-357) written into `scale_out` / `offset_out` — what :func:`minmax_by_tile` with running
    buffers followed by :func:`parameters_for_range` gives, bit for bit; a per-tensor quantizer takes ONE launch."""
    if _host_route(data):
        minmax_by_tile(data, tile_size, running_min=running_min, running_max=running_max, status_flags=status_flags)
        parameters_for_range(running_min, running_max, num_bits, symmetric, allow_one_sided, scale_out, offset_out, want_offset=offset_out is not None)
        return
    if _native_route(data):  # dispatcher -> C++ (csrc/ffq_torch.cpp) -> C ABI: 448 calls per calibration step of Llama-3-8B
        torch.ops.fastforward_amd.running_minmax_step(data, list(tile_size), running_min, running_max, status_flags, float(num_bits), bool(symmetric),
                                                      bool(allow_one_sided), scale_out, offset_out)
        return
    _running_minmax_step(data, tile_size, running_min, running_max, status_flags, num_bits, symmetric, allow_one_sided, scale_out, offset_out)


def _running_minmax_step(data, tile_size, running_min, running_max, status_flags, num_bits, symmetric, allow_one_sided, scale_out, offset_out) -> None:  # type: ignore[no-untyped-def]
    """Python implementation of the ``running_minmax_step`` operator (Python -> ctypes -> C ABI)."""
    data_c = data.detach().contiguous()
    lib, stream = _prepare(data_c, running_min, running_max, status_flags, scale_out, offset_out)
    tiling = _tile_of(data_c, tile_size)
    ntiles = lib.ffq_num_tiles(ctypes.byref(tiling))
    if ntiles < 0:
        lib.check(-ntiles)
    for t in (running_min, running_max):
        if t.numel() != ntiles or t.dtype != data_c.dtype or not t.is_contiguous():
            raise RuntimeError(f"running min/max must hold {ntiles} contiguous values of dtype {data_c.dtype}")
    if scale_out.numel() != ntiles or not scale_out.is_contiguous() or (offset_out is not None and (offset_out.numel() != ntiles or not offset_out.is_contiguous())):
        raise RuntimeError(f"scale / offset must hold {ntiles} contiguous values")
    nbytes = lib.ffq_minmax_workspace_bytes(ctypes.byref(tiling), _tag(data_c.dtype))
    ws = _workspace(nbytes, data_c.device)
    ticket = _tickets(1, data_c.device, stream, kind="minmax") if data_c.is_cuda and ntiles == 1 else None
    lib.check(
        lib.ffq_running_minmax_step(
            _ptr(data_c), _tag(data_c.dtype), ctypes.byref(tiling), _ptr(running_min), _ptr(running_max), _ptr(status_flags),
            float(num_bits), int(symmetric), int(allow_one_sided), _ptr(scale_out), _tag(scale_out.dtype),
            _ptr(offset_out), _tag(offset_out.dtype) if offset_out is not None else 0, _ptr(ws), nbytes, _ptr(ticket), stream,
        )
    )


def running_minmax_quantize(
    data: torch.Tensor,
    tile_size: Sequence[int],
    running_min: torch.Tensor,
    running_max: torch.Tensor,
    status_flags: torch.Tensor | None,
    num_bits: float,
    symmetric: bool,
    allow_one_sided: bool,
    scale_out: torch.Tensor,
    offset_out: torch.Tensor,
    output_dtype: torch.dtype,
) -> torch.Tensor | None:
    """:func:`running_minmax_step` AND ``quantize_by_tile(data, scale_out, tile_size, num_bits, output_dtype, offset_out)`` in one
    pass over `data` (C ABI ``ffq_running_minmax_quantize``): what ``estimate_ranges`` runs per quantizer call with a RunningMinMax
    estimator (reference range_setting/common.py:218-238). Returns the codes, or None — nothing written — where the one-pass kernel
    does not apply (host tensors, one tile, tiles that are not short contiguous runs, parameters that are not contiguous fp32):
    take the two calls."""
    if _host_route(data) or not data.is_contiguous() or running_min.dtype != data.dtype or running_max.dtype != data.dtype:
        return None
    if scale_out.dtype != torch.float32 or offset_out.dtype != torch.float32 or not scale_out.is_contiguous() or not offset_out.is_contiguous():
        return None
    if not running_min.is_contiguous() or not running_max.is_contiguous():
        return None
    data_c = data.detach()
    lib, stream = _prepare(data_c, running_min, running_max, scale_out, offset_out, status_flags)
    tiling = _tile_of(data_c, tile_size)
    ntiles = lib.ffq_num_tiles(ctypes.byref(tiling))
    if ntiles < 0:
        lib.check(-ntiles)
    if ntiles <= 1 or running_min.numel() != ntiles or running_max.numel() != ntiles or scale_out.numel() != ntiles or offset_out.numel() != ntiles:
        return None
    out = torch.empty(data_c.shape, dtype=output_dtype, device=data_c.device)
    ticket = _tickets(2, data_c.device, stream, kind="minmax") if symmetric and allow_one_sided else None
    status = lib.ffq_running_minmax_quantize(
        _ptr(data_c), _tag(data_c.dtype), ctypes.byref(tiling), _ptr(running_min), _ptr(running_max), _ptr(status_flags), float(num_bits),
        int(symmetric), int(allow_one_sided), _ptr(scale_out), _ptr(offset_out), _ptr(out), _tag(output_dtype), _ptr(ticket), stream,
    )
    if status == 6:  # FFQ_ERR_DTYPE: outside the one-pass kernel; no buffer was touched
        return None
    lib.check(status)
    return out


def parameters_for_range(
    min_range: torch.Tensor,
    max_range: torch.Tensor,
    num_bits: float,
    symmetric: bool,
    allow_one_sided: bool,
    scale_out: torch.Tensor | None = None,
    offset_out: torch.Tensor | None = None,
    want_offset: bool = True,
) -> tuple[torch.Tensor, torch.Tensor | None]:
    """A5 on the device — see :func:`fastforward_amd.quantization.affine.parameters_for_range`.

    Writes into `scale_out` / `offset_out` when given (the quantizer's own parameters), else into
    fresh fp32 tensors. When the symmetric two-sided branch is taken the offset output holds zeros
    (the reference returns None there and its range setter fills the buffer with 0).
    """
    mn, mx = _flat(min_range), _flat(max_range)
    if _host_route(mn):
        scale, offset = _host.parameters_for_range(mn, mx.to(mn.device), num_bits, symmetric, allow_one_sided)
        if scale_out is None:
            scale_out = torch.empty(mn.numel(), dtype=torch.float32)
        scale_out.reshape(-1).copy_(scale)
        if offset_out is None and want_offset:
            offset_out = torch.empty(mn.numel(), dtype=torch.float32)
        if offset_out is not None:  # the reference returns None for the symmetric two-sided branch; its range setter fills the buffer with 0
            offset_out.reshape(-1).copy_(offset if offset is not None else torch.zeros_like(scale))
        return scale_out, offset_out
    if mn.dtype != mx.dtype:
        common = torch.promote_types(mn.dtype, mx.dtype)
        mn, mx = mn.to(common), mx.to(common)
    lib, stream = _prepare(mn, mx, scale_out, offset_out)
    n = mn.numel()
    if mx.numel() != n:
        raise RuntimeError(f"min_range and max_range must have the same number of elements ({n} vs {mx.numel()})")
    if scale_out is None:
        scale_out = torch.empty(n, dtype=torch.float32, device=mn.device)
    if offset_out is None and want_offset:
        offset_out = torch.empty(n, dtype=torch.float32, device=mn.device)
    for name, t in (("scale", scale_out), ("offset", offset_out)):
        if t is not None and (t.numel() != n or not t.is_contiguous()):
            raise RuntimeError(f"{name} output must be contiguous with {n} elements, got {tuple(t.shape)}")
    nbytes = lib.ffq_parameters_for_range_workspace_bytes(n, int(symmetric), int(allow_one_sided))
    ws = _workspace(nbytes, mn.device)
    lib.check(
        lib.ffq_parameters_for_range(
            _ptr(mn), _ptr(mx), _tag(mn.dtype), n, float(num_bits), int(symmetric), int(allow_one_sided),
            _ptr(scale_out), _tag(scale_out.dtype), _ptr(offset_out),
            _tag(offset_out.dtype) if offset_out is not None else 0, _ptr(ws), nbytes, stream,
        )
    )
    return scale_out, offset_out


def pack_int4(codes: torch.Tensor, block: int = 32) -> torch.Tensor:
    """A7 — pack codes in [-8, 7] two per byte, GGUF Q4_0 nibble order (reference _packing.py:44-53)."""
    codes_c = codes.detach().contiguous()
    lib, stream = _prepare(codes_c)
    n = codes_c.numel()
    out = torch.empty(n // 2, dtype=torch.uint8, device=codes_c.device)
    lib.check(lib.ffq_pack_int4(_ptr(codes_c), _tag(codes_c.dtype), n, int(block), _ptr(out), stream))
    return out


def unpack_int4(packed: torch.Tensor, shape: Sequence[int], dtype: torch.dtype = torch.int8, block: int = 32) -> torch.Tensor:
    """Inverse of :func:`pack_int4`: ``unpack_int4(pack_int4(q), q.shape, q.dtype) == q``."""
    packed_c = packed.detach().contiguous()
    lib, stream = _prepare(packed_c)
    out = torch.empty(tuple(shape), dtype=dtype, device=packed_c.device)
    if out.numel() != packed_c.numel() * 2:
        raise ValueError(f"shape {tuple(shape)} does not hold {packed_c.numel() * 2} codes")
    lib.check(lib.ffq_unpack_int4(_ptr(packed_c), out.numel(), int(block), _ptr(out), _tag(dtype), stream))
    return out


def gptq_block(
    weights: torch.Tensor,
    quantized: torch.Tensor,
    errors: torch.Tensor,
    col0: int,
    block_cols: int,
    hessian_inverse: torch.Tensor,
    scale: torch.Tensor,
    offset: torch.Tensor | None,
    num_bits: float,
) -> bool:
    """GPTQ's column loop for ``weights[:, col0 : col0 + block_cols]`` in one launch (reference
    quantization/gptq.py:101-131): fills the block's columns of `quantized` and `errors` in place. fp32 matrices,
    one (scale, offset) per row or one in total. Returns False when the kernel does not cover the call."""
    tensors = (weights, quantized, errors, hessian_inverse)
    if any(t.dtype != torch.float32 or t.dim() != 2 or not t.is_contiguous() for t in tensors) or block_cols > 128:
        return False
    if not (weights.shape == quantized.shape == errors.shape):
        return False
    sc = scale.detach().reshape(-1).to(torch.float32).contiguous()
    of = None if offset is None else offset.detach().reshape(-1).to(torch.float32).contiguous()
    rows = weights.shape[0]
    if sc.numel() not in (1, rows) or (of is not None and of.numel() not in (1, rows)):
        return False
    lib, stream = _prepare(weights, quantized, errors, hessian_inverse, sc, of)
    lib.check(
        lib.ffq_gptq_block(
            _ptr(weights), _ptr(quantized), _ptr(errors), rows, weights.shape[1], int(col0), int(block_cols),
            _ptr(hessian_inverse), hessian_inverse.shape[1], _ptr(sc), sc.numel(), _ptr(of), of.numel() if of is not None else 0,
            float(num_bits), stream,
        )
    )
    return True


def grid_sqerror_by_tile(
    data: torch.Tensor,
    scales: torch.Tensor,
    offsets: torch.Tensor | None,
    tile_size: Sequence[int],
    num_bits: float,
    out: torch.Tensor | None = None,
) -> torch.Tensor | None:
    """Sum over every tile of ``(dequantize(quantize(data)) - data) ** 2`` for each candidate parameter set
    (``scales`` / ``offsets``: ``[candidates, tiles]`` fp32), all candidates in ONE pass over `data` — the inner loop
    of the min-error grid estimator (reference range_setting/min_error.py:218-231). With `out` given the sums are
    added to it. Returns None when the tiling is outside the kernel's range (the caller loops over A1 / A2)."""
    data_c = data.detach().contiguous()
    sc = scales.detach().to(torch.float32).contiguous()
    of = None if offsets is None else offsets.detach().to(torch.float32).contiguous()
    if data_c.dtype not in (torch.float32, torch.bfloat16, torch.float16) or sc.dim() != 2:
        return None
    lib, stream = _prepare(data_c, sc, of, out)
    tiling = _tile_of(data_c, tile_size)
    ntiles = lib.ffq_num_tiles(ctypes.byref(tiling))
    if ntiles < 0:
        lib.check(-ntiles)
    ncand = sc.shape[0]
    if sc.shape[1] != ntiles or (of is not None and of.shape != sc.shape):
        raise RuntimeError(f"candidate parameters must be [candidates, {ntiles}], got {tuple(sc.shape)}")
    accumulate = out is not None
    if out is None:
        out = torch.empty((ncand, ntiles), dtype=torch.float32, device=data_c.device)
    elif out.shape != sc.shape or out.dtype != torch.float32 or not out.is_contiguous():
        raise RuntimeError("`out` must be a contiguous fp32 [candidates, tiles] tensor")
    nbytes = lib.ffq_grid_sqerror_workspace_bytes(ctypes.byref(tiling), ncand)
    ws = _workspace(nbytes, data_c.device)
    status = lib.ffq_grid_sqerror_by_tile(
        _ptr(data_c), _tag(data_c.dtype), _ptr(sc), _ptr(of), ncand, ctypes.byref(tiling), float(num_bits), _ptr(out),
        int(accumulate), _ptr(ws), nbytes, stream,
    )
    if status == 6:  # FFQ_ERR_DTYPE: tiling not covered
        return None
    lib.check(status)
    return out


def _pack_gguf(int_codes: torch.Tensor, scales: torch.Tensor, fmt: int) -> torch.Tensor:
    codes = int_codes.detach().to(torch.int8).contiguous()
    if codes.dim() != 2 or codes.shape[1] != 32:
        raise ValueError(f"GGUF block-32 formats expect codes of shape (n_blocks, 32), got {tuple(codes.shape)}")
    sc = scales.detach().reshape(-1).to(torch.float32).contiguous()
    if sc.numel() != codes.shape[0]:
        raise RuntimeError(f"expected {codes.shape[0]} scales, got {sc.numel()}")
    lib, stream = _prepare(codes, sc)
    out = torch.empty((codes.shape[0], 18 if fmt == 4 else 34), dtype=torch.uint8, device=codes.device)
    lib.check(lib.ffq_pack_gguf_blocks(_ptr(codes), _ptr(sc), codes.shape[0], fmt, _ptr(out), stream))
    return out


def pack_q4_0_blocks(int_codes: torch.Tensor, scales: torch.Tensor) -> torch.Tensor:
    """``(n_blocks, 32)`` codes in [-8, 7] + per-block scales -> ``(n_blocks, 18)`` raw GGUF Q4_0 bytes
    (reference export/stages/gguf/_packing.py:23-55)."""
    return _pack_gguf(int_codes, scales, 4)


def pack_q8_0_blocks(int_codes: torch.Tensor, scales: torch.Tensor) -> torch.Tensor:
    """``(n_blocks, 32)`` codes in [-128, 127] + per-block scales -> ``(n_blocks, 34)`` raw GGUF Q8_0 bytes, codes
    clipped to [-127, 127] (reference export/stages/gguf/_packing.py:58-79)."""
    return _pack_gguf(int_codes, scales, 8)


def quantize_pack_int4(
    data: torch.Tensor, scale: torch.Tensor, tile_size: Sequence[int], offset: torch.Tensor | None = None, block: int = 32
) -> torch.Tensor:
    """A1 (4 bits) + A7 in one pass: ``pack_int4(quantize_by_tile(data, ..., num_bits=4, int8), block)`` without the
    codes' round trip through HBM. Tilings / dtypes outside the fused kernel's range compose the two steps."""
    data_c = data.detach().contiguous()
    scale_c, offset_c = _flat(scale), _flat(offset)
    fast = scale_c.dtype == torch.float32 and (offset_c is None or offset_c.dtype == torch.float32) and data_c.dtype in (torch.float32, torch.bfloat16, torch.float16)
    if fast:
        lib, stream = _prepare(data_c, scale_c, offset_c)
        tiling = _tile_of(data_c, tile_size)
        out = torch.empty(data_c.numel() // 2, dtype=torch.uint8, device=data_c.device)
        status = lib.ffq_quantize_pack_int4(
            _ptr(data_c), _tag(data_c.dtype), _ptr(scale_c), scale_c.numel(), _ptr(offset_c),
            offset_c.numel() if offset_c is not None else 0, ctypes.byref(tiling), int(block), _ptr(out), stream,
        )
        if status == 0:
            return out
        if status != 6:  # FFQ_ERR_DTYPE: not covered by the fused kernel
            lib.check(status)
    return pack_int4(quantize_by_tile(data, scale, tile_size, 4, torch.int8, offset), block)


def unpack_dequantize_int4(
    packed: torch.Tensor, scale: torch.Tensor, shape: Sequence[int], tile_size: Sequence[int], offset: torch.Tensor | None = None,
    block: int = 32, output_dtype: torch.dtype = torch.bfloat16,
) -> torch.Tensor:
    """A7 + A2 in one pass: ``dequantize_by_tile(unpack_int4(packed, shape, int8, block), ...)``."""
    packed_c = packed.detach().contiguous()
    scale_c, offset_c = _flat(scale), _flat(offset)
    fast = scale_c.dtype == torch.float32 and (offset_c is None or offset_c.dtype == torch.float32) and output_dtype in (torch.float32, torch.bfloat16, torch.float16)
    if fast:
        lib, stream = _prepare(packed_c, scale_c, offset_c)
        out = torch.empty(tuple(shape), dtype=output_dtype, device=packed_c.device)
        if out.numel() != packed_c.numel() * 2:
            raise ValueError(f"shape {tuple(shape)} does not hold {packed_c.numel() * 2} codes")
        tiling = _tile_of(out, tile_size)
        status = lib.ffq_unpack_dequantize_int4(
            _ptr(packed_c), _ptr(scale_c), scale_c.numel(), _ptr(offset_c), offset_c.numel() if offset_c is not None else 0,
            ctypes.byref(tiling), int(block), _ptr(out), _tag(output_dtype), stream,
        )
        if status == 0:
            return out
        if status != 6:
            lib.check(status)
    return dequantize_by_tile(unpack_int4(packed, shape, torch.int8, block), scale, tile_size, offset, output_dtype)


def linear_w8a8(
    x_codes: torch.Tensor,
    w_codes: torch.Tensor,
    x_scale: torch.Tensor,
    x_offset: torch.Tensor | None,
    w_scale: torch.Tensor,
    w_offset: torch.Tensor | None,
    bias: torch.Tensor | None = None,
    out_dtype: torch.dtype = torch.bfloat16,
    out_scale: torch.Tensor | None = None,
    out_offset: torch.Tensor | None = None,
    out_num_bits: float = 8.0,
    w_rowsum: torch.Tensor | None = None,
    requant_from: torch.dtype | None = None,
) -> torch.Tensor:
    """A6 — int8 codes in, real-valued (or re-quantized) linear output out.

    `x_codes` is [..., K] int8, `w_codes` is [N, K] int8. Scales/offsets are fp32 with one entry
    (per-tensor) or one per row (per-token for x, per-output-channel for w). `w_rowsum` (int32 [N], optional): the row
    sums of `w_codes` when the caller already has them (:func:`quantize_rows_rowsum`) — same result, one launch fewer.

    With `out_scale` (and optionally `out_offset`) the output quantizer of reference _gen/fallback.py:110-111 runs in
    the GEMM's epilogue: the linear's result is rounded to `requant_from` (the dtype the reference's float GEMM returns:
    the input's dequantize dtype; default bf16), A1 is applied to it, and `out` holds the codes in the container
    `out_dtype` — exactly ``quantize_by_tile(linear_w8a8(..., out_dtype=requant_from), out_scale, shape, bits, out_dtype,
    out_offset)`` without the real-valued tensor's round trip through HBM.
    """
    if _native_route(x_codes):  # dispatcher -> C++ (csrc/ffq_torch.cpp) -> C ABI
        return torch.ops.fastforward_amd.linear_w8a8(x_codes, w_codes, x_scale, x_offset, w_scale, w_offset, bias, out_dtype, out_scale, out_offset,
                                                     float(out_num_bits), w_rowsum, requant_from)
    return _linear_w8a8(x_codes, w_codes, x_scale, x_offset, w_scale, w_offset, bias, out_dtype, out_scale, out_offset, out_num_bits, w_rowsum, requant_from)


def _linear_w8a8(x_codes, w_codes, x_scale, x_offset, w_scale, w_offset, bias, out_dtype, out_scale, out_offset, out_num_bits, w_rowsum, requant_from):  # type: ignore[no-untyped-def]
    """Python implementation of the ``linear_w8a8`` operator (Python -> ctypes -> C ABI); arguments in schema order."""
    if x_codes.dtype != torch.int8 or w_codes.dtype != torch.int8:
        raise TypeError("linear_w8a8 expects int8 codes")
    xc = x_codes.detach().contiguous()
    wc = w_codes.detach().contiguous()
    K = xc.shape[-1]
    N = wc.shape[0]
    M = xc.numel() // K if K else 0
    if wc.dim() != 2 or wc.shape[1] != K:
        raise RuntimeError(f"mat1 and mat2 shapes cannot be multiplied ({M}x{K} and {tuple(wc.shape)}^T)")

    def f32(t: torch.Tensor | None) -> torch.Tensor | None:
        return None if t is None else t.detach().reshape(-1).to(torch.float32).contiguous()

    xs, xo, ws_, wo, os_, oo = f32(x_scale), f32(x_offset), f32(w_scale), f32(w_offset), f32(out_scale), f32(out_offset)
    bias_c = None if bias is None else bias.detach().contiguous()
    lib, stream = _prepare(xc, wc, xs, xo, ws_, wo, bias_c, os_, oo)
    x_per_row = int(xs.numel() != 1)
    w_per_row = int(ws_.numel() != 1)
    if x_per_row and xs.numel() != M:
        raise RuntimeError(f"activation scale must have 1 or {M} entries, got {xs.numel()}")
    if w_per_row and ws_.numel() != N:
        raise RuntimeError(f"weight scale must have 1 or {N} entries, got {ws_.numel()}")
    out = torch.empty((*xc.shape[:-1], N), dtype=out_dtype, device=xc.device)
    nbytes = lib.ffq_linear_w8a8_workspace_bytes(M, N, K)
    ws = _workspace(nbytes, xc.device)
    if w_rowsum is not None and (w_rowsum.dtype != torch.int32 or w_rowsum.numel() != N or not w_rowsum.is_contiguous() or w_rowsum.device != wc.device):
        raise RuntimeError(f"w_rowsum must be a contiguous int32 tensor with {N} entries on the codes' device")
    y_dt = _tag(requant_from or torch.bfloat16) if os_ is not None else 0
    lib.check(
        lib.ffq_linear_w8a8(
            _ptr(xc), _ptr(wc), _ptr(w_rowsum), _ptr(xs), _ptr(xo), x_per_row, _ptr(ws_), _ptr(wo), w_per_row,
            _ptr(bias_c), _tag(bias_c.dtype) if bias_c is not None else 0, _ptr(out), _tag(out_dtype),
            _ptr(os_), _ptr(oo), float(out_num_bits), y_dt, M, N, K, _ptr(ws), nbytes, stream,
        )
    )
    return out


def quantize_by_tile_unless_same(
    data: torch.Tensor, scale: torch.Tensor, offset: torch.Tensor | None, num_bits: float,
    earlier_scale: torch.Tensor, earlier_offset: torch.Tensor | None,
) -> torch.Tensor | None:
    """A1 of a per-tensor quantizer into an int8 container that does NOTHING where an earlier quantizer of the same tensor holds the
    same parameters — decided on the device from the scale's bits and the rounded offsets (C ABI
    ``ffq_quantize_by_tile_unless_same``): the result is then UNWRITTEN memory, and whoever reads it must be told about the earlier
    codes (:func:`linear_w8a8_earlier`, :func:`mlp_gate_up_w8a8_estimating`), which are this quantizer's codes in that case. Where
    the parameters differ the codes are :func:`quantize_by_tile`'s. None (nothing launched) outside the launch's coverage: fp32
    one-element parameters on the data's device, whole 16-element chunks of f32 / bf16 / f16 data."""
    if _host_route(data) or data.dtype not in (torch.float32, torch.bfloat16, torch.float16) or data.numel() % 16 != 0 or data.numel() == 0:
        return None
    tensors = (scale, offset, earlier_scale, earlier_offset)
    if any(t is not None and (t.dtype != torch.float32 or t.numel() != 1 or t.device != data.device) for t in tensors) or not (float(num_bits) == int(num_bits) and 1 <= num_bits <= 8):
        return None
    data_c = data.detach().contiguous()
    lib, stream = _prepare(data_c, scale, offset, earlier_scale, earlier_offset)
    out = torch.empty(data_c.shape, dtype=torch.int8, device=data_c.device)
    status = lib.ffq_quantize_by_tile_unless_same(
        _ptr(data_c), _tag(data_c.dtype), _ptr(scale.detach()), _ptr(None if offset is None else offset.detach()), data_c.numel(), float(num_bits),
        _ptr(earlier_scale.detach()), _ptr(None if earlier_offset is None else earlier_offset.detach()), _ptr(out), stream,
    )
    if status == 6:  # FFQ_ERR_DTYPE: alignment
        return None
    lib.check(status)
    return out


def linear_w8a8_takes_earlier(M: int, N: int, K: int) -> bool:
    """Whether :func:`linear_w8a8_earlier` covers an [M, K] x [N, K]^T linear (the persistent int8 GEMM's shape class)."""
    return bool(_native.library().ffq_linear_w8a8_takes_earlier(int(M), int(N), int(K)))


def linear_w8a8_earlier(
    x_codes: torch.Tensor,
    earlier: tuple[torch.Tensor, torch.Tensor, torch.Tensor | None],
    w_codes: torch.Tensor,
    x_scale: torch.Tensor,
    x_offset: torch.Tensor | None,
    w_scale: torch.Tensor,
    w_offset: torch.Tensor | None,
    out_dtype: torch.dtype = torch.bfloat16,
    w_rowsum: torch.Tensor | None = None,
) -> torch.Tensor | None:
    """:func:`linear_w8a8` (per-tensor activation parameters, no bias, real-valued output) on activation codes that come from
    :func:`quantize_by_tile_unless_same`: ``earlier = (codes, scale, offset)`` of the earlier quantizer of the same tensor; the
    launch reads those codes where the two parameter pairs are the same (they are this linear's codes then) and `x_codes` where
    they are not. None outside ``linear_w8a8_takes_earlier`` or for per-token parameters — then the caller has to settle which
    codes are in force before anything reads `x_codes`."""
    if x_codes.dtype != torch.int8 or w_codes.dtype != torch.int8 or earlier[0].dtype != torch.int8:
        raise TypeError("linear_w8a8_earlier expects int8 codes")
    xc, wc, ec = x_codes.detach().contiguous(), w_codes.detach().contiguous(), earlier[0].detach().contiguous()
    K, N = xc.shape[-1], wc.shape[0]
    M = xc.numel() // K if K else 0
    if wc.dim() != 2 or wc.shape[1] != K:
        raise RuntimeError(f"mat1 and mat2 shapes cannot be multiplied ({M}x{K} and {tuple(wc.shape)}^T)")
    if ec.shape != xc.shape:
        raise RuntimeError(f"earlier codes of shape {tuple(ec.shape)} for activation codes of shape {tuple(xc.shape)}")

    def f32(t: torch.Tensor | None) -> torch.Tensor | None:
        return None if t is None else t.detach().reshape(-1).to(torch.float32).contiguous()

    xs, xo, ws_, wo, es, eo = f32(x_scale), f32(x_offset), f32(w_scale), f32(w_offset), f32(earlier[1]), f32(earlier[2])
    if xs.numel() != 1 or es.numel() != 1 or ws_.numel() not in (1, N) or out_dtype not in (torch.float32, torch.bfloat16, torch.float16):
        return None
    if w_rowsum is not None and (w_rowsum.dtype != torch.int32 or w_rowsum.numel() != N or not w_rowsum.is_contiguous() or w_rowsum.device != wc.device):
        raise RuntimeError(f"w_rowsum must be a contiguous int32 tensor with {N} entries on the codes' device")
    lib, stream = _prepare(xc, ec, wc, xs, xo, ws_, wo, es, eo)
    if not lib.ffq_linear_w8a8_takes_earlier(M, N, K):
        return None
    out = torch.empty((*xc.shape[:-1], N), dtype=out_dtype, device=xc.device)
    nbytes = lib.ffq_linear_w8a8_workspace_bytes(M, N, K)
    ws = _workspace(nbytes, xc.device)
    lib.check(
        lib.ffq_linear_w8a8_earlier(
            _ptr(xc), _ptr(ec), _ptr(es), _ptr(eo), _ptr(wc), _ptr(w_rowsum), _ptr(xs), _ptr(xo), _ptr(ws_), _ptr(wo), int(ws_.numel() != 1),
            _ptr(out), _tag(out_dtype), M, N, K, _ptr(ws), nbytes, stream,
        )
    )
    return out


def linear_w8a8_gated(
    x_codes: torch.Tensor,
    w_codes: torch.Tensor,
    x_scale: torch.Tensor,
    x_offset: torch.Tensor | None,
    w_scale: torch.Tensor,
    w_offset: torch.Tensor | None,
    gate: torch.Tensor,
    w_rowsum: torch.Tensor | None = None,
    want_extrema: bool = False,
) -> torch.Tensor | tuple[torch.Tensor, torch.Tensor] | None:
    """``silu(gate) * linear(x, w)`` with the linear on int8 codes and the product formed in its epilogue: the second
    projection of a gated MLP (reference docs/examples/doc_helpers/quantized_llama/mlp.py:36-38) when the first one's bf16 result
    `gate` [..., N] is at hand and the product's own quantizer is not yet known (range estimation). Equals
    ``silu_mul_quantize(gate, linear_w8a8(...), (), want_product=True)[0]`` bit for bit. None where the one-launch form does not
    apply (then take those two calls). ``want_extrema``: returns ``(product, pair)`` with ``pair = [min, max]`` of the product in
    bf16 (``minmax_by_tile`` over the whole tensor), left by the same launch: the reduction a RunningMinMax step on the product
    would start with."""
    if x_codes.dtype != torch.int8 or w_codes.dtype != torch.int8:
        raise TypeError("linear_w8a8_gated expects int8 codes")
    xc, wc, gc = x_codes.detach().contiguous(), w_codes.detach().contiguous(), gate.detach().contiguous()
    K, N = xc.shape[-1], wc.shape[0]
    M = xc.numel() // K if K else 0
    if wc.dim() != 2 or wc.shape[1] != K:
        raise RuntimeError(f"mat1 and mat2 shapes cannot be multiplied ({M}x{K} and {tuple(wc.shape)}^T)")
    if gc.dtype != torch.bfloat16 or gc.numel() != M * N or gc.shape[-1] != N or M == 0 or N == 0:
        return None

    def f32(t: torch.Tensor | None) -> torch.Tensor | None:
        return None if t is None else t.detach().reshape(-1).to(torch.float32).contiguous()

    xs, xo, ws_, wo = f32(x_scale), f32(x_offset), f32(w_scale), f32(w_offset)
    if xs.numel() not in (1, M) or ws_.numel() not in (1, N):
        return None
    if w_rowsum is not None and (w_rowsum.dtype != torch.int32 or w_rowsum.numel() != N or not w_rowsum.is_contiguous() or w_rowsum.device != wc.device):
        raise RuntimeError(f"w_rowsum must be a contiguous int32 tensor with {N} entries on the codes' device")
    lib, stream = _prepare(xc, wc, gc, xs, xo, ws_, wo)
    out = torch.empty((*xc.shape[:-1], N), dtype=torch.bfloat16, device=xc.device)
    nbytes = lib.ffq_linear_w8a8_workspace_bytes(M, N, K)
    ws = _workspace(nbytes, xc.device)
    words = _extrema_words(xc.device, stream) if want_extrema else None
    pair = torch.empty(2, dtype=torch.bfloat16, device=xc.device) if want_extrema else None
    status = lib.ffq_linear_w8a8_gated(
        _ptr(xc), _ptr(wc), _ptr(w_rowsum), _ptr(xs), _ptr(xo), int(xs.numel() != 1), _ptr(ws_), _ptr(wo), int(ws_.numel() != 1),
        _ptr(gc), _ptr(out), M, N, K, _ptr(ws), nbytes, _ptr(words), _ptr(pair), stream,
    )
    if status == 6:  # FFQ_ERR_DTYPE: outside the persistent kernel's whole-line path
        return None
    lib.check(status)
    return (out, pair) if want_extrema else out


def bmm_w8a8(
    x_codes: torch.Tensor,
    w_codes: torch.Tensor,
    x_scale: torch.Tensor,
    x_offset: torch.Tensor | None,
    w_scale: torch.Tensor,
    w_offset: torch.Tensor | None,
    out_dtype: torch.dtype = torch.bfloat16,
    out_scale: torch.Tensor | None = None,
    out_offset: torch.Tensor | None = None,
    out_num_bits: float = 8.0,
    requant_from: torch.dtype | None = None,
) -> torch.Tensor:
    """``torch.bmm`` on int8 codes in ONE launch: `x_codes` [B, M, K], `w_codes` [B, N, K] (the right operand K-contiguous),
    one parameter pair per operand (per-tensor quantizers) -> [B, M, N]; per matrix pair exactly :func:`linear_w8a8`, the output
    quantizer optionally in the epilogue (reference _gen/fallback.py:699-798: dequantize, bmm, output quantizer)."""
    if _native_route(x_codes):
        return torch.ops.fastforward_amd.bmm_w8a8(x_codes, w_codes, x_scale, x_offset, w_scale, w_offset, out_dtype, out_scale, out_offset, float(out_num_bits), requant_from)
    return _bmm_w8a8(x_codes, w_codes, x_scale, x_offset, w_scale, w_offset, out_dtype, out_scale, out_offset, out_num_bits, requant_from)


def _bmm_w8a8(x_codes, w_codes, x_scale, x_offset, w_scale, w_offset, out_dtype, out_scale, out_offset, out_num_bits, requant_from):  # type: ignore[no-untyped-def]
    """Python implementation of the ``bmm_w8a8`` operator; arguments in schema order."""
    if x_codes.dtype != torch.int8 or w_codes.dtype != torch.int8 or x_codes.dim() != 3 or w_codes.dim() != 3:
        raise TypeError("bmm_w8a8 expects int8 codes of shape [B, M, K] and [B, N, K]")
    xc, wc = x_codes.detach().contiguous(), w_codes.detach().contiguous()
    B, M, K = xc.shape
    if wc.shape[0] != B or wc.shape[2] != K:
        raise RuntimeError(f"batch1 and batch2 shapes cannot be multiplied ({tuple(xc.shape)} and {tuple(wc.shape)}^T)")
    N = wc.shape[1]
    f32 = lambda t: None if t is None else t.detach().reshape(-1).to(torch.float32).contiguous()  # noqa: E731
    xs, xo, ws_, wo, os_, oo = f32(x_scale), f32(x_offset), f32(w_scale), f32(w_offset), f32(out_scale), f32(out_offset)
    if xs.numel() != 1 or ws_.numel() != 1:
        raise RuntimeError("bmm_w8a8 takes per-tensor parameters (one scale per operand)")
    lib, stream = _prepare(xc, wc, xs, xo, ws_, wo, os_, oo)
    out = torch.empty((B, M, N), dtype=out_dtype, device=xc.device)
    nbytes = lib.ffq_bmm_w8a8_workspace_bytes(B, M, N, K)
    ws = _workspace(nbytes, xc.device)
    y_dt = _tag(requant_from or torch.bfloat16) if os_ is not None else 0
    lib.check(
        lib.ffq_bmm_w8a8(
            _ptr(xc), _ptr(wc), _ptr(xs), _ptr(xo), _ptr(ws_), _ptr(wo), _ptr(out), _tag(out_dtype), _ptr(os_), _ptr(oo),
            float(out_num_bits), y_dt, B, M, N, K, _ptr(ws), nbytes, stream,
        )
    )
    return out


def linear_wq(
    x: torch.Tensor,
    w_codes: torch.Tensor,
    w_scale: torch.Tensor,
    w_offset: torch.Tensor | None,
    group: int | None = None,
    bias: torch.Tensor | None = None,
    out_dtype: torch.dtype | None = None,
    pack_block: int = 0,
    two_pass: bool | None = None,
    split: int = 0,
) -> torch.Tensor | None:
    """A6, weight-only — ``F.linear(x, dequantize(w_codes))`` with the dequantization inside the GEMM's operand path
    (reference _gen/fallback.py:86-112: quantized weight, plain input).

    `x` is [..., K] bf16. `w_codes` is [N, K] int8 (codes of any bit-width <= 8) or, with ``pack_block`` > 0, the uint8
    output of :func:`pack_int4` / :func:`quantize_pack_int4` for an [N, K] weight packed with that block (two 4-bit codes per
    byte, [N * K / 2] or [N, K / 2]). `w_scale` / `w_offset` fp32 with 1 entry (per-tensor), N entries (per output channel)
    or N * K / group entries ([N, K / group] row-major: groups of `group` input channels, PerBlock(1, group, 0)).
    The weight the matrix cores see is bit for bit A2's bf16 result. ``two_pass``: None = the library's rule (from 4096
    tokens on A2 runs once per call into a scratch tensor and the GEMM streams that image), False = always convert inside
    the GEMM, True = offer the scratch tensor regardless of M (the library still decides). ``split``: 0 = the library's plan
    for cutting the K range of every output tile into slices when the launch has fewer tiles than the chip has CUs
    (``ffq_linear_wq_split``), >= 1 forces that many slices (tests, tuning).
    Returns None when the kernel does not cover the problem (dtypes, K % 64, group % 64): the caller dequantizes and runs a
    float GEMM as the reference does."""
    packed = pack_block > 0
    if packed:
        K = x.shape[-1]
        if w_codes.dtype != torch.uint8 or K == 0 or (w_codes.numel() * 2) % K:
            raise RuntimeError("packed weights are the uint8 output of pack_int4 for an [N, K] weight")
        N = w_codes.numel() * 2 // K
    else:
        if w_codes.dim() != 2:
            raise RuntimeError("linear_wq expects a [N, K] weight")
        N, K = w_codes.shape
    if x.shape[-1] != K:
        raise RuntimeError(f"mat1 and mat2 shapes cannot be multiplied ({x.numel() // max(x.shape[-1], 1)}x{x.shape[-1]} and {N}x{K}^T)")
    group = K if group is None else int(group)
    out_dtype = out_dtype or x.dtype
    if x.dtype not in _TAGS or w_codes.dtype not in _TAGS or out_dtype not in _TAGS:
        return None
    M = x.numel() // K if K else 0
    lib = _native.library()
    if not lib.ffq_linear_wq_supported(_tag(x.dtype), _tag(w_codes.dtype), _tag(out_dtype), M, N, K, group, int(pack_block)):
        return None
    if _native_route(x):  # dispatcher -> C++ (csrc/ffq_torch.cpp) -> C ABI
        return torch.ops.fastforward_amd.linear_wq(x, w_codes, w_scale, w_offset, group, bias, out_dtype, int(pack_block), -1 if two_pass is None else int(bool(two_pass)), int(split))
    return _linear_wq(x, w_codes, w_scale, w_offset, group, bias, out_dtype, int(pack_block), -1 if two_pass is None else int(bool(two_pass)), int(split))


def _linear_wq(x, w_codes, w_scale, w_offset, group, bias, out_dtype, pack_block, two_pass, split):  # type: ignore[no-untyped-def]
    """Python implementation of the ``linear_wq`` operator for a problem the kernel covers; arguments in schema order
    (`two_pass`: -1 = the library's rule, 0 = never, 1 = offer the image's scratch whatever M)."""
    K = x.shape[-1]
    N = w_codes.numel() * 2 // K if pack_block > 0 else w_codes.shape[0]
    M = x.numel() // K
    two_pass = None if two_pass < 0 else bool(two_pass)
    xc, wc = x.detach().contiguous(), w_codes.detach().contiguous()
    sc = w_scale.detach().reshape(-1).to(torch.float32).contiguous()
    of = None if w_offset is None else w_offset.detach().reshape(-1).to(torch.float32).contiguous()
    if of is not None and of.numel() != sc.numel():
        raise RuntimeError(f"scale has {sc.numel()} entries, offset {of.numel()}")
    bias_c = None if bias is None else bias.detach().contiguous()
    lib, stream = _prepare(xc, wc, sc, of, bias_c)
    out = torch.empty((*xc.shape[:-1], N), dtype=out_dtype, device=xc.device)
    nbytes, tickets = _wq_scratch(lib, M, N, K, False, two_pass, split, xc.device, stream)
    ws = _workspace(nbytes, xc.device)
    lib.check(
        lib.ffq_linear_wq(
            _ptr(xc), _tag(xc.dtype), _ptr(wc), _tag(wc.dtype), int(pack_block), _ptr(sc), _ptr(of), sc.numel(), group,
            _ptr(bias_c), _tag(bias_c.dtype) if bias_c is not None else 0, _ptr(out), _tag(out_dtype), M, N, K, _ptr(ws), nbytes,
            _ptr(tickets), int(split), stream,
        )
    )
    return out


def linear_wq_multi(
    x: torch.Tensor,
    w_codes: Sequence[torch.Tensor],
    w_scales: Sequence[torch.Tensor],
    w_offsets: Sequence[torch.Tensor | None],
    group: int | None = None,
    out_dtype: torch.dtype | None = None,
    pack_block: int = 0,
    two_pass: bool | None = None,
    split: int = 0,
) -> list[torch.Tensor] | None:
    """Two or three weight-only linears on the SAME input in one launch (q_proj / k_proj / v_proj: three ``QuantizedLinear``
    modules reading one hidden state, reference nn/linear.py:32-39): ``[linear_wq(x, w_i, ...) for i]`` as separate tensors, from
    one tile walk over all the column tiles. Operands as :func:`linear_wq`; all weights share dtype, packing, `group`, the
    granularity kind and the presence of offsets, every weight but the last has a multiple of 256 rows. None when that does
    not hold (the caller runs the linears one by one)."""
    count = len(w_codes)
    if not (2 <= count <= 3) or len(w_scales) != count or len(w_offsets) != count:
        return None
    K = x.shape[-1]
    packed = pack_block > 0
    rows = []
    for c in w_codes:
        if packed:
            if c.dtype != torch.uint8 or K == 0 or (c.numel() * 2) % K:
                return None
            rows.append(c.numel() * 2 // K)
        else:
            if c.dim() != 2 or c.shape[1] != K:
                return None
            rows.append(c.shape[0])
    group = K if group is None else int(group)
    out_dtype = out_dtype or x.dtype
    if x.dtype not in _TAGS or any(c.dtype != w_codes[0].dtype for c in w_codes) or w_codes[0].dtype not in _TAGS or out_dtype not in _TAGS:
        return None
    if any(n % 256 for n in rows[:-1]) or any((o is None) != (w_offsets[0] is None) for o in w_offsets):
        return None
    M = x.numel() // K if K else 0
    N = sum(rows)
    lib = _native.library()
    if not lib.ffq_linear_wq_supported(_tag(x.dtype), _tag(w_codes[0].dtype), _tag(out_dtype), M, N, K, group, int(pack_block)):
        return None
    flat = lambda t: None if t is None else t.detach().reshape(-1).to(torch.float32).contiguous()  # noqa: E731
    scales, offsets = [flat(t) for t in w_scales], [flat(t) for t in w_offsets]
    kinds = {int(s_.numel() != 1) for s_ in scales}
    if len(kinds) != 1:
        return None
    per_row = kinds.pop()
    for n, s_, o_ in zip(rows, scales, offsets):
        if s_.numel() != (n * (K // group) if per_row else 1) or (o_ is not None and o_.numel() != s_.numel()):
            return None
    if not per_row and group != K:
        return None
    xc = x.detach().contiguous()
    codes = [c.detach().contiguous() for c in w_codes]
    lib, stream = _prepare(xc, *codes, *scales, *[o for o in offsets if o is not None])
    outs = [torch.empty((*xc.shape[:-1], n), dtype=out_dtype, device=xc.device) for n in rows]
    nbytes, tickets = _wq_scratch(lib, M, N, K, False, two_pass, split, xc.device, stream)
    ws = _workspace(nbytes, xc.device)
    ptrs = lambda tensors: (ctypes.c_void_p * count)(*[_ptr(t) for t in tensors])  # noqa: E731
    lib.check(
        lib.ffq_linear_wq_multi(
            _ptr(xc), _tag(xc.dtype), count, ptrs(codes), _tag(codes[0].dtype), int(pack_block), ptrs(scales), ptrs(offsets), per_row, group,
            ptrs(outs), _tag(out_dtype), M, (ctypes.c_int64 * count)(*rows), K, _ptr(ws), nbytes, _ptr(tickets), int(split), stream,
        )
    )
    return outs


def _wq_scratch(lib: Any, M: int, N: int, K: int, mlp: bool, two_pass: bool | None, split: int, device: torch.device, stream: int) -> tuple[int, torch.Tensor | None]:
    """(workspace bytes, ticket buffer) of a weight-only GEMM launch: the split-K slabs of the plan (or of a forced `split`) at the
    front, the bf16 image(s) of the two-pass form behind them."""
    tickets = int(lib.ffq_linear_wq_tickets(M, N, K, int(mlp)))  # two per tile of the last round
    plan = int(lib.ffq_linear_wq_split(M, N, K, int(mlp)))
    use = max(1, int(split) if split > 0 else plan)
    slabs = int(lib.ffq_linear_wq_slab_bytes(M, N, K, int(mlp), use))
    if two_pass is False:
        image = 0
    elif two_pass:
        image = (2 if mlp else 1) * N * K * 2
    else:  # the library's rule: its figure minus the slabs of its own plan
        full = int(lib.ffq_mlp_gate_up_wq_workspace_bytes(M, N, K) if mlp else lib.ffq_linear_wq_workspace_bytes(M, N, K))
        image = full - int(lib.ffq_linear_wq_slab_bytes(M, N, K, int(mlp), plan))
    # (tickets whenever slabs are offered: where the preferred form declines the weight's storage, the form that takes over has a plan of its own)
    return slabs + image, (_tickets(tickets, device, stream) if (use > 1 or slabs > 0) and tickets > 0 else None)


def mlp_gate_up_wq(
    x: torch.Tensor,
    gate_codes: torch.Tensor,
    up_codes: torch.Tensor,
    gate_scale: torch.Tensor,
    gate_offset: torch.Tensor | None,
    up_scale: torch.Tensor,
    up_offset: torch.Tensor | None,
    group: int | None = None,
    pack_block: int = 0,
    two_pass: bool | None = None,
    split: int = 0,
) -> torch.Tensor | None:
    """``silu(gate_proj(x)) * up_proj(x)`` of a weight-only quantized MLP (reference quantized_llama/mlp.py:30-40 over
    _gen/fallback.py:86-112) in one launch: bit for bit ``silu_mul_quantize(linear_wq(x, gate), linear_wq(x, up), want_product=True)``
    without the two bf16 projections in HBM. Operands as :func:`linear_wq` (both weights in the same form); bf16 only.
    None when the kernel does not cover the problem."""
    K = x.shape[-1]
    if pack_block > 0:
        if gate_codes.dtype != torch.uint8 or K == 0 or (gate_codes.numel() * 2) % K:
            raise RuntimeError("packed weights are the uint8 output of pack_int4 for an [N, K] weight")
        N = gate_codes.numel() * 2 // K
    else:
        if gate_codes.dim() != 2 or gate_codes.shape[1] != K:
            raise RuntimeError("mlp_gate_up_wq expects [N, K] weights")
        N = gate_codes.shape[0]
    if gate_codes.shape != up_codes.shape or gate_codes.dtype != up_codes.dtype:
        raise RuntimeError("gate and up weights differ in shape or dtype")
    group = K if group is None else int(group)
    if x.dtype != torch.bfloat16 or gate_codes.dtype not in _TAGS or N % 128 or (gate_offset is None) != (up_offset is None):
        return None
    M = x.numel() // K if K else 0
    lib = _native.library()
    if not lib.ffq_linear_wq_supported(_tag(x.dtype), _tag(gate_codes.dtype), _tag(torch.bfloat16), M, N, K, group, int(pack_block)):
        return None
    xc, gc, uc = x.detach().contiguous(), gate_codes.detach().contiguous(), up_codes.detach().contiguous()
    flat = lambda t: None if t is None else t.detach().reshape(-1).to(torch.float32).contiguous()  # noqa: E731
    gs, go, us, uo = flat(gate_scale), flat(gate_offset), flat(up_scale), flat(up_offset)
    if gs.numel() != us.numel() or (go is not None and (go.numel() != gs.numel() or uo.numel() != gs.numel())):
        raise RuntimeError("gate and up parameters differ in count")
    lib, stream = _prepare(xc, gc, uc, gs, go, us, uo)
    out = torch.empty((*xc.shape[:-1], N), dtype=torch.bfloat16, device=xc.device)
    nbytes, tickets = _wq_scratch(lib, M, N, K, True, two_pass, split, xc.device, stream)
    ws = _workspace(nbytes, xc.device)
    lib.check(
        lib.ffq_mlp_gate_up_wq(
            _ptr(xc), _tag(xc.dtype), _ptr(gc), _ptr(uc), _tag(gc.dtype), int(pack_block), _ptr(gs), _ptr(go), _ptr(us), _ptr(uo),
            gs.numel(), group, _ptr(out), M, N, K, _ptr(ws), nbytes, _ptr(tickets), int(split), stream,
        )
    )
    return out


def mlp_gate_up_w8a8(
    x_codes: torch.Tensor,
    gate_codes: torch.Tensor,
    up_codes: torch.Tensor,
    x_scale: torch.Tensor,
    x_offset: torch.Tensor | None,
    gate_scale: torch.Tensor,
    up_scale: torch.Tensor,
    out_scale: torch.Tensor,
    out_offset: torch.Tensor | None,
    out_num_bits: float = 8.0,
    gate_rowsum: torch.Tensor | None = None,
    up_rowsum: torch.Tensor | None = None,
) -> torch.Tensor | None:
    """gate_proj + up_proj + ``silu(gate) * up`` + the down_proj input quantizer in ONE launch (reference
    quantized_llama/mlp.py:30-40): int8 codes of the product, equal to
    ``silu_mul_quantize(linear_w8a8(x, gate), linear_w8a8(x, up))`` exactly. Per-tensor activation parameters,
    per-output-channel symmetric weights. Returns None when the shapes are outside the kernel's range."""
    xc, gc, uc = x_codes.detach().contiguous(), gate_codes.detach().contiguous(), up_codes.detach().contiguous()
    if not (xc.dtype == gc.dtype == uc.dtype == torch.int8) or gc.shape != uc.shape or gc.dim() != 2:
        raise TypeError("mlp_gate_up_w8a8 expects int8 codes and equally shaped gate / up weights")
    K, N = xc.shape[-1], gc.shape[0]
    M = xc.numel() // K if K else 0
    if gc.shape[1] != K:
        raise RuntimeError(f"mat1 and mat2 shapes cannot be multiplied ({M}x{K} and {tuple(gc.shape)}^T)")
    if N % 128 or K % 128 or K < 256:
        return None

    def f32(t: torch.Tensor | None, n: int) -> torch.Tensor | None:
        if t is None:
            return None
        t = t.detach().reshape(-1).to(torch.float32).contiguous()
        if t.numel() != n:
            raise RuntimeError(f"expected {n} parameter entries, got {t.numel()}")
        return t

    xs, xo, gs, us, os_, oo = f32(x_scale, 1), f32(x_offset, 1), f32(gate_scale, N), f32(up_scale, N), f32(out_scale, 1), f32(out_offset, 1)
    lib, stream = _prepare(xc, gc, uc, xs, xo, gs, us, os_, oo)
    out = torch.empty((*xc.shape[:-1], N), dtype=torch.int8, device=xc.device)
    nbytes = lib.ffq_mlp_gate_up_w8a8_workspace_bytes(M, N, K)
    ws = _workspace(nbytes, xc.device)
    for rs in (gate_rowsum, up_rowsum):
        if rs is not None and (rs.dtype != torch.int32 or rs.numel() != N or not rs.is_contiguous() or rs.device != gc.device):
            raise RuntimeError(f"row sums must be contiguous int32 tensors with {N} entries on the codes' device")
    status = lib.ffq_mlp_gate_up_w8a8(
        _ptr(xc), _ptr(gc), _ptr(uc), _ptr(gate_rowsum), _ptr(up_rowsum), _ptr(xs), _ptr(xo), _ptr(gs), _ptr(us), _ptr(out), _ptr(os_), _ptr(oo),
        float(out_num_bits), M, N, K, _ptr(ws), nbytes, stream,
    )
    if status == 6:
        return None
    lib.check(status)
    return out


def mlp_gate_up_w8a8_estimating(
    x_codes_gate: torch.Tensor,
    x_codes_up: torch.Tensor,
    gate_codes: torch.Tensor,
    up_codes: torch.Tensor,
    x_params_gate: tuple[torch.Tensor, torch.Tensor | None],
    x_params_up: tuple[torch.Tensor, torch.Tensor | None],
    gate_params: tuple[torch.Tensor, torch.Tensor | None],
    up_params: tuple[torch.Tensor, torch.Tensor | None],
    want_extrema: bool = False,
) -> torch.Tensor | tuple[torch.Tensor, torch.Tensor] | None:
    """``silu(linear(xg, Wg)) * linear(xu, Wu)`` as a bf16 tensor for int8 operands whose quantizers are still being calibrated
    (C ABI ``ffq_mlp_gate_up_w8a8_estimating``): `x_codes_gate` / `x_codes_up` are the codes gate_proj's and up_proj's own input
    quantizers produced from the same activation, each with its (scale, offset). Whether the two hold equal parameters — then the
    one-launch gate + up + SiLU * up kernel runs on one of the code tensors — is decided on the device; otherwise the two linears
    run, the second with the gated epilogue. Same values either way: ``silu_mul_quantize(linear_w8a8(xg, ...), linear_w8a8(xu, ...),
    (), want_product=True)[0]``. ``want_extrema``: also ``[min, max]`` of the product. None outside the kernels' shapes."""
    xg, xu = x_codes_gate.detach().contiguous(), x_codes_up.detach().contiguous()
    gc, uc = gate_codes.detach().contiguous(), up_codes.detach().contiguous()
    if not (xg.dtype == xu.dtype == gc.dtype == uc.dtype == torch.int8) or gc.shape != uc.shape or gc.dim() != 2 or xg.shape != xu.shape:
        raise TypeError("mlp_gate_up_w8a8_estimating expects int8 codes, equally shaped gate / up weights and equally shaped activations")
    K, N = xg.shape[-1], gc.shape[0]
    M = xg.numel() // K if K else 0
    if gc.shape[1] != K:
        raise RuntimeError(f"mat1 and mat2 shapes cannot be multiplied ({M}x{K} and {tuple(gc.shape)}^T)")
    if N % 128 or K % 128 or K < 256 or M < 128 or ((M + 255) // 256) * ((N + 255) // 256) < 64:
        return None

    def f32(t: torch.Tensor | None, n: int) -> torch.Tensor | None:
        if t is None:
            return None
        t = t.detach().reshape(-1).to(torch.float32).contiguous()
        return t if t.numel() == n else None

    xsg, xsu, gs, us = f32(x_params_gate[0], 1), f32(x_params_up[0], 1), f32(gate_params[0], N), f32(up_params[0], N)
    if xsg is None or xsu is None or gs is None or us is None:
        return None
    xog, xou, go, uo = f32(x_params_gate[1], 1), f32(x_params_up[1], 1), f32(gate_params[1], N), f32(up_params[1], N)
    if any(p[1] is not None and o is None for p, o in ((x_params_gate, xog), (x_params_up, xou), (gate_params, go), (up_params, uo))):
        return None  # an offset of another granularity
    lib, stream = _prepare(xg, xu, gc, uc, xsg, xsu, gs, us, xog, xou, go, uo)
    product = torch.empty((*xg.shape[:-1], N), dtype=torch.bfloat16, device=xg.device)
    gate_scratch = torch.empty_like(product)  # touched only where the two-launch route runs
    nbytes = lib.ffq_mlp_gate_up_w8a8_estimating_workspace_bytes(M, N, K)
    ws = _workspace(nbytes, xg.device)
    words = _extrema_words(xg.device, stream) if want_extrema else None
    pair = torch.empty(2, dtype=torch.bfloat16, device=xg.device) if want_extrema else None
    status = lib.ffq_mlp_gate_up_w8a8_estimating(
        _ptr(xg), _ptr(xu), _ptr(gc), _ptr(uc), _ptr(xsg), _ptr(xog), _ptr(xsu), _ptr(xou), _ptr(gs), _ptr(go), _ptr(us), _ptr(uo),
        _ptr(gate_scratch), _ptr(product), M, N, K, _ptr(ws), nbytes, _ptr(words), _ptr(pair), stream,
    )
    if status == 6:
        return None
    lib.check(status)
    return (product, pair) if want_extrema else product


def quantize_rows_rowsum(
    weight: torch.Tensor, scale: torch.Tensor, offset: torch.Tensor | None, num_bits: float = 8.0, rowsum_out: torch.Tensor | None = None
) -> tuple[torch.Tensor, torch.Tensor] | None:
    """A1 of a ``[rows, cols]`` weight with one (scale, offset) per row into int8 codes AND the int32 row sums of those
    codes (the zero-point term of the W8A8 linear), one pass. Codes equal ``quantize_by_tile(weight, scale, (1, cols), ...)``.
    ``rowsum_out``: a ZEROED contiguous int32 ``[rows]`` tensor to receive the sums (the kernel adds into it; a forward hands
    out slices of one buffer it zeroes once). Returns None where the one-pass kernel does not apply (not bf16,
    ``cols % 1024 != 0``, a fractional bit width): take ``quantize_by_tile``."""
    if weight.dim() != 2:
        raise RuntimeError("quantize_rows_rowsum expects a [rows, cols] weight")
    rows, cols = weight.shape
    if weight.dtype != torch.bfloat16 or cols % 1024 or not weight.is_contiguous() or float(num_bits) != int(num_bits):
        return None  # (a fractional bit width: the kernel clamps before it rounds, which needs integer bounds)
    sc = scale.detach().reshape(-1).to(torch.float32).contiguous()
    of = None if offset is None else offset.detach().reshape(-1).to(torch.float32).contiguous()
    if sc.numel() != rows or (of is not None and of.numel() != rows):
        raise RuntimeError(f"expected one scale (and offset) per row ({rows}), got {sc.numel()}")
    wd = weight.detach()
    lib, stream = _prepare(wd, sc, of)
    codes = torch.empty((rows, cols), dtype=torch.int8, device=wd.device)
    if rowsum_out is None:
        rowsum = torch.zeros((rows,), dtype=torch.int32, device=wd.device)
    else:
        rowsum = rowsum_out
        if rowsum.dtype != torch.int32 or rowsum.shape != (rows,) or not rowsum.is_contiguous() or rowsum.device != wd.device:
            raise RuntimeError(f"rowsum_out must be a zeroed contiguous int32 tensor with {rows} entries on the weight's device")
    lib.check(lib.ffq_quantize_rows_rowsum(_ptr(wd), _tag(wd.dtype), _ptr(sc), _ptr(of), rows, cols, float(num_bits), _ptr(codes), _ptr(rowsum), stream))
    return codes, rowsum


def quantize_rows_batch(
    weights: Sequence[torch.Tensor], scales: Sequence[torch.Tensor], offsets: Sequence[torch.Tensor | None], num_bits: float = 8.0,
    rowsums: Sequence[torch.Tensor] | None = None,
) -> list[torch.Tensor] | None:
    """A1 of up to 8 ``[rows, cols]`` bf16 weights with one (scale, offset) per row into int8 codes, ONE launch; each result
    equals ``quantize_by_tile(weight, scale, (1, cols), num_bits, torch.int8, offset)``. The seven linears of a decoder layer
    are re-quantized on every forward (reference nn/linear.py:34); as seven launches the short ones (k_proj / v_proj) run far
    below the streaming rate. `rowsums` (one ZEROED contiguous int32 [rows] tensor per weight, cols % 1024 == 0): the launch also
    adds each row's code sum into it — what :func:`linear_w8a8` takes as ``w_rowsum``. Returns None where the one-launch
    kernel does not apply (then quantize member by member)."""
    if not weights or len(weights) > FFQ_MAX_BATCH or not (len(weights) == len(scales) == len(offsets)) or float(num_bits) != int(num_bits):
        return None
    if rowsums is not None and (len(rowsums) != len(weights) or any(
            r.dtype != torch.int32 or r.numel() != w.shape[0] or not r.is_contiguous() or w.shape[1] % 1024 for r, w in zip(rowsums, weights))):
        return None
    sc, of = [], []
    for w, s, o in zip(weights, scales, offsets):
        if w.dim() != 2 or w.dtype != torch.bfloat16 or not w.is_contiguous() or w.shape[1] % 16 or (w.numel() // 16) % 256 or w.data_ptr() % 16:
            return None  # (a 16-byte-misaligned view: the member-by-member kernels take it)
        s32 = s.detach().reshape(-1).to(torch.float32).contiguous()
        o32 = None if o is None else o.detach().reshape(-1).to(torch.float32).contiguous()
        if s32.numel() != w.shape[0] or (o32 is not None and o32.numel() != w.shape[0]):
            return None
        sc.append(s32)
        of.append(o32)
    lib, stream = _prepare(*[w.detach() for w in weights], *sc, *[o for o in of if o is not None])
    codes = [torch.empty(w.shape, dtype=torch.int8, device=w.device) for w in weights]
    batch = RowsBatch()
    batch.count, batch.num_bits = len(weights), float(num_bits)
    for i, (w, s, o, c) in enumerate(zip(weights, sc, of, codes)):
        batch.data[i], batch.scale[i], batch.offset[i], batch.codes[i] = _ptr(w.detach()), _ptr(s), _ptr(o), _ptr(c)
        batch.rows[i], batch.cols[i] = w.shape
        batch.rowsum[i] = None if rowsums is None else _ptr(rowsums[i])
    status = lib.ffq_quantize_rows_batch(ctypes.byref(batch), _tag(torch.bfloat16), stream)
    if status == 6:
        return None
    lib.check(status)
    return codes


def _fan(quantizers: Sequence[tuple[torch.Tensor, torch.Tensor | None]], num_bits: float, shape: Sequence[int], device: torch.device):
    """(FanOut struct, code tensors, tensors kept alive) for the static per-tensor quantizers of a fused producer."""
    scales, offsets, keep = [], [], []
    for scale, offset in quantizers:
        s = scale.detach().reshape(-1).to(torch.float32)
        o = None if offset is None else offset.detach().reshape(-1).to(torch.float32)
        if s.numel() != 1 or (o is not None and o.numel() != 1):
            raise RuntimeError("fused producers take per-tensor quantizers (one scale, one offset)")
        scales.append(s)
        offsets.append(o)
        keep += [s, o]
    codes = [torch.empty(tuple(shape), dtype=torch.int8, device=device) for _ in quantizers]
    fan = FanOut.make(num_bits, [_ptr(s) for s in scales], [_ptr(o) for o in offsets], [_ptr(c) for c in codes])
    return fan, codes, keep


def add_rmsnorm_quantize(
    x: torch.Tensor,
    delta: torch.Tensor | None,
    weight: torch.Tensor,
    eps: float,
    quantizers: Sequence[tuple[torch.Tensor, torch.Tensor | None]] = (),
    num_bits: float = 8.0,
    want_sum: bool = True,
    want_norm: bool = False,
    sum_inplace: bool = False,
) -> tuple[torch.Tensor | None, torch.Tensor | None, list[torch.Tensor]]:
    """Residual add + RMSNorm + A1 for up to three per-tensor int8 quantizers, one pass
    (reference docs/examples/doc_helpers/quantized_llama/rms_norm.py:17-35 behind decoder.py:60-90).

    Returns ``(x + delta, normalised or None, [codes per quantizer])``; with ``delta is None`` the first
    element is `x` itself. ``sum_inplace`` writes the sum over `x` (the residual stream of a decoder).
    """
    xc = x.detach().contiguous()
    dc = None if delta is None else delta.detach().contiguous()
    wc = weight.detach().contiguous()
    if dc is not None and dc.shape != xc.shape:
        raise RuntimeError(f"residual shapes differ: {tuple(xc.shape)} vs {tuple(dc.shape)}")
    if wc.dim() != 1 or wc.shape[0] != xc.shape[-1] or wc.dtype != xc.dtype or (dc is not None and dc.dtype != xc.dtype):
        raise RuntimeError("RMSNorm weight must be [hidden] in the activation dtype")
    lib, stream = _prepare(xc, dc, wc, *[t for q in quantizers for t in q])
    cols = xc.shape[-1]
    rows = xc.numel() // cols if cols else 0
    if sum_inplace and xc.data_ptr() != x.data_ptr():
        raise RuntimeError("sum_inplace needs a contiguous residual tensor")
    total = xc if dc is None or sum_inplace else (torch.empty_like(xc) if want_sum else None)
    norm = torch.empty_like(xc) if want_norm else None
    fan, codes, keep = _fan(quantizers, num_bits, xc.shape, xc.device)
    lib.check(
        lib.ffq_add_rmsnorm_quantize(
            _ptr(xc), _ptr(dc), None if dc is None else _ptr(total), _ptr(wc), _tag(xc.dtype), rows, cols, float(eps),
            _ptr(norm), ctypes.byref(fan), stream,
        )
    )
    del keep
    if sum_inplace and dc is not None:
        torch.autograd.graph.increment_version(x)  # written through a raw pointer
    return total, norm, codes


def silu_mul_quantize(
    gate: torch.Tensor,
    up: torch.Tensor,
    quantizers: Sequence[tuple[torch.Tensor, torch.Tensor | None]] = (),
    num_bits: float = 8.0,
    want_product: bool = False,
) -> tuple[torch.Tensor | None, list[torch.Tensor]]:
    """``silu(gate) * up`` + A1, one pass (reference quantized_llama/mlp.py:30-40)."""
    gc, uc = gate.detach().contiguous(), up.detach().contiguous()
    if gc.shape != uc.shape or gc.dtype != uc.dtype:
        raise RuntimeError(f"gate and up differ: {tuple(gc.shape)} {gc.dtype} vs {tuple(uc.shape)} {uc.dtype}")
    lib, stream = _prepare(gc, uc, *[t for q in quantizers for t in q])
    product = torch.empty_like(gc) if want_product else None
    fan, codes, keep = _fan(quantizers, num_bits, gc.shape, gc.device)
    lib.check(lib.ffq_silu_mul_quantize(_ptr(gc), _ptr(uc), _tag(gc.dtype), gc.numel(), _ptr(product), ctypes.byref(fan), stream))
    del keep
    return product, codes


def rope_(q: torch.Tensor | None, k: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, head_dim: int) -> None:
    """Rotary embedding IN PLACE on the q/k projections laid out ``[batch, seq, heads * head_dim]``
    (reference quantized_llama/attention.py:20-41); `cos`/`sin` are ``[seq, head_dim]``. ``q=None``: k alone (q is rotated
    inside :func:`attention` when that call gets the tables: ``q_rope=(cos, sin)``)."""
    if not ((q is None or q.is_contiguous()) and k.is_contiguous() and cos.is_contiguous() and sin.is_contiguous()):
        raise RuntimeError("rope_ works in place on contiguous projections")
    if (q is not None and (q.dim() != 3 or q.shape[:2] != k.shape[:2])) or k.dim() != 3 or cos.shape != (k.shape[1], head_dim) or sin.shape != cos.shape:
        raise RuntimeError("rope_ expects q/k [batch, seq, heads * head_dim] and cos/sin [seq, head_dim]")
    if not ((q is None or q.dtype == k.dtype) and k.dtype == cos.dtype == sin.dtype):
        raise RuntimeError("rope_ expects one dtype")
    lib, stream = _prepare(q, k, cos, sin)
    tokens = k.shape[0] * k.shape[1]
    lib.check(
        lib.ffq_rope_inplace(
            _ptr(q), 0 if q is None else q.shape[2] // head_dim, _ptr(k), k.shape[2] // head_dim, _tag(k.dtype), tokens, k.shape[1], head_dim,
            _ptr(cos), _ptr(sin), stream,
        )
    )
    # written through raw pointers: tell the version counters (whatever is keyed on them, e.g. the activation-code memo, must miss)
    if q is not None:
        torch.autograd.graph.increment_version(q)
    torch.autograd.graph.increment_version(k)


def attention(
    q: torch.Tensor,
    k: torch.Tensor,
    v: torch.Tensor,
    head_dim: int,
    causal: bool = True,
    quantizer: tuple[torch.Tensor, torch.Tensor | None] | None = None,
    num_bits: float = 8.0,
    want_context: bool = True,
    softmax_scale: float | None = None,
    q_rope: tuple[torch.Tensor, torch.Tensor] | None = None,
) -> tuple[torch.Tensor | None, torch.Tensor | None]:
    """Attention on the projections as they leave q/k/v_proj — ``q`` ``[batch, seq, heads * head_dim]``, ``k`` / ``v``
    ``[batch, seq, kv_heads * head_dim]``, rotary embedding already applied — as one flash-style launch (reference
    quantized_llama/attention.py:45-92), optionally with the static per-tensor input quantizer of ``o_proj`` (A1)
    fused in. ``q_rope=(cos, sin)`` (``[seq, head_dim]`` each): `q` is given UN-rotated and rotated as the kernel loads it —
    equal to ``rope_`` on q followed by this call, bit for bit, without the pass over q (k is rotated by the caller:
    ``rope_(None, k, ...)``). Returns ``(context [batch, seq, heads * head_dim] or None, int8 codes or None)``."""
    if not (q.is_contiguous() and k.is_contiguous() and v.is_contiguous()):
        raise RuntimeError("attention expects contiguous projections")
    if q.dim() != 3 or k.dim() != 3 or k.shape != v.shape or q.shape[:2] != k.shape[:2] or q.shape[2] % head_dim or k.shape[2] % head_dim:
        raise RuntimeError("attention expects q [batch, seq, heads * head_dim] and k / v [batch, seq, kv_heads * head_dim]")
    if not (q.dtype == k.dtype == v.dtype):
        raise RuntimeError("attention expects one dtype")
    if quantizer is None and not want_context:
        raise RuntimeError("attention: nothing to compute (no context, no codes)")
    scale = offset = codes = None
    if quantizer is not None:
        scale = quantizer[0].detach().reshape(-1).to(torch.float32)
        offset = None if quantizer[1] is None else quantizer[1].detach().reshape(-1).to(torch.float32)
        if scale.numel() != 1 or (offset is not None and offset.numel() != 1):
            raise RuntimeError("the fused quantizer is per-tensor (one scale, one offset)")
    cos = sin = None
    if q_rope is not None:
        cos, sin = q_rope
        if cos.shape != (q.shape[1], head_dim) or sin.shape != cos.shape or not (cos.dtype == sin.dtype == q.dtype) or not (cos.is_contiguous() and sin.is_contiguous()):
            raise RuntimeError("attention: q_rope is (cos, sin), contiguous [seq, head_dim] tables in the activations' dtype")
    lib, stream = _prepare(q, k, v, scale, offset, cos, sin)
    if quantizer is not None:
        codes = torch.empty(q.shape, dtype=torch.int8, device=q.device)
    ctx = torch.empty_like(q) if want_context else None
    b, s, _ = q.shape
    lib.check(
        lib.ffq_attention(
            _ptr(q), _ptr(k), _ptr(v), _tag(q.dtype), b, s, q.shape[2] // head_dim, k.shape[2] // head_dim, head_dim,
            float(head_dim**-0.5 if softmax_scale is None else softmax_scale), int(bool(causal)),
            _ptr(ctx), _ptr(codes), _ptr(scale), _ptr(offset), float(num_bits), _ptr(cos), _ptr(sin), stream,
        )
    )
    return ctx, codes


# ---------------------------------------------------------------------------------------------
# torch custom-op registration: same four schemas as the reference's `fastforward::` ops, in this
# package's own namespace (defining `fastforward::*` twice in one process is an error).
# ---------------------------------------------------------------------------------------------
_LIBRARY = torch.library.Library("fastforward_amd", "DEF")
_LIBRARY.define(
    "quantize_by_tile(Tensor data, Tensor scale, SymInt[] tile_size, float num_bits, "
    "ScalarType? output_dtype, Tensor? offset=None) -> Tensor"
)
_LIBRARY.define(
    "dequantize_by_tile(Tensor data, Tensor scale, SymInt[] tile_size, Tensor? offset=None, "
    "ScalarType? output_dtype=None) -> Tensor"
)
_LIBRARY.define(
    "quantize_dynamic_by_tile(Tensor data, SymInt[] tile_size, float num_bits, bool symmetric, "
    "bool allow_one_sided, ScalarType? output_dtype) -> (Tensor, Tensor, Tensor)"
)
_LIBRARY.define(
    "quantize_by_tile_backward(Tensor data, Tensor output_grad, Tensor scale, SymInt[] tile_size, "
    "float num_bits, Tensor? offset=None) -> Tensor[]"
)
# ... and the hot entry points behind the dispatcher and the range estimator as operators of the same library (round 5), so that
# they too have C++ device kernels (csrc/ffq_torch.cpp); the Python bodies stay registered and serve whenever the extension is absent
_LIBRARY.define(
    "running_minmax_step(Tensor data, SymInt[] tile_size, Tensor(a!) running_min, Tensor(b!) running_max, Tensor(c!)? status_flags, "
    "float num_bits, bool symmetric, bool allow_one_sided, Tensor(d!) scale_out, Tensor(e!)? offset_out) -> ()"
)
_LIBRARY.define(
    "linear_w8a8(Tensor x_codes, Tensor w_codes, Tensor x_scale, Tensor? x_offset, Tensor w_scale, Tensor? w_offset, Tensor? bias, "
    "ScalarType out_dtype, Tensor? out_scale, Tensor? out_offset, float out_num_bits, Tensor? w_rowsum, ScalarType? requant_from) -> Tensor"
)
_LIBRARY.define(
    "bmm_w8a8(Tensor x_codes, Tensor w_codes, Tensor x_scale, Tensor? x_offset, Tensor w_scale, Tensor? w_offset, ScalarType out_dtype, "
    "Tensor? out_scale, Tensor? out_offset, float out_num_bits, ScalarType? requant_from) -> Tensor"
)
_LIBRARY.define(
    "linear_wq(Tensor x, Tensor w_codes, Tensor w_scale, Tensor? w_offset, int group, Tensor? bias, ScalarType out_dtype, int pack_block, "
    "int two_pass, int split) -> Tensor"
)
_LIBRARY.impl("running_minmax_step", _running_minmax_step, "CompositeExplicitAutograd")
_LIBRARY.impl("linear_w8a8", _linear_w8a8, "CompositeExplicitAutograd")
_LIBRARY.impl("bmm_w8a8", _bmm_w8a8, "CompositeExplicitAutograd")
_LIBRARY.impl("linear_wq", _linear_wq, "CompositeExplicitAutograd")
_LIBRARY.impl("quantize_by_tile", quantize_by_tile, "CompositeExplicitAutograd")
_LIBRARY.impl("dequantize_by_tile", dequantize_by_tile, "CompositeExplicitAutograd")
_LIBRARY.impl("quantize_dynamic_by_tile", quantize_dynamic_by_tile, "CompositeExplicitAutograd")
_LIBRARY.impl("quantize_by_tile_backward", quantize_by_tile_backward, "CompositeExplicitAutograd")


# Fake / Meta implementations of all four ops (reference _quantizer_impl.py:288-339): shapes, dtypes and devices of the
# real outputs without touching data, so the ops trace under FakeTensor / torch.compile / torch.export.
def _float_result(*dtypes: torch.dtype) -> torch.dtype:
    out = dtypes[0]
    for d in dtypes[1:]:
        out = torch.promote_types(out, d)
    return out if out.is_floating_point else torch.float32


def _meta_quantize_by_tile(data, scale, tile_size, num_bits, output_dtype, offset=None):  # type: ignore[no-untyped-def]
    if output_dtype is None:
        output_dtype = _float_result(data.dtype, scale.dtype, (offset if offset is not None else scale).dtype)
    return torch.empty(data.shape, dtype=output_dtype, device=data.device)


def _meta_dequantize_by_tile(data, scale, tile_size, offset=None, output_dtype=None):  # type: ignore[no-untyped-def]
    if output_dtype is None:
        output_dtype = _float_result(data.dtype, scale.dtype, *(() if offset is None else (offset.dtype,)))
    return torch.empty(data.shape, dtype=output_dtype, device=data.device)


def _meta_quantize_dynamic_by_tile(data, tile_size, num_bits, symmetric, allow_one_sided, output_dtype):  # type: ignore[no-untyped-def]
    tile = 1
    for extent in tile_size:
        tile *= extent
    ntiles = data.numel() // tile if tile else 0
    if output_dtype is None:
        output_dtype = data.dtype if data.dtype in (torch.float32, torch.float64) else torch.float32
    params = lambda: torch.empty(ntiles, dtype=torch.float32, device=data.device)  # noqa: E731
    return torch.empty(data.shape, dtype=output_dtype, device=data.device), params(), params()


def _meta_quantize_by_tile_backward(data, output_grad, scale, tile_size, num_bits, offset=None):  # type: ignore[no-untyped-def]
    doffset = scale.new_empty(0) if offset is None else torch.empty_like(scale)
    return [torch.empty(data.shape, dtype=data.dtype, device=data.device), torch.empty_like(scale), doffset]


def _meta_linear_w8a8(x_codes, w_codes, x_scale, x_offset, w_scale, w_offset, bias, out_dtype, out_scale, out_offset, out_num_bits, w_rowsum, requant_from):  # type: ignore[no-untyped-def]
    return torch.empty((*x_codes.shape[:-1], w_codes.shape[0]), dtype=out_dtype, device=x_codes.device)


def _meta_bmm_w8a8(x_codes, w_codes, x_scale, x_offset, w_scale, w_offset, out_dtype, out_scale, out_offset, out_num_bits, requant_from):  # type: ignore[no-untyped-def]
    return torch.empty((x_codes.shape[0], x_codes.shape[1], w_codes.shape[1]), dtype=out_dtype, device=x_codes.device)


def _meta_linear_wq(x, w_codes, w_scale, w_offset, group, bias, out_dtype, pack_block, two_pass, split):  # type: ignore[no-untyped-def]
    n = w_codes.numel() * 2 // x.shape[-1] if pack_block > 0 else w_codes.shape[0]
    return torch.empty((*x.shape[:-1], n), dtype=out_dtype, device=x.device)


_LIBRARY.impl("running_minmax_step", lambda *args: None, "Meta")
_LIBRARY.impl("linear_w8a8", _meta_linear_w8a8, "Meta")
_LIBRARY.impl("bmm_w8a8", _meta_bmm_w8a8, "Meta")
_LIBRARY.impl("linear_wq", _meta_linear_wq, "Meta")
_LIBRARY.impl("quantize_by_tile", _meta_quantize_by_tile, "Meta")
_LIBRARY.impl("dequantize_by_tile", _meta_dequantize_by_tile, "Meta")
_LIBRARY.impl("quantize_dynamic_by_tile", _meta_quantize_dynamic_by_tile, "Meta")
_LIBRARY.impl("quantize_by_tile_backward", _meta_quantize_by_tile_backward, "Meta")


# Every operator above also has a C++ device kernel (csrc/ffq_torch.cpp -> csrc/libffq_torch.so, registered for the HIP
# dispatch key): torch.ops.fastforward_amd.* on a HIP tensor then runs dispatcher -> C++ -> the C ABI without entering the
# interpreter. Same library, same kernels, same results as the Python implementations above, which stay registered (and serve
# when the extension is absent or FFQ_NO_TORCH_EXT=1 — they are the HIP path too).
TORCH_EXTENSION_PATH = _native.LIBRARY_PATH.with_name("libffq_torch.so")


def _load_torch_extension() -> bool:
    import os
    import warnings

    if os.environ.get("FFQ_NO_TORCH_EXT") == "1" or not TORCH_EXTENSION_PATH.exists() or not _native.is_available():
        return False
    try:
        torch.ops.load_library(str(TORCH_EXTENSION_PATH))
    except OSError as e:  # built against another PyTorch
        warnings.warn(f"fastforward_amd: cannot load {TORCH_EXTENSION_PATH} ({e}); the operators run through the Python implementations")
        return False
    return True


NATIVE_DISPATCH: bool = _load_torch_extension()
