"""A4 and A5: per-tile min / max (RunningMinMax, reference range_setting/minmax.py:215-239), the estimator step as one backend call, the
step fused with the quantizer's forward, and ``parameters_for_range`` (quantization/affine/range.py:54-122)."""

from __future__ import annotations

import ctypes

from typing import Sequence

import torch

from fastforward_amd import _host
from fastforward_amd._cabi import FLAG_INF, FLAG_NAN
from fastforward_amd.ops import _base
from fastforward_amd.ops._base import _flat, _host_route, _native_route, _ptr, _tag, _tickets, _tile_of, _workspace


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
    lib, stream = _base._prepare(data_c, running_min, running_max, status_flags)
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
    lib, stream = _base._prepare(data_c, running_min, running_max, status_flags, scale_out, offset_out)
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
    lib, stream = _base._prepare(data_c, running_min, running_max, scale_out, offset_out, status_flags)
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
    lib, stream = _base._prepare(mn, mx, scale_out, offset_out)
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
