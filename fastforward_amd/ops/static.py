"""The four registry operators of the reference (A1 / A2 / A3 / A8: ``fastforward::{quantize,dequantize,quantize_dynamic}_by_tile``,
``quantize_by_tile_backward``, reference quantization/_quantizer_impl.py:144-285) and the row-batched / sibling forms of A1 the Llama
harness uses. Each call is one enqueue through the C ABI of include/ffq.h on torch's current HIP stream."""

from __future__ import annotations

import ctypes

from typing import Sequence

import torch

from fastforward_amd import _host
from fastforward_amd._cabi import FFQ_MAX_BATCH, DType, RowsBatch
from fastforward_amd.ops import _base
from fastforward_amd.ops._base import _DTYPES, _flat, _host_route, _ptr, _tag, _tickets, _tile_of, _workspace


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
    lib, stream = _base._prepare(data_c, scale_c, offset_c)
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
    lib, stream = _base._prepare(data_c, scale_c, offset_c)
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
    lib, stream = _base._prepare(data_c)
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
        lib, stream = _base._prepare(data_c, grad_c, scale_c, offset_c)
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
    lib, stream = _base._prepare(data_c, scale, offset, earlier_scale, earlier_offset)
    out = torch.empty(data_c.shape, dtype=torch.int8, device=data_c.device)
    status = lib.ffq_quantize_by_tile_unless_same(
        _ptr(data_c), _tag(data_c.dtype), _ptr(scale.detach()), _ptr(None if offset is None else offset.detach()), data_c.numel(), float(num_bits),
        _ptr(earlier_scale.detach()), _ptr(None if earlier_offset is None else earlier_offset.detach()), _ptr(out), stream,
    )
    if status == 6:  # FFQ_ERR_DTYPE: alignment
        return None
    lib.check(status)
    return out


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
    lib, stream = _base._prepare(wd, sc, of)
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
    rowsums: Sequence[torch.Tensor] | None = None, codes_out: Sequence[torch.Tensor] | None = None,
) -> list[torch.Tensor] | None:
    """A1 of up to 8 ``[rows, cols]`` bf16 weights with one (scale, offset) per row into int8 codes, ONE launch; each result
    equals ``quantize_by_tile(weight, scale, (1, cols), num_bits, torch.int8, offset)``. The seven linears of a decoder layer
    are re-quantized on every forward (reference nn/linear.py:34); as seven launches the short ones (k_proj / v_proj) run far
    below the streaming rate. `rowsums` (one ZEROED contiguous int32 [rows] tensor per weight, cols % 1024 == 0): the launch also
    adds each row's code sum into it — what :func:`linear_w8a8` takes as ``w_rowsum``. Returns None where the one-launch
    kernel does not apply (then quantize member by member). `codes_out`: int8 tensors to receive the codes instead of fresh ones."""
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
    lib, stream = _base._prepare(*[w.detach() for w in weights], *sc, *[o for o in of if o is not None])
    if codes_out is not None:  # the caller's own int8 tensors (e.g. slices of one buffer: q / k / v codes side by side for linear_w8a8_multi)
        if len(codes_out) != len(weights) or any(c.dtype != torch.int8 or c.shape != w.shape or not c.is_contiguous() or c.device != w.device or c.data_ptr() % 16
                                                 for c, w in zip(codes_out, weights)):
            return None
        codes = list(codes_out)
    else:
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
