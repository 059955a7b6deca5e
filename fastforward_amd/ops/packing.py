"""A7 and the callers either side of it: int4 packing in the GGUF nibble order (reference export/stages/gguf/_packing.py:23-98) alone and
fused with A1 / A2, the GGUF block writers, the GPTQ column loop (quantization/gptq.py:101-235) and the grid estimator's error pass
(range_setting/min_error.py:171-231)."""

from __future__ import annotations

import ctypes

from typing import Sequence

import torch

from fastforward_amd.ops import _base
from fastforward_amd.ops._base import _flat, _ptr, _tag, _tile_of, _workspace
from fastforward_amd.ops.static import dequantize_by_tile, quantize_by_tile


def pack_int4(codes: torch.Tensor, block: int = 32) -> torch.Tensor:
    """A7 — pack codes in [-8, 7] two per byte, GGUF Q4_0 nibble order (reference _packing.py:44-53)."""
    codes_c = codes.detach().contiguous()
    lib, stream = _base._prepare(codes_c)
    n = codes_c.numel()
    out = torch.empty(n // 2, dtype=torch.uint8, device=codes_c.device)
    lib.check(lib.ffq_pack_int4(_ptr(codes_c), _tag(codes_c.dtype), n, int(block), _ptr(out), stream))
    return out


def unpack_int4(packed: torch.Tensor, shape: Sequence[int], dtype: torch.dtype = torch.int8, block: int = 32) -> torch.Tensor:
    """Inverse of :func:`pack_int4`: ``unpack_int4(pack_int4(q), q.shape, q.dtype) == q``."""
    packed_c = packed.detach().contiguous()
    lib, stream = _base._prepare(packed_c)
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
    lib, stream = _base._prepare(weights, quantized, errors, hessian_inverse, sc, of)
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
    lib, stream = _base._prepare(data_c, sc, of, out)
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
    lib, stream = _base._prepare(codes, sc)
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
        lib, stream = _base._prepare(data_c, scale_c, offset_c)
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
        lib, stream = _base._prepare(packed_c, scale_c, offset_c)
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
