"""The reference's eager ATen chain for the hot path, restated with plain torch CPU ops.

TEST / BASELINE INFRASTRUCTURE ONLY — never imported by fastforward_amd. bench.py times these
functions on the GPU box's host cores for the `cpu_baseline` object ("kind": "port"), and
tests/test_eager_chain.py checks them against the golden fixtures (and therefore against the
reference and the C oracle).

Each function is the op sequence of the cited reference function, so that the baseline pays for the
same unfused passes and fp32 temporaries the reference pays for on a CPU:
  quantize    tiles_to_rows -> div -> sub -> round -> clamp -> rows_to_tiles -> to
              (src/fastforward/quantization/_quantizer_impl.py:154-169)
  dequantize  tiles_to_rows -> add -> mul -> rows_to_tiles -> to            (:181-190)
  minmax      tiles_to_rows -> min(-1), max(-1) -> isinf().any() -> running merge
              (src/fastforward/range_setting/minmax.py:227-237)
  params      src/fastforward/quantization/affine/range.py:90-122
  linear      dequantize, dequantize, F.linear, output quantizer (src/fastforward/_gen/fallback.py:94-111)
"""

from __future__ import annotations

import torch
import torch.nn.functional as F


def tiles_to_rows(data: torch.Tensor, tile) -> torch.Tensor:
    if data.numel() == 0:
        return data.reshape(1, 0)
    tile = tuple(data.shape) if isinstance(tile, str) else tuple(tile)
    split = []
    for n, t in zip(data.shape, tile):
        split += [n // t, t]
    rank = data.dim()
    perm = list(range(0, 2 * rank, 2)) + list(range(1, 2 * rank, 2))
    numel_tile = 1
    for t in tile:
        numel_tile *= t
    return data.reshape(split).permute(perm).reshape(data.numel() // numel_tile, -1)


def rows_to_tiles(rows: torch.Tensor, shape, tile) -> torch.Tensor:
    if rows.numel() == 0:
        return rows.reshape(tuple(shape))
    shape = tuple(shape)
    tile = shape if isinstance(tile, str) else tuple(tile)
    split = []
    for n, t in zip(shape, tile):
        split += [n // t, t]
    rank = len(shape)
    perm = list(range(0, 2 * rank, 2)) + list(range(1, 2 * rank, 2))
    inverse = [0] * len(perm)
    for dst, src in enumerate(perm):
        inverse[src] = dst
    return rows.reshape([split[p] for p in perm]).permute(inverse).reshape(shape)


def quantize(data, scale, tile, num_bits, output_dtype=None, offset=None):
    scale = scale.reshape(-1)
    offset = torch.round(offset.reshape(-1)) if offset is not None else torch.zeros_like(scale)
    lo = -(2 ** (num_bits - 1))
    hi = -lo - 1
    rows = tiles_to_rows(data, tile)
    q = torch.round(rows / scale[:, None] - offset[:, None])
    q = torch.clamp(q, lo, hi)
    out = rows_to_tiles(q, data.shape, tile)
    return out.to(output_dtype or out.dtype)


def dequantize(data, scale, tile, offset=None, output_dtype=None):
    scale = scale.reshape(-1)
    offset = torch.round(offset.reshape(-1)) if offset is not None else torch.zeros_like(scale)
    rows = tiles_to_rows(data, tile)
    out = rows_to_tiles((rows + offset[:, None]) * scale[:, None], data.shape, tile)
    return out.to(output_dtype) if output_dtype else out


def minmax(data, tile):
    rows = tiles_to_rows(data, tile)
    lo, hi = torch.min(rows, -1).values, torch.max(rows, -1).values
    if lo.isinf().any() or hi.isinf().any():
        raise NotImplementedError("Infinite")
    return lo, hi


def parameters_for_range(lo, hi, num_bits, symmetric, allow_one_sided):
    lo, hi = lo.to(torch.float32), hi.to(torch.float32)
    one_sided = bool(lo.min() >= 0) and allow_one_sided
    int_min = -(2 ** (num_bits - 1))
    int_max = -int_min - 1
    if symmetric and one_sided:
        lo = torch.zeros_like(lo)
    if symmetric and not one_sided:
        return torch.max(torch.abs(lo) / abs(int_min), torch.abs(hi) / abs(int_max)), None
    scale = ((hi - lo) / (2**num_bits - 1)).clamp(torch.finfo(torch.float32).eps)
    return scale, lo / scale - int_min


def linear_w8a8(x, weight, x_scale, x_offset, w_scale, w_offset, num_bits=8, bias=None):
    """QuantizedLinear.forward with initialised quantizers (reference nn/linear.py:32-39 + fallback)."""
    xq = quantize(x, x_scale, x.shape, num_bits, x.dtype, x_offset)                  # input quantizer
    wq = quantize(weight, w_scale, (1, weight.shape[1]), num_bits, weight.dtype, w_offset)  # weight, every call
    xd = dequantize(xq, x_scale, x.shape, x_offset, x.dtype)
    wd = dequantize(wq, w_scale, (1, weight.shape[1]), w_offset, weight.dtype)
    return F.linear(xd, wd, bias)
