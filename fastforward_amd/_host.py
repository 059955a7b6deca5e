"""Host-tensor route of the four operators, the min/max reduction and the range -> parameter rule.

BASELINE configs[0] is "single nn.Linear 1024 x 1024, 8-bit per-tensor weight LinearQuantizer on CPU eager (plumbing, no GPU)",
and the reference's quantizers default to ``device="cpu"`` (nn/linear_quantizer.py:147-173): its operators are device-agnostic
ATen chains (quantization/_quantizer_impl.py:144-285). The HIP kernels of this package take device pointers only, so a tensor that
lives in HOST memory takes the same chain here — tiles_to_rows, div, sub, round, clamp, rows_to_tiles, cast — written once more
in terms of this package's own ``tiled_tensor`` helpers. This is not a fallback of the device path:

* a HIP tensor never reaches this module (``ops._host_route`` looks at the tensor's device, nothing else), and without
  ``libffq_hip.so`` every operator on a HIP tensor still raises ``BackendError``;
* nothing here is timed, fused or tuned — host tensors are the plumbing case (conversion, calibration and a forward of a small
  module before it is moved to the GPU), checked against the golden fixtures G1-G5 in ``tests/test_host_route.py``;
* the quantized-operator dispatcher claims HIP operands only, so a quantized linear on host tensors runs the generated
  fallback — dequantize, ``F.linear``, output quantizer — exactly as in the reference (_gen/fallback.py:77-112).
"""

from __future__ import annotations

from typing import Sequence

import torch

from fastforward_amd.exceptions import QuantizationError
from fastforward_amd.quantization.tiled_tensor import rows_to_tiles, tiles_to_rows


# explicit mantissa bits by torch.finfo(...).dtype name; complex dtypes report their component's name
_MANTISSA_BITS = {"bfloat16": 7, "float16": 10, "float32": 23, "float64": 52,
                  "float8_e4m3fn": 3, "float8_e4m3fnuz": 3, "float8_e5m2": 2, "float8_e5m2fnuz": 2}


def can_support_bitwidth(dtype: torch.dtype, num_bits: float) -> bool:
    """reference _quantizer_impl.py:44-75: the container's precision bits + 2 (the first unrepresentable integer lies beyond
    2^(mantissa + 1), and the sign is a bit of its own) must cover `num_bits`; precision bits = explicit mantissa of a float
    (fp8 variants included, complex dtypes by their component), the width of an integer type, and — a float type the table does
    not know — `num_bits` itself, with a warning, as the reference decides."""
    if dtype.is_complex or dtype.is_floating_point:
        name = torch.finfo(dtype).dtype
        if name in _MANTISSA_BITS:
            precision = _MANTISSA_BITS[name]
        else:
            import logging

            logging.getLogger(__name__).warning(f"Unknown mantissa size for {dtype}; precision loss possible.")
            precision = num_bits
    else:
        precision = torch.iinfo(dtype).bits
    return precision + 2 >= num_bits


def _params(scale: torch.Tensor, offset: torch.Tensor | None, rows: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
    s = scale.reshape(-1)
    o = torch.round(offset.reshape(-1)) if offset is not None else torch.zeros_like(s)  # _infer_offset (:140-141)
    if rows.numel() and s.numel() not in (1, rows.shape[0]):
        if rows.shape[0] == 1:
            raise ValueError(f"tiled_data is expected to be of size (1, L) but scale has {s.numel()} entries")
        raise RuntimeError(f"The size of tensor a ({rows.shape[0]}) must match the size of tensor b ({s.numel()}) at non-singleton dimension 0")
    return s, o


def quantize_by_tile(data: torch.Tensor, scale: torch.Tensor, tile_size: Sequence[int], num_bits: float, output_dtype: torch.dtype | None,
                     offset: torch.Tensor | None = None) -> torch.Tensor:
    """A1 (:144-169)."""
    lo = -(2 ** (num_bits - 1))
    hi = -lo - 1
    rows = tiles_to_rows(data, tile_size)
    s, o = _params(scale, offset, rows)
    q = torch.clamp(torch.round(rows / s[:, None] - o[:, None]), lo, hi)
    out = rows_to_tiles(q, data.shape, tile_size)
    output_dtype = output_dtype or out.dtype
    if not can_support_bitwidth(output_dtype, num_bits):
        raise RuntimeError(f"Provided dtype ({output_dtype}) is not enough to store {num_bits} bits quantized values.")
    return out.to(output_dtype)


def dequantize_by_tile(data: torch.Tensor, scale: torch.Tensor, tile_size: Sequence[int], offset: torch.Tensor | None = None,
                       output_dtype: torch.dtype | None = None) -> torch.Tensor:
    """A2 (:172-190)."""
    rows = tiles_to_rows(data, tile_size)
    s, o = _params(scale, offset, rows)
    out = rows_to_tiles((rows + o[:, None]) * s[:, None], data.shape, tile_size)
    return out.to(output_dtype) if output_dtype is not None else out


def minmax_by_tile(data: torch.Tensor, tile_size: Sequence[int]) -> tuple[torch.Tensor, torch.Tensor]:
    """A4's reduction (range_setting/minmax.py:227-228): per-tile extrema in the data dtype."""
    if data.numel() == 0:
        raise QuantizationError(f"Cannot dynamically quantize an empty tensor of shape {tuple(data.shape)}")
    rows = tiles_to_rows(data, tile_size)
    return torch.min(rows, dim=1).values, torch.max(rows, dim=1).values


def parameters_for_range(min_range: torch.Tensor, max_range: torch.Tensor, num_bits: float, symmetric: bool, allow_one_sided: bool,
                         round_offset: bool = False) -> tuple[torch.Tensor, torch.Tensor | None]:
    """A5 (affine/range.py:54-122)."""
    lo, hi = min_range.reshape(-1).to(torch.float32), max_range.reshape(-1).to(torch.float32)
    one_sided = bool(lo.min() >= 0) and allow_one_sided
    int_min = -(2 ** (num_bits - 1))
    if symmetric and one_sided:
        lo = torch.zeros_like(lo)
    if symmetric and not one_sided:
        return torch.max(torch.abs(lo) / abs(int_min), torch.abs(hi) / abs(-int_min - 1)), None
    scale = ((hi - lo) / (2**num_bits - 1)).clamp(torch.finfo(torch.float32).eps)
    offset = lo / scale - int_min
    return scale, torch.round(offset) if round_offset else offset


def quantize_dynamic_by_tile(data: torch.Tensor, tile_size: Sequence[int], num_bits: float, symmetric: bool, allow_one_sided: bool,
                             output_dtype: torch.dtype | None) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """A3 (:243-285)."""
    lo, hi = minmax_by_tile(data, tile_size)
    scale, offset = parameters_for_range(lo, hi, num_bits, symmetric, allow_one_sided)
    offset = torch.round(offset) if offset is not None else torch.zeros_like(scale)
    rows = tiles_to_rows(data, tile_size)
    bound = -(2 ** (num_bits - 1))
    q = torch.clamp(torch.round(rows / scale[:, None] - offset[:, None]), bound, -bound - 1)
    out = rows_to_tiles(q, data.shape, tile_size)
    output_dtype = output_dtype or out.dtype
    if not can_support_bitwidth(output_dtype, num_bits):
        raise RuntimeError(f"Provided dtype ({output_dtype}) is not enough to store {num_bits} bits quantized values.")
    return out.to(output_dtype), scale, offset
