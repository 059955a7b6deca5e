"""The W8A8 linear — and mm / matmul / bmm — registered in the quantized-operator dispatcher (plug-in seam #2).

The reference registers no kernel for ``"linear"``; every quantized linear therefore runs
``fallback.linear`` (src/fastforward/_gen/fallback.py:77-112): dequantize x, dequantize w, float
GEMM, output quantizer. This module registers a kernel whose predicate accepts the combinations the
int8-MFMA kernel A6 covers and returns False for everything else, so the reference behaviour (the
float fallback in :mod:`fastforward_amd.nn.functional`) still applies there.

The kernel accepts both calling conventions the dispatcher produces:
``kernel(input, weight, bias)`` from ``QuantizedTensor.__torch_function__`` when user code calls
``torch.nn.functional.linear`` on quantized tensors, and
``kernel(input=..., weight=..., bias=..., output_quantizer=..., strict_quantization=...)`` from
``fastforward_amd.nn.functional.linear``.
"""

from __future__ import annotations

import contextlib
import os
import weakref

from typing import Any

import torch

from fastforward_amd import _native, ops
from fastforward_amd.dispatcher import Predicate, register
from fastforward_amd.exceptions import QuantizationError
from fastforward_amd.quantization import granularity as granularities
from fastforward_amd.quantization._linear_quantized_ops import _static_affine
from fastforward_amd.quantized_tensor import QuantizedTensor


def _row_mode(tensor: QuantizedTensor) -> str | None:
    """'tensor' (one parameter pair), 'row' (one pair per row of the last-dim-contiguous matrix), or None."""
    params = tensor.quantization_context.quantization_params
    tile = params.granularity.tile_size(tensor.shape)
    if isinstance(tile, str) or tuple(tile) == tuple(tensor.shape):
        return "tensor"
    if tensor.dim() >= 1 and all(t == 1 for t in tile[:-1]) and tile[-1] == tensor.shape[-1]:
        return "row"
    return None


def _supported(input: Any, weight: Any, bias: Any = None, **_: Any) -> bool:
    if not (_static_affine(input) and _static_affine(weight)):
        return False
    if not _on_backend(input, weight) or weight.dim() != 2 or input.dim() < 1:
        return False
    if input.shape[-1] != weight.shape[1] or weight.shape[1] % 16 != 0 or input.numel() == 0:
        return False
    xp, wp = input.quantization_context.quantization_params, weight.quantization_context.quantization_params
    if xp.num_bits > 8 or wp.num_bits > 8 or xp.num_bits != int(xp.num_bits) or wp.num_bits != int(wp.num_bits):
        return False
    if _row_mode(input) is None or _row_mode(weight) is None:
        return False
    deq = xp.dequantize_dtype or torch.get_default_dtype()
    if deq not in (torch.bfloat16, torch.float16, torch.float32):
        return False
    if (wp.dequantize_dtype or deq) != deq:
        return False  # the float fallback would raise on mixed dtypes; let it
    if isinstance(bias, QuantizedTensor) and not _static_affine(bias):
        return False
    return True


def _int8_codes(tensor: QuantizedTensor) -> torch.Tensor:
    """Codes as int8. Float / wider containers hold the same integers; they are
    converted exactly by A1 with scale 1 (x / 1 - 0, round, clamp to the int8 range)."""
    raw = tensor.raw_data
    if raw.dtype == torch.int8:
        return raw
    one = torch.ones(1, dtype=torch.float32, device=raw.device)
    return ops.quantize_by_tile(raw, one, raw.shape, 8, torch.int8)


# A symmetric quantizer carries an offset BUFFER that is all zeros unless its data is one-sided (reference
# nn/linear_quantizer.py:164-170). Knowing that on the host lets the GEMM skip the weight-offset terms and take its
# persistent form; the answer is read ONCE per (tensor object, version) — the range setter bumps the version — and never
# while a hipGraph is being captured (then the offset is simply passed on: same result, the kernel checks it on the device).
_ZERO_OFFSETS: dict[int, tuple[Any, int, bool | None]] = {}


def _known_zero(offset: Any) -> bool:
    if not isinstance(offset, torch.Tensor):
        return False
    hit = _ZERO_OFFSETS.get(id(offset))
    seen = hit is not None and hit[0]() is offset and hit[1] == offset._version
    if seen and hit[2] is not None:
        return hit[2]
    if offset.is_cuda and torch.cuda.is_current_stream_capturing():
        return False
    if len(_ZERO_OFFSETS) > 4096:  # entries of tensors that are gone
        for key in [k for k, v in _ZERO_OFFSETS.items() if v[0]() is None]:
            del _ZERO_OFFSETS[key]
    if not seen:
        # first sighting of this version: no host read yet. A range estimator re-sets the range on every call (a new
        # version each time), and a read-back per linear would serialise the sync-free calibration with the host.
        _ZERO_OFFSETS[id(offset)] = (weakref.ref(offset), offset._version, None)
        return False
    zero = not bool(offset.detach().any())  # the version was stable across two calls: read it once
    _ZERO_OFFSETS[id(offset)] = (weakref.ref(offset), offset._version, zero)
    return zero


def fused_linear(input: QuantizedTensor, weight: QuantizedTensor, bias: Any = None, *, output_quantizer: Any = None, strict_quantization: bool | None = None) -> torch.Tensor:
    if strict_quantization and output_quantizer is None:
        raise QuantizationError("'output_quantizer' must be provided if strict_quantization=True")
    xp, wp = input.quantization_context.quantization_params, weight.quantization_context.quantization_params
    deq = xp.dequantize_dtype or torch.get_default_dtype()
    if isinstance(bias, QuantizedTensor):
        bias = bias.dequantize()
    w_offset = None if wp.offset is None or _known_zero(wp.offset) else torch.as_tensor(wp.offset, device=weight.device)
    out = ops.linear_w8a8(
        _int8_codes(input), _int8_codes(weight),
        x_scale=torch.as_tensor(xp.scale, device=input.device), x_offset=None if xp.offset is None else torch.as_tensor(xp.offset, device=input.device),
        w_scale=torch.as_tensor(wp.scale, device=weight.device), w_offset=w_offset,
        bias=bias, out_dtype=deq,
    )
    return output_quantizer(out) if output_quantizer is not None else out


fused_linear_predicate = Predicate(_supported)
_registration = register("linear", fused_linear_predicate, fused_linear)


# ---------------------------------------------------------------------------------------------------------------
# Weight-only: a static-affine QuantizedTensor weight and a PLAIN bf16 input (BASELINE configs 2 and 4; reference
# fallback.py:86-112 with strict quantization off: weight.dequantize() + F.linear). The kernel dequantizes the codes in
# the GEMM's operand load (ops.linear_wq); weight granularities: per tensor, per output channel, or groups of G input
# channels per output channel (PerBlock(block_dims=1, block_sizes=G, per_channel_dims=0), G % 64 == 0).
# ---------------------------------------------------------------------------------------------------------------
# Which GEMM a weight-only linear takes. Measured on the MI355X at T = 16384 on the Llama-3-8B shapes
# (tools/wq_time.py, profiles/r02_wq_time.txt): the hand-written kernel 1.15 PFLOP/s (group-128: 1.05), A2 into a bf16
# tensor + the vendor's hand-tuned bf16 GEMM 1.55 PFLOP/s — the dequantization pass it saves (3 B/elem, 3.7 ms per
# forward) is worth less than the GEMM gap (51 ms). The dispatcher therefore only claims weight-only linears when asked
# to: ``with ff.fused_linear.weight_only_kernel(True)`` or FFQ_WEIGHT_ONLY_KERNEL=1; otherwise they run the reference's
# path (fallback.py:86-112: A2 + F.linear). Both produce the same operands bit for bit.
_WEIGHT_ONLY_KERNEL = os.environ.get("FFQ_WEIGHT_ONLY_KERNEL", "0") not in ("0", "", "false", "no")


@contextlib.contextmanager
def weight_only_kernel(enabled: bool = True):
    """Route weight-only quantized linears (plain bf16 input, quantized weight) to the hand-written bf16 x int8-code
    kernel (``ops.linear_wq``) inside the block."""
    global _WEIGHT_ONLY_KERNEL
    previous, _WEIGHT_ONLY_KERNEL = _WEIGHT_ONLY_KERNEL, bool(enabled)
    try:
        yield
    finally:
        _WEIGHT_ONLY_KERNEL = previous


def _weight_group(weight: QuantizedTensor) -> int | None:
    """Input channels sharing one parameter pair within a row ([N, K / group] parameters), or None if not covered."""
    tile = weight.quantization_context.quantization_params.granularity.tile_size(weight.shape)
    n, k = weight.shape
    if isinstance(tile, str) or tuple(tile) == (n, k):
        return k  # per tensor: one pair
    if tile[0] == 1 and k % tile[1] == 0:
        return int(tile[1])  # (1, K): per output channel; (1, G): groups along the input channels
    return None


def _supported_weight_only(input: Any, weight: Any, bias: Any = None, **_: Any) -> bool:
    if not _WEIGHT_ONLY_KERNEL or isinstance(input, QuantizedTensor) or not isinstance(input, torch.Tensor) or not _static_affine(weight):
        return False
    if not _on_backend(input, weight) or weight.dim() != 2 or input.dim() < 1 or input.numel() == 0:
        return False
    if input.dtype != torch.bfloat16 or input.shape[-1] != weight.shape[1]:
        return False
    wp = weight.quantization_context.quantization_params
    if wp.num_bits > 8 or wp.num_bits != int(wp.num_bits) or (wp.dequantize_dtype or input.dtype) != input.dtype:
        return False
    group = _weight_group(weight)
    if group is None:
        return False
    k = weight.shape[1]
    if k % 64 or k < 128 or (group != k and group % 64):
        return False
    return not (isinstance(bias, QuantizedTensor) and not _static_affine(bias))


def fused_linear_weight_only(input: torch.Tensor, weight: QuantizedTensor, bias: Any = None, *, output_quantizer: Any = None, strict_quantization: bool | None = None) -> torch.Tensor:
    if strict_quantization:  # the reference's messages, fallback.py:83-92
        if output_quantizer is None:
            raise QuantizationError("'output_quantizer' must be provided if strict_quantization=True")
        raise QuantizationError("Expected 'input' to be an instance of 'QuantizedTensor' because strict_quantization=True.")
    wp = weight.quantization_context.quantization_params
    if isinstance(bias, QuantizedTensor):
        bias = bias.dequantize()
    out = ops.linear_wq(
        input, _int8_codes(weight), torch.as_tensor(wp.scale, device=weight.device),
        None if wp.offset is None else torch.as_tensor(wp.offset, device=weight.device),
        group=_weight_group(weight), bias=bias, out_dtype=input.dtype,
    )
    if out is None:  # a shape the kernel does not cover after all: the reference's path
        out = torch.nn.functional.linear(input, weight.dequantize(), bias)
    return output_quantizer(out) if output_quantizer is not None else out


fused_linear_weight_only_predicate = Predicate(_supported_weight_only)
_registration_weight_only = register("linear", fused_linear_weight_only_predicate, fused_linear_weight_only)


# ---------------------------------------------------------------------------------------------------------------
# mm / matmul / bmm: the same fallback pattern in the reference (src/fastforward/_gen/fallback.py:699-798: dequantize both
# operands, float matmul, output quantizer), the same int8 contraction here. The right operand arrives as [K, N]; the GEMM
# contracts K-contiguous rows, so its codes are transposed once (1 B/elem; free when the operand is itself a transposed
# view of a K-contiguous tensor, e.g. ``k.transpose(-1, -2)``). Per-tensor or per-COLUMN parameters on the right operand
# (PerChannel(-1): one pair per output column), per-tensor or per-row on the left; everything else takes the float fallback.
# ---------------------------------------------------------------------------------------------------------------
def _col_mode(tensor: QuantizedTensor) -> str | None:
    """'tensor' or 'col' (one parameter pair per column of a [K, N] matrix) for the right operand of a matmul."""
    params = tensor.quantization_context.quantization_params
    tile = params.granularity.tile_size(tensor.shape)
    if isinstance(tile, str) or tuple(tile) == tuple(tensor.shape):
        return "tensor"
    if tensor.dim() == 2 and tile[0] == tensor.shape[0] and tile[1] == 1:
        return "col"
    return None


def _on_backend(*tensors: Any) -> bool:
    """All operands on a HIP device and the backend library loadable — anything else takes the float fallback."""
    return all(t.is_cuda for t in tensors) and _native.is_available()


def _bits_and_dtypes_ok(a: QuantizedTensor, b: QuantizedTensor) -> bool:
    ap, bp = a.quantization_context.quantization_params, b.quantization_context.quantization_params
    if ap.num_bits > 8 or bp.num_bits > 8 or ap.num_bits != int(ap.num_bits) or bp.num_bits != int(bp.num_bits):
        return False
    deq = ap.dequantize_dtype or torch.get_default_dtype()
    return deq in (torch.bfloat16, torch.float16, torch.float32) and (bp.dequantize_dtype or deq) == deq


def _supported_mm(input: Any = None, other: Any = None, mat2: Any = None, **_: Any) -> bool:
    right = other if other is not None else mat2
    if not (_static_affine(input) and _static_affine(right)) or not _on_backend(input, right):
        return False
    if right.dim() != 2 or input.dim() < 1 or input.shape[-1] != right.shape[0] or right.shape[0] % 16 or input.numel() == 0 or right.numel() == 0:
        return False
    return _bits_and_dtypes_ok(input, right) and _row_mode(input) is not None and _col_mode(right) is not None


def _params_of(t: QuantizedTensor) -> tuple[torch.Tensor, torch.Tensor | None]:
    p = t.quantization_context.quantization_params
    return torch.as_tensor(p.scale, device=t.device), None if p.offset is None else torch.as_tensor(p.offset, device=t.device)


def fused_mm(input: QuantizedTensor, other: QuantizedTensor | None = None, *, mat2: QuantizedTensor | None = None, output_quantizer: Any = None,
             strict_quantization: bool | None = None) -> torch.Tensor:
    right = other if other is not None else mat2
    if strict_quantization and output_quantizer is None:
        raise QuantizationError("'output_quantizer' must be provided if strict_quantization=True")
    deq = input.quantization_context.quantization_params.dequantize_dtype or torch.get_default_dtype()
    (xs, xo), (ws, wo) = _params_of(input), _params_of(right)
    w_codes = _int8_codes(right).t().contiguous()  # [N, K]
    out = ops.linear_w8a8(_int8_codes(input), w_codes, x_scale=xs, x_offset=xo, w_scale=ws, w_offset=wo, bias=None, out_dtype=deq)
    return output_quantizer(out) if output_quantizer is not None else out


def _supported_bmm(input: Any = None, mat2: Any = None, **_: Any) -> bool:
    if not (_static_affine(input) and _static_affine(mat2)) or not _on_backend(input, mat2):
        return False
    if input.dim() != 3 or mat2.dim() != 3 or input.shape[0] != mat2.shape[0] or input.shape[2] != mat2.shape[1]:
        return False
    if input.shape[2] % 16 or input.numel() == 0 or mat2.numel() == 0 or input.shape[0] > 256:
        return False
    # one parameter pair for each operand: the batch shares it, every matrix of the batch is one GEMM
    return _bits_and_dtypes_ok(input, mat2) and _row_mode(input) == "tensor" and _row_mode(mat2) == "tensor"


def fused_bmm(input: QuantizedTensor, mat2: QuantizedTensor, *, output_quantizer: Any = None, strict_quantization: bool | None = None) -> torch.Tensor:
    if strict_quantization and output_quantizer is None:
        raise QuantizationError("'output_quantizer' must be provided if strict_quantization=True")
    deq = input.quantization_context.quantization_params.dequantize_dtype or torch.get_default_dtype()
    (xs, xo), (ws, wo) = _params_of(input), _params_of(mat2)
    x_codes, w_codes = _int8_codes(input), _int8_codes(mat2).transpose(1, 2).contiguous()  # [B, N, K]
    out = torch.stack([ops.linear_w8a8(x_codes[b], w_codes[b], x_scale=xs, x_offset=xo, w_scale=ws, w_offset=wo, bias=None, out_dtype=deq)
                       for b in range(x_codes.shape[0])])
    return output_quantizer(out) if output_quantizer is not None else out


fused_mm_predicate = Predicate(_supported_mm)
fused_bmm_predicate = Predicate(_supported_bmm)
_registration_mm = register("mm", fused_mm_predicate, fused_mm)
_registration_matmul = register("matmul", fused_mm_predicate, fused_mm)
_registration_bmm = register("bmm", fused_bmm_predicate, fused_bmm)
