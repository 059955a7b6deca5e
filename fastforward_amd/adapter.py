"""Attach the HIP kernels to a real ``fastforward`` installation (when it is importable).

The reference defines its four hot-path ops with ``torch.library.custom_op`` (reference
src/fastforward/quantization/_quantizer_impl.py:127-134); each resulting ``CustomOpDef`` exposes
``register_kernel(device_type)``. ``install()`` registers this package's C-ABI-backed functions as the
``"cuda"`` (= HIP on ROCm) kernels of those ops, and registers the fused W8A8 linear in the
reference's own operator dispatcher (reference src/fastforward/dispatcher.py:233-265). After that an
unmodified FastForward program runs its fake-quantization hot path on the MI355X kernels:

    import fastforward as ff, fastforward_amd.adapter
    fastforward_amd.adapter.install()
    ff.quantize_model(model) ...           # the usual recipe, unchanged

Nothing here is needed (or importable) on a machine without the reference; this package's own
surface (``fastforward_amd.nn`` etc.) calls the same kernels directly.
"""

from __future__ import annotations

from typing import Any, Sequence

import torch

from fastforward_amd import ops


def install(device_types: Sequence[str] = ("cuda",), register_linear: bool = True) -> list[str]:
    """Returns the names of the reference hooks that were attached."""
    import fastforward as ff

    from fastforward.quantization import _quantizer_impl as impl

    global _DEVICE_TYPES
    _DEVICE_TYPES = tuple(device_types)
    attached = []
    table = {
        "quantize_by_tile_impl": ops.quantize_by_tile,
        "dequantize_by_tile_impl": ops.dequantize_by_tile,
        "quantize_dynamic_by_tile_impl": ops.quantize_dynamic_by_tile,
        "quant_dequant_by_tile_grad_impl": ops.quantize_by_tile_backward,
    }
    for attr, fn in table.items():
        op_def = getattr(impl, attr)
        for device_type in device_types:
            op_def.register_kernel(device_type)(fn)
        attached.append(f"fastforward::{op_def._opname if hasattr(op_def, '_opname') else attr}")
    if register_linear:
        on = lambda name, predicate, kernel: ff.dispatcher.register(name, ff.dispatcher.Predicate(predicate), kernel)  # noqa: E731
        on("linear", _reference_predicate, _reference_linear)
        on("linear", _reference_weight_only_predicate, _reference_weight_only_linear)
        # mm / matmul / bmm: the same fallback pattern in the reference (_gen/fallback.py:699-798), the same int8 GEMM here
        on("mm", _reference_mm_predicate, _reference_mm)
        on("matmul", _reference_mm_predicate, _reference_mm)
        on("bmm", _reference_bmm_predicate, _reference_bmm)
        attached += ["dispatcher:linear", "dispatcher:linear(weight-only)", "dispatcher:mm", "dispatcher:matmul", "dispatcher:bmm"]
    return attached


# device type the hooks accept; install(device_types=("cpu",)) in the adapter test drives them with the oracle injected
_DEVICE_TYPES: tuple[str, ...] = ("cuda",)


def _params(t: Any) -> Any:
    return t.quantization_context.quantization_params


def _reference_predicate(input: Any = None, weight: Any = None, bias: Any = None, **_: Any) -> bool:
    """Same acceptance rule as fastforward_amd.fused_linear, written against the reference's types."""
    import fastforward as ff

    from fastforward.quantization.affine import AffineQuantizationFunction, StaticAffineQuantParams

    for t in (input, weight):
        if not isinstance(t, ff.QuantizedTensor) or t.device.type not in _DEVICE_TYPES:
            return False
        ctx = t.quantization_context
        if not (issubclass(ctx.quantization_fn, AffineQuantizationFunction) and isinstance(ctx.quantization_params, StaticAffineQuantParams)):
            return False
        if ctx.quantization_params.num_bits > 8:
            return False
    if weight.dim() != 2 or input.shape[-1] != weight.shape[1] or weight.shape[1] % 16 != 0:
        return False
    for t in (input, weight):
        tile = _params(t).granularity.tile_size(t.shape)
        whole = isinstance(tile, str) or tuple(tile) == tuple(t.shape)
        per_row = not isinstance(tile, str) and all(v == 1 for v in tile[:-1]) and tile[-1] == t.shape[-1]
        if not (whole or per_row):
            return False
    return (_params(input).dequantize_dtype or torch.float32) in (torch.bfloat16, torch.float16, torch.float32)


def _codes(t: Any) -> torch.Tensor:
    raw = t.raw_data
    if raw.dtype == torch.int8:
        return raw
    return ops.quantize_by_tile(raw, torch.ones(1, dtype=torch.float32, device=raw.device), raw.shape, 8, torch.int8)


def _reference_linear(input: Any, weight: Any, bias: Any = None, *, output_quantizer: Any = None, strict_quantization: bool | None = None) -> torch.Tensor:
    import fastforward as ff

    if strict_quantization and output_quantizer is None:
        raise ff.exceptions.QuantizationError("'output_quantizer' must be provided if strict_quantization=True")
    xp, wp = _params(input), _params(weight)
    if isinstance(bias, ff.QuantizedTensor):
        bias = bias.dequantize()
    as_t = lambda v, dev: None if v is None else torch.as_tensor(v, device=dev)  # noqa: E731
    out = ops.linear_w8a8(
        _codes(input), _codes(weight), as_t(xp.scale, input.device), as_t(xp.offset, input.device),
        as_t(wp.scale, weight.device), as_t(wp.offset, weight.device), bias=bias,
        out_dtype=xp.dequantize_dtype or torch.float32,
    )
    return output_quantizer(out) if output_quantizer is not None else out


# ---- the other dispatcher hooks, written against the reference's types ---------------------------------------------------
def _static_affine_ref(t: Any, max_bits: int = 8) -> bool:
    import fastforward as ff

    from fastforward.quantization.affine import AffineQuantizationFunction, StaticAffineQuantParams

    if not isinstance(t, ff.QuantizedTensor) or t.device.type not in _DEVICE_TYPES:
        return False
    ctx = t.quantization_context
    if not (issubclass(ctx.quantization_fn, AffineQuantizationFunction) and isinstance(ctx.quantization_params, StaticAffineQuantParams)):
        return False
    bits = ctx.quantization_params.num_bits
    return bits <= max_bits and bits == int(bits)


def _tile(t: Any) -> tuple[int, ...]:
    tile = _params(t).granularity.tile_size(t.shape)
    return tuple(t.shape) if isinstance(tile, str) else tuple(tile)


def _scale_offset(t: Any) -> tuple[torch.Tensor, torch.Tensor | None]:
    p = _params(t)
    return torch.as_tensor(p.scale, device=t.device), None if p.offset is None else torch.as_tensor(p.offset, device=t.device)


def _reference_weight_only_predicate(input: Any = None, weight: Any = None, bias: Any = None, **_: Any) -> bool:
    """A plain bf16 input and a static-affine quantized weight (reference fallback.py:86-112, strict off): per tensor,
    per output channel, or groups of a multiple of 64 input channels per output channel."""
    import fastforward as ff

    if isinstance(input, ff.QuantizedTensor) or not isinstance(input, torch.Tensor) or not _static_affine_ref(weight):
        return False
    if input.device.type not in _DEVICE_TYPES or input.dtype != torch.bfloat16 or weight.dim() != 2 or input.numel() == 0:
        return False
    n, k = weight.shape
    if input.shape[-1] != k or k % 64 or k < 128 or (_params(weight).dequantize_dtype or input.dtype) != input.dtype:
        return False
    tile = _tile(weight)
    return tile == (n, k) or (tile[0] == 1 and k % tile[1] == 0 and (tile[1] == k or tile[1] % 64 == 0))


def _reference_weight_only_linear(input: Any, weight: Any, bias: Any = None, *, output_quantizer: Any = None, strict_quantization: bool | None = None) -> torch.Tensor:
    import fastforward as ff

    if strict_quantization:
        raise ff.exceptions.QuantizationError("Expected 'input' to be an instance of 'QuantizedTensor' because strict_quantization=True.")
    scale, offset = _scale_offset(weight)
    tile = _tile(weight)
    if isinstance(bias, ff.QuantizedTensor):
        bias = bias.dequantize()
    out = ops.linear_wq(input, _codes(weight), scale, offset, group=weight.shape[1] if tile == tuple(weight.shape) else tile[1], bias=bias, out_dtype=input.dtype)
    if out is None:
        out = torch.nn.functional.linear(input, weight.dequantize(), bias)
    return output_quantizer(out) if output_quantizer is not None else out


def _reference_mm_predicate(input: Any = None, other: Any = None, mat2: Any = None, **_: Any) -> bool:
    right = other if other is not None else mat2
    if not (_static_affine_ref(input) and _static_affine_ref(right)) or right.dim() != 2 or input.dim() < 1:
        return False
    if input.shape[-1] != right.shape[0] or right.shape[0] % 16 or input.numel() == 0 or right.numel() == 0:
        return False
    it, rt = _tile(input), _tile(right)
    left_ok = it == tuple(input.shape) or (all(v == 1 for v in it[:-1]) and it[-1] == input.shape[-1])
    right_ok = rt == tuple(right.shape) or (rt[0] == right.shape[0] and rt[1] == 1)  # per tensor or per column
    deq = _params(input).dequantize_dtype or torch.float32
    return left_ok and right_ok and deq in (torch.bfloat16, torch.float16, torch.float32) and (_params(right).dequantize_dtype or deq) == deq


def _reference_mm(input: Any, other: Any = None, *, mat2: Any = None, output_quantizer: Any = None, strict_quantization: bool | None = None) -> torch.Tensor:
    import fastforward as ff

    right = other if other is not None else mat2
    if strict_quantization and output_quantizer is None:
        raise ff.exceptions.QuantizationError("'output_quantizer' must be provided if strict_quantization=True")
    (xs, xo), (ws, wo) = _scale_offset(input), _scale_offset(right)
    out = ops.linear_w8a8(_codes(input), _codes(right).t().contiguous(), xs, xo, ws, wo, bias=None, out_dtype=_params(input).dequantize_dtype or torch.float32)
    return output_quantizer(out) if output_quantizer is not None else out


def _reference_bmm_predicate(input: Any = None, mat2: Any = None, **_: Any) -> bool:
    if not (_static_affine_ref(input) and _static_affine_ref(mat2)) or input.dim() != 3 or mat2.dim() != 3:
        return False
    if input.shape[0] != mat2.shape[0] or input.shape[2] != mat2.shape[1] or input.shape[2] % 16 or input.shape[0] > 256 or input.numel() == 0 or mat2.numel() == 0:
        return False
    deq = _params(input).dequantize_dtype or torch.float32
    return _tile(input) == tuple(input.shape) and _tile(mat2) == tuple(mat2.shape) and deq in (torch.bfloat16, torch.float16, torch.float32) and (_params(mat2).dequantize_dtype or deq) == deq


def _reference_bmm(input: Any, mat2: Any, *, output_quantizer: Any = None, strict_quantization: bool | None = None) -> torch.Tensor:
    import fastforward as ff

    if strict_quantization and output_quantizer is None:
        raise ff.exceptions.QuantizationError("'output_quantizer' must be provided if strict_quantization=True")
    (xs, xo), (ws, wo) = _scale_offset(input), _scale_offset(mat2)
    x_codes, w_codes = _codes(input), _codes(mat2).transpose(1, 2).contiguous()
    deq = _params(input).dequantize_dtype or torch.float32
    out = torch.stack([ops.linear_w8a8(x_codes[b], w_codes[b], xs, xo, ws, wo, bias=None, out_dtype=deq) for b in range(x_codes.shape[0])])
    return output_quantizer(out) if output_quantizer is not None else out
