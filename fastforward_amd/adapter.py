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
        ff.dispatcher.register("linear", ff.dispatcher.Predicate(_reference_predicate), _reference_linear)
        attached.append("dispatcher:linear")
    return attached


def _params(t: Any) -> Any:
    return t.quantization_context.quantization_params


def _reference_predicate(input: Any = None, weight: Any = None, bias: Any = None, **_: Any) -> bool:
    """Same acceptance rule as fastforward_amd.fused_linear, written against the reference's types."""
    import fastforward as ff

    from fastforward.quantization.affine import AffineQuantizationFunction, StaticAffineQuantParams

    for t in (input, weight):
        if not isinstance(t, ff.QuantizedTensor) or not t.is_cuda:
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
