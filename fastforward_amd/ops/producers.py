"""The callers either side of the quantized linears as one-pass kernels with A1 fused in (reference docs/examples/doc_helpers/
quantized_llama/{rms_norm,mlp,rotary_embedding,attention}.py): residual add + RMSNorm, SiLU*up, rotary embedding, causal attention."""

from __future__ import annotations

import ctypes

from typing import Sequence

import torch

from fastforward_amd._cabi import FanOut
from fastforward_amd.ops import _base
from fastforward_amd.ops._base import _ptr, _tag


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
    lib, stream = _base._prepare(xc, dc, wc, *[t for q in quantizers for t in q])
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
    lib, stream = _base._prepare(gc, uc, *[t for q in quantizers for t in q])
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
    lib, stream = _base._prepare(q, k, cos, sin)
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
    lib, stream = _base._prepare(q, k, v, scale, offset, cos, sin)
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
