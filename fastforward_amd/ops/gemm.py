"""A6 on int8 codes (reference _gen/fallback.py:77-112 and :699-798): the W8A8 linear with its optional fused output quantizer, the
forms range estimation needs (earlier codes, gated epilogue, the device-decided MLP), bmm, and gate + up + SiLU*up + quantize in one launch."""

from __future__ import annotations

import ctypes

import torch

from fastforward_amd import _native
from fastforward_amd.ops import _base
from fastforward_amd.ops._base import _extrema_words, _native_route, _ptr, _tag, _workspace


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
    lib, stream = _base._prepare(xc, wc, xs, xo, ws_, wo, bias_c, os_, oo)
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


def linear_w8a8_multi(
    x_codes: torch.Tensor,
    w_codes: torch.Tensor,
    x_scale: torch.Tensor,
    x_offset: torch.Tensor | None,
    w_scale: torch.Tensor,
    rows: Sequence[int],
    out_dtype: torch.dtype = torch.bfloat16,
    w_rowsum: torch.Tensor | None = None,
) -> list[torch.Tensor] | None:
    """Two or three W8A8 linears on the SAME activation codes in one launch (q_proj / k_proj / v_proj: three ``QuantizedLinear`` modules
    on one quantized hidden state, reference nn/linear.py:32-39): ``w_codes`` holds the matrices' int8 codes one after the other
    (``[sum(rows), K]``), ``w_scale`` (and ``w_rowsum``) their per-row scales (row sums) in the same order; returns one ``[..., rows[i]]``
    tensor per matrix, each bit for bit what :func:`linear_w8a8` returns for that matrix alone. None where the one-launch form does not
    apply (fewer than 64 tiles of 256 x 256, K % 128 != 0, a matrix but the last that is no multiple of 256 rows): the caller launches
    the linears one by one."""
    if x_codes.dtype != torch.int8 or w_codes.dtype != torch.int8:
        raise TypeError("linear_w8a8_multi expects int8 codes")
    count = len(rows)
    if not 2 <= count <= 3 or out_dtype not in (torch.float32, torch.bfloat16, torch.float16):
        return None
    xc, wc = x_codes.detach().contiguous(), w_codes.detach().contiguous()
    K, N = xc.shape[-1], int(sum(rows))
    M = xc.numel() // K if K else 0
    if wc.dim() != 2 or wc.shape != (N, K) or any(int(r) <= 0 for r in rows) or any(int(r) % 256 for r in rows[:-1]):
        return None

    def f32(t: torch.Tensor | None) -> torch.Tensor | None:
        return None if t is None else t.detach().reshape(-1).to(torch.float32).contiguous()

    xs, xo, ws_ = f32(x_scale), f32(x_offset), f32(w_scale)
    if xs.numel() not in (1, M) or ws_.numel() != N:
        return None
    if w_rowsum is not None and (w_rowsum.dtype != torch.int32 or w_rowsum.numel() != N or not w_rowsum.is_contiguous() or w_rowsum.device != wc.device):
        raise RuntimeError(f"w_rowsum must be a contiguous int32 tensor with {N} entries on the codes' device")
    lib, stream = _base._prepare(xc, wc, xs, xo, ws_)
    outs = [torch.empty((*xc.shape[:-1], int(r)), dtype=out_dtype, device=xc.device) for r in rows]
    nbytes = lib.ffq_linear_w8a8_workspace_bytes(M, N, K)
    ws = _workspace(nbytes, xc.device)
    status = lib.ffq_linear_w8a8_multi(
        _ptr(xc), _ptr(wc), _ptr(w_rowsum), _ptr(xs), _ptr(xo), int(xs.numel() != 1), _ptr(ws_), count, (ctypes.c_void_p * count)(*[_ptr(o) for o in outs]),
        _tag(out_dtype), M, (ctypes.c_int64 * count)(*[int(r) for r in rows]), K, _ptr(ws), nbytes, stream,
    )
    if status == 6:  # FFQ_ERR_DTYPE: not the persistent kernel's shape class
        return None
    lib.check(status)
    return outs


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
    lib, stream = _base._prepare(xc, ec, wc, xs, xo, ws_, wo, es, eo)
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
    lib, stream = _base._prepare(xc, wc, gc, xs, xo, ws_, wo)
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
    lib, stream = _base._prepare(xc, wc, xs, xo, ws_, wo, os_, oo)
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
    lib, stream = _base._prepare(xc, gc, uc, xs, xo, gs, us, os_, oo)
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
    lib, stream = _base._prepare(xg, xu, gc, uc, xsg, xsu, gs, us, xog, xou, go, uo)
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
