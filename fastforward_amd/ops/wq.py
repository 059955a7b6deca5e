"""A6 with a quantized weight and a plain bf16 input (reference _gen/fallback.py:86-112: A2 of the weight, then F.linear): one to three
weight matrices per launch, gate + up + SiLU*up in one launch, and the scratch (split-K slabs, ticket words, two-pass image) the C ABI's
plan functions size."""

from __future__ import annotations

import ctypes

from typing import Any, Sequence

import torch

from fastforward_amd import _native
from fastforward_amd.ops import _base
from fastforward_amd.ops._base import _TAGS, _native_route, _ptr, _tag, _tickets, _workspace


def linear_wq(
    x: torch.Tensor,
    w_codes: torch.Tensor,
    w_scale: torch.Tensor,
    w_offset: torch.Tensor | None,
    group: int | None = None,
    bias: torch.Tensor | None = None,
    out_dtype: torch.dtype | None = None,
    pack_block: int = 0,
    two_pass: bool | None = None,
    split: int = 0,
) -> torch.Tensor | None:
    """A6, weight-only — ``F.linear(x, dequantize(w_codes))`` with the dequantization inside the GEMM's operand path
    (reference _gen/fallback.py:86-112: quantized weight, plain input).

    `x` is [..., K] bf16. `w_codes` is [N, K] int8 (codes of any bit-width <= 8) or, with ``pack_block`` > 0, the uint8
    output of :func:`pack_int4` / :func:`quantize_pack_int4` for an [N, K] weight packed with that block (two 4-bit codes per
    byte, [N * K / 2] or [N, K / 2]). `w_scale` / `w_offset` fp32 with 1 entry (per-tensor), N entries (per output channel)
    or N * K / group entries ([N, K / group] row-major: groups of `group` input channels, PerBlock(1, group, 0)).
    The weight the matrix cores see is bit for bit A2's bf16 result. ``two_pass``: None = the library's rule (from 4096
    tokens on, and from 1536 where the launch has at least 144 tiles of 256 x 256, A2 runs once per call into a scratch tensor and the
    GEMM streams that image), False = always convert inside
    the GEMM, True = offer the scratch tensor regardless of M (the library still decides). ``split``: 0 = the library's plan
    for cutting the K range of every output tile into slices when the launch has fewer tiles than the chip has CUs
    (``ffq_linear_wq_split``), >= 1 forces that many slices (tests, tuning).
    Returns None when the kernel does not cover the problem (dtypes, K % 64, group % 64): the caller dequantizes and runs a
    float GEMM as the reference does."""
    packed = pack_block > 0
    if packed:
        K = x.shape[-1]
        if w_codes.dtype != torch.uint8 or K == 0 or (w_codes.numel() * 2) % K:
            raise RuntimeError("packed weights are the uint8 output of pack_int4 for an [N, K] weight")
        N = w_codes.numel() * 2 // K
    else:
        if w_codes.dim() != 2:
            raise RuntimeError("linear_wq expects a [N, K] weight")
        N, K = w_codes.shape
    if x.shape[-1] != K:
        raise RuntimeError(f"mat1 and mat2 shapes cannot be multiplied ({x.numel() // max(x.shape[-1], 1)}x{x.shape[-1]} and {N}x{K}^T)")
    group = K if group is None else int(group)
    out_dtype = out_dtype or x.dtype
    if x.dtype not in _TAGS or w_codes.dtype not in _TAGS or out_dtype not in _TAGS:
        return None
    M = x.numel() // K if K else 0
    lib = _native.library()
    if not lib.ffq_linear_wq_supported(_tag(x.dtype), _tag(w_codes.dtype), _tag(out_dtype), M, N, K, group, int(pack_block)):
        return None
    if _native_route(x):  # dispatcher -> C++ (csrc/ffq_torch.cpp) -> C ABI
        return torch.ops.fastforward_amd.linear_wq(x, w_codes, w_scale, w_offset, group, bias, out_dtype, int(pack_block), -1 if two_pass is None else int(bool(two_pass)), int(split))
    return _linear_wq(x, w_codes, w_scale, w_offset, group, bias, out_dtype, int(pack_block), -1 if two_pass is None else int(bool(two_pass)), int(split))


def _linear_wq(x, w_codes, w_scale, w_offset, group, bias, out_dtype, pack_block, two_pass, split):  # type: ignore[no-untyped-def]
    """Python implementation of the ``linear_wq`` operator for a problem the kernel covers; arguments in schema order
    (`two_pass`: -1 = the library's rule, 0 = never, 1 = offer the image's scratch whatever M)."""
    K = x.shape[-1]
    N = w_codes.numel() * 2 // K if pack_block > 0 else w_codes.shape[0]
    M = x.numel() // K
    two_pass = None if two_pass < 0 else bool(two_pass)
    xc, wc = x.detach().contiguous(), w_codes.detach().contiguous()
    sc = w_scale.detach().reshape(-1).to(torch.float32).contiguous()
    of = None if w_offset is None else w_offset.detach().reshape(-1).to(torch.float32).contiguous()
    if of is not None and of.numel() != sc.numel():
        raise RuntimeError(f"scale has {sc.numel()} entries, offset {of.numel()}")
    bias_c = None if bias is None else bias.detach().contiguous()
    lib, stream = _base._prepare(xc, wc, sc, of, bias_c)
    out = torch.empty((*xc.shape[:-1], N), dtype=out_dtype, device=xc.device)
    nbytes, tickets = _wq_scratch(lib, M, N, K, False, two_pass, split, xc.device, stream)
    ws = _workspace(nbytes, xc.device)
    lib.check(
        lib.ffq_linear_wq(
            _ptr(xc), _tag(xc.dtype), _ptr(wc), _tag(wc.dtype), int(pack_block), _ptr(sc), _ptr(of), sc.numel(), group,
            _ptr(bias_c), _tag(bias_c.dtype) if bias_c is not None else 0, _ptr(out), _tag(out_dtype), M, N, K, _ptr(ws), nbytes,
            _ptr(tickets), int(split), stream,
        )
    )
    return out


def linear_wq_multi(
    x: torch.Tensor,
    w_codes: Sequence[torch.Tensor],
    w_scales: Sequence[torch.Tensor],
    w_offsets: Sequence[torch.Tensor | None],
    group: int | None = None,
    out_dtype: torch.dtype | None = None,
    pack_block: int = 0,
    two_pass: bool | None = None,
    split: int = 0,
) -> list[torch.Tensor] | None:
    """Two or three weight-only linears on the SAME input in one launch (q_proj / k_proj / v_proj: three ``QuantizedLinear``
    modules reading one hidden state, reference nn/linear.py:32-39): ``[linear_wq(x, w_i, ...) for i]`` as separate tensors, from
    one tile walk over all the column tiles. Operands as :func:`linear_wq`; all weights share dtype, packing, `group`, the
    granularity kind and the presence of offsets, every weight but the last has a multiple of 256 rows. None when that does
    not hold (the caller runs the linears one by one)."""
    count = len(w_codes)
    if not (2 <= count <= 3) or len(w_scales) != count or len(w_offsets) != count:
        return None
    K = x.shape[-1]
    packed = pack_block > 0
    rows = []
    for c in w_codes:
        if packed:
            if c.dtype != torch.uint8 or K == 0 or (c.numel() * 2) % K:
                return None
            rows.append(c.numel() * 2 // K)
        else:
            if c.dim() != 2 or c.shape[1] != K:
                return None
            rows.append(c.shape[0])
    group = K if group is None else int(group)
    out_dtype = out_dtype or x.dtype
    if x.dtype not in _TAGS or any(c.dtype != w_codes[0].dtype for c in w_codes) or w_codes[0].dtype not in _TAGS or out_dtype not in _TAGS:
        return None
    if any(n % 256 for n in rows[:-1]) or any((o is None) != (w_offsets[0] is None) for o in w_offsets):
        return None
    M = x.numel() // K if K else 0
    N = sum(rows)
    lib = _native.library()
    if not lib.ffq_linear_wq_supported(_tag(x.dtype), _tag(w_codes[0].dtype), _tag(out_dtype), M, N, K, group, int(pack_block)):
        return None
    flat = lambda t: None if t is None else t.detach().reshape(-1).to(torch.float32).contiguous()  # noqa: E731
    scales, offsets = [flat(t) for t in w_scales], [flat(t) for t in w_offsets]
    kinds = {int(s_.numel() != 1) for s_ in scales}
    if len(kinds) != 1:
        return None
    per_row = kinds.pop()
    for n, s_, o_ in zip(rows, scales, offsets):
        if s_.numel() != (n * (K // group) if per_row else 1) or (o_ is not None and o_.numel() != s_.numel()):
            return None
    if not per_row and group != K:
        return None
    xc = x.detach().contiguous()
    codes = [c.detach().contiguous() for c in w_codes]
    lib, stream = _base._prepare(xc, *codes, *scales, *[o for o in offsets if o is not None])
    outs = [torch.empty((*xc.shape[:-1], n), dtype=out_dtype, device=xc.device) for n in rows]
    nbytes, tickets = _wq_scratch(lib, M, N, K, False, two_pass, split, xc.device, stream)
    ws = _workspace(nbytes, xc.device)
    ptrs = lambda tensors: (ctypes.c_void_p * count)(*[_ptr(t) for t in tensors])  # noqa: E731
    lib.check(
        lib.ffq_linear_wq_multi(
            _ptr(xc), _tag(xc.dtype), count, ptrs(codes), _tag(codes[0].dtype), int(pack_block), ptrs(scales), ptrs(offsets), per_row, group,
            ptrs(outs), _tag(out_dtype), M, (ctypes.c_int64 * count)(*rows), K, _ptr(ws), nbytes, _ptr(tickets), int(split), stream,
        )
    )
    return outs


def _wq_scratch(lib: Any, M: int, N: int, K: int, mlp: bool, two_pass: bool | None, split: int, device: torch.device, stream: int) -> tuple[int, torch.Tensor | None]:
    """(workspace bytes, ticket buffer) of a weight-only GEMM launch: the split-K slabs of the plan (or of a forced `split`) at the
    front, the bf16 image(s) of the two-pass form behind them."""
    tickets = int(lib.ffq_linear_wq_tickets(M, N, K, int(mlp)))  # two per tile of the last round
    plan = int(lib.ffq_linear_wq_split(M, N, K, int(mlp)))
    use = max(1, int(split) if split > 0 else plan)
    slabs = int(lib.ffq_linear_wq_slab_bytes(M, N, K, int(mlp), use))
    if two_pass is False:
        image = 0
    elif two_pass:
        image = (2 if mlp else 1) * N * K * 2
    else:  # the library's rule: its figure minus the slabs of its own plan
        full = int(lib.ffq_mlp_gate_up_wq_workspace_bytes(M, N, K) if mlp else lib.ffq_linear_wq_workspace_bytes(M, N, K))
        image = full - int(lib.ffq_linear_wq_slab_bytes(M, N, K, int(mlp), plan))
    # (tickets whenever slabs are offered: where the preferred form declines the weight's storage, the form that takes over has a plan of its own)
    return slabs + image, (_tickets(tickets, device, stream) if (use > 1 or slabs > 0) and tickets > 0 else None)


def mlp_gate_up_wq(
    x: torch.Tensor,
    gate_codes: torch.Tensor,
    up_codes: torch.Tensor,
    gate_scale: torch.Tensor,
    gate_offset: torch.Tensor | None,
    up_scale: torch.Tensor,
    up_offset: torch.Tensor | None,
    group: int | None = None,
    pack_block: int = 0,
    two_pass: bool | None = None,
    split: int = 0,
) -> torch.Tensor | None:
    """``silu(gate_proj(x)) * up_proj(x)`` of a weight-only quantized MLP (reference quantized_llama/mlp.py:30-40 over
    _gen/fallback.py:86-112) in one launch: bit for bit ``silu_mul_quantize(linear_wq(x, gate), linear_wq(x, up), want_product=True)``
    without the two bf16 projections in HBM. Operands as :func:`linear_wq` (both weights in the same form); bf16 only.
    None when the kernel does not cover the problem."""
    K = x.shape[-1]
    if pack_block > 0:
        if gate_codes.dtype != torch.uint8 or K == 0 or (gate_codes.numel() * 2) % K:
            raise RuntimeError("packed weights are the uint8 output of pack_int4 for an [N, K] weight")
        N = gate_codes.numel() * 2 // K
    else:
        if gate_codes.dim() != 2 or gate_codes.shape[1] != K:
            raise RuntimeError("mlp_gate_up_wq expects [N, K] weights")
        N = gate_codes.shape[0]
    if gate_codes.shape != up_codes.shape or gate_codes.dtype != up_codes.dtype:
        raise RuntimeError("gate and up weights differ in shape or dtype")
    group = K if group is None else int(group)
    if x.dtype != torch.bfloat16 or gate_codes.dtype not in _TAGS or N % 128 or (gate_offset is None) != (up_offset is None):
        return None
    M = x.numel() // K if K else 0
    lib = _native.library()
    if not lib.ffq_linear_wq_supported(_tag(x.dtype), _tag(gate_codes.dtype), _tag(torch.bfloat16), M, N, K, group, int(pack_block)):
        return None
    xc, gc, uc = x.detach().contiguous(), gate_codes.detach().contiguous(), up_codes.detach().contiguous()
    flat = lambda t: None if t is None else t.detach().reshape(-1).to(torch.float32).contiguous()  # noqa: E731
    gs, go, us, uo = flat(gate_scale), flat(gate_offset), flat(up_scale), flat(up_offset)
    if gs.numel() != us.numel() or (go is not None and (go.numel() != gs.numel() or uo.numel() != gs.numel())):
        raise RuntimeError("gate and up parameters differ in count")
    lib, stream = _base._prepare(xc, gc, uc, gs, go, us, uo)
    out = torch.empty((*xc.shape[:-1], N), dtype=torch.bfloat16, device=xc.device)
    nbytes, tickets = _wq_scratch(lib, M, N, K, True, two_pass, split, xc.device, stream)
    ws = _workspace(nbytes, xc.device)
    lib.check(
        lib.ffq_mlp_gate_up_wq(
            _ptr(xc), _tag(xc.dtype), _ptr(gc), _ptr(uc), _tag(gc.dtype), int(pack_block), _ptr(gs), _ptr(go), _ptr(us), _ptr(uo),
            gs.numel(), group, _ptr(out), M, N, K, _ptr(ws), nbytes, _ptr(tickets), int(split), stream,
        )
    )
    return out
