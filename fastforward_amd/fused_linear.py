"""The quantized linear — and mm / matmul / bmm — registered in the quantized-operator dispatcher (plug-in seam #2).

The reference registers no kernel for ``"linear"``; every quantized linear therefore runs
``fallback.linear`` (src/fastforward/_gen/fallback.py:77-112): dequantize x, dequantize w, float
GEMM, output quantizer. This module registers kernels whose predicates accept the combinations the
hand-written GEMMs cover and return False for everything else, so the reference behaviour (the
float fallback in :mod:`fastforward_amd.nn.functional`) still applies there:

* W8A8 (both operands static-affine codes, <= 8 bits, weight per tensor / per output channel): the int8-MFMA GEMM
  A6 (``ops.linear_w8a8``) with the affine parameters in its epilogue;
* weight-only (plain bf16 input, quantized weight: BASELINE configs 2 and 4) and quantized inputs against GROUPED
  weights (``PerBlock(in, G)``: W4-g128 x A8): the bf16-MFMA GEMM with A2 of the weight codes in its operand path
  (``ops.linear_wq``); a quantized input is dequantized first (A2), exactly the reference's operands;
* a static per-tensor ``LinearQuantizer`` as ``output_quantizer`` (fallback.py:110-111) runs INSIDE the int8 GEMM's
  epilogue (``ops.linear_w8a8(out_scale=...)``): one launch instead of two, the real-valued output never visits HBM.

ONE implementation serves two registrations: this package's own dispatcher (below) and, through
``fastforward_amd.adapter.install()``, the reference's — :class:`DispatcherKernels` is written against a
:class:`Surface` (the QuantizedTensor / quantizer / error types of either package), so both seams apply the same
predicates and launch the same kernels.

The kernels accept both calling conventions the dispatcher produces:
``kernel(input, weight, bias)`` from ``QuantizedTensor.__torch_function__`` when user code calls
``torch.nn.functional.linear`` on quantized tensors, and
``kernel(input=..., weight=..., bias=..., output_quantizer=..., strict_quantization=...)`` from
``fastforward_amd.nn.functional.linear``.
"""

from __future__ import annotations

import contextlib
import dataclasses
import math
import weakref

from typing import Any, Callable

import torch

from fastforward_amd import _native, ops
from fastforward_amd.dispatcher import Predicate, register


@dataclasses.dataclass(frozen=True)
class Surface:
    """The Python types a set of dispatcher kernels is written against."""

    quantized_tensor: type
    affine_function: type
    static_params: type
    linear_quantizer: type
    quantization_context: type
    error: type
    export_mode: Callable[[], bool]


def own_surface() -> Surface:
    from fastforward_amd import flags
    from fastforward_amd.exceptions import QuantizationError
    from fastforward_amd.nn.linear_quantizer import LinearQuantizer
    from fastforward_amd.quantization.affine import AffineQuantizationFunction, StaticAffineQuantParams
    from fastforward_amd.quantization.function import QuantizationContext
    from fastforward_amd.quantized_tensor import QuantizedTensor

    return Surface(QuantizedTensor, AffineQuantizationFunction, StaticAffineQuantParams, LinearQuantizer, QuantizationContext,
                   QuantizationError, flags.get_export_mode)


def _on_backend(*tensors: Any) -> bool:
    """All operands on a HIP device and the backend library loadable — anything else takes the float fallback.
    (``oracle/inject.py`` swaps this check for a host one while the oracle stands in as the library: tests only.)"""
    return all(t.is_cuda for t in tensors) and _native.is_available()


# Which GEMM a weight-only linear takes: the hand-written bf16 x weight-code kernel at EVERY token count (round 4: a launch with
# fewer output tiles than the chip has CUs cuts the K range of its last round's tiles into slices — csrc/ffq_wlinear.hip — so the
# round-3 token threshold, below which the reference's A2 + vendor GEMM ran, is gone), or, inside ``with weight_only_kernel(False)``,
# the reference's own path (fallback.py:86-112: A2 into a bf16 tensor + F.linear — the A/B arm of tools/bench_configs.py).
# Both see the same operands bit for bit.
_WEIGHT_ONLY_KERNEL = True


@contextlib.contextmanager
def weight_only_kernel(enabled: bool = True):
    """Route weight-only quantized linears (plain bf16 input, quantized weight) to the hand-written GEMM (default) or, with
    ``enabled=False``, leave them to the float fallback (A2 + ``F.linear``) inside the block."""
    global _WEIGHT_ONLY_KERNEL
    previous = _WEIGHT_ONLY_KERNEL
    _WEIGHT_ONLY_KERNEL = bool(enabled)
    try:
        yield
    finally:
        _WEIGHT_ONLY_KERNEL = previous


_FLOATS = (torch.bfloat16, torch.float16, torch.float32)


def _module_hooked(module: Any) -> bool:
    """A forward / backward hook on `module`, or a global module hook, would run if the module were called."""
    from torch.nn.modules import module as nn_module

    if any(getattr(nn_module, name, None) for name in ("_global_forward_hooks", "_global_forward_pre_hooks", "_global_backward_hooks", "_global_backward_pre_hooks")):
        return True
    return bool(getattr(module, "_forward_hooks", None) or getattr(module, "_forward_pre_hooks", None) or getattr(module, "_backward_hooks", None)
                or getattr(module, "_backward_pre_hooks", None))

# A symmetric quantizer carries an offset BUFFER that is all zeros unless its data is one-sided (reference
# nn/linear_quantizer.py:164-170). The int8 GEMM recognises that on the device (a one-block check ahead of the launch and a
# slightly heavier epilogue: no host read, which is what a range estimator that rewrites the parameters on every step needs);
# for parameters that have STOPPED changing the answer is read once on the host and such offsets are not passed at all.
# Keyed on (tensor object, version): the range setter bumps the version; never read while a hipGraph is being captured, and
# not at the first sighting of a version (calibration: every step is a first sighting).
# Only BUFFERS are remembered — the derived offset of a symmetric quantizer, which nothing but the range setter writes
# (nn/linear_quantizer.py:_write_parameters_for_range: version bumped). A learnable offset (an ``nn.Parameter``: asymmetric
# quantizers) is never remembered: user code and optimizers write parameters, also through ``.data``, which no version counter
# sees — those always take the device-side decision. ``forget_zero_offsets()`` drops everything remembered (for code that
# writes a quantizer's buffers behind its back).
_ZERO_OFFSETS: dict[int, tuple[Any, int, bool | None]] = {}


def forget_zero_offsets() -> None:
    _ZERO_OFFSETS.clear()


def known_zero_offset(offset: Any) -> bool:
    """True when `offset` is known (from an earlier call with the same version) to round to all zeros."""
    if not isinstance(offset, torch.Tensor) or isinstance(offset, torch.nn.Parameter):
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
        _ZERO_OFFSETS[id(offset)] = (weakref.ref(offset), offset._version, None)
        return False
    zero = not bool(torch.round(offset.detach()).any())  # the version was stable across two calls: one read
    _ZERO_OFFSETS[id(offset)] = (weakref.ref(offset), offset._version, zero)
    return zero


class DispatcherKernels:
    """Predicates and kernels of ``linear`` / ``mm`` / ``matmul`` / ``bmm`` against one :class:`Surface`."""

    def __init__(self, surface: Callable[[], Surface]) -> None:
        self._make_surface = surface
        self._surface: Surface | None = None

    @property
    def surface(self) -> Surface:
        if self._surface is None:
            self._surface = self._make_surface()
        return self._surface

    # ---- what a tensor is ------------------------------------------------------------------------------------------
    def static_affine(self, t: Any) -> bool:
        s = self.surface
        if not isinstance(t, s.quantized_tensor):
            return False
        ctx = t.quantization_context
        return issubclass(ctx.quantization_fn, s.affine_function) and isinstance(ctx.quantization_params, s.static_params)

    @staticmethod
    def _params(t: Any) -> Any:
        return t.quantization_context.quantization_params

    @classmethod
    def _tile(cls, t: Any) -> tuple[int, ...]:
        tile = cls._params(t).granularity.tile_size(t.shape)
        return tuple(t.shape) if isinstance(tile, str) else tuple(tile)

    @classmethod
    def _bits_ok(cls, *tensors: Any) -> bool:
        return all(cls._params(t).num_bits <= 8 and cls._params(t).num_bits == int(cls._params(t).num_bits) for t in tensors)

    @classmethod
    def row_mode(cls, t: Any) -> str | None:
        """'tensor' (one parameter pair), 'row' (one pair per row of the last-dim-contiguous matrix), or None."""
        tile = cls._tile(t)
        if tile == tuple(t.shape):
            return "tensor"
        if t.dim() >= 1 and all(v == 1 for v in tile[:-1]) and tile[-1] == t.shape[-1]:
            return "row"
        return None

    @classmethod
    def col_mode(cls, t: Any) -> str | None:
        """'tensor' or 'col' (one parameter pair per column of a [K, N] matrix) for the right operand of a matmul."""
        tile = cls._tile(t)
        if tile == tuple(t.shape):
            return "tensor"
        if t.dim() == 2 and tile[0] == t.shape[0] and tile[1] == 1:
            return "col"
        return None

    @classmethod
    def weight_group(cls, weight: Any) -> int | None:
        """Input channels sharing one parameter pair within a row ([N, K / group] parameters), or None if not such a tiling."""
        tile = cls._tile(weight)
        n, k = weight.shape
        if tile == (n, k):
            return k  # per tensor: one pair
        if len(tile) == 2 and tile[0] == 1 and k % tile[1] == 0:
            return int(tile[1])  # (1, K): per output channel; (1, G): groups along the input channels
        return None

    @staticmethod
    def _int8_codes(t: Any) -> torch.Tensor:
        """Codes as int8. Float / wider containers hold the same integers; they are converted exactly
        by A1 with scale 1 (x / 1 - 0, round, clamp to the int8 range)."""
        raw = t.raw_data
        if raw.dtype == torch.int8:
            return raw
        one = torch.ones(1, dtype=torch.float32, device=raw.device)
        return ops.quantize_by_tile(raw, one, raw.shape, 8, torch.int8)

    @classmethod
    def _scale_offset(cls, t: Any) -> tuple[torch.Tensor, torch.Tensor | None]:
        p = cls._params(t)
        offset = None if p.offset is None or known_zero_offset(p.offset) else torch.as_tensor(p.offset, device=t.device)
        return torch.as_tensor(p.scale, device=t.device), offset

    @classmethod
    def _deq_dtype(cls, t: Any) -> torch.dtype:
        return cls._params(t).dequantize_dtype or torch.get_default_dtype()

    # ---- the output quantizer inside the GEMM's epilogue ----------------------------------------------------------------
    def _requant(self, output_quantizer: Any, deq: torch.dtype) -> dict[str, Any] | None:
        """Arguments of ``ops.linear_w8a8`` that run `output_quantizer` in the GEMM's epilogue, or None when it has to run
        as its own pass: anything but a plain, initialised, per-tensor ``LinearQuantizer`` with fp32 parameters and no active
        override (range estimation, disable_quantization ... install overrides), export mode, or a caller that may want
        gradients (the fused launch has no autograd formula; the quantizer's own forward does)."""
        s = self.surface
        q = output_quantizer
        if q is None or type(q) is not s.linear_quantizer or s.export_mode():
            return None
        if _module_hooked(q):  # the fused launch never calls q: a forward hook on it (code recorders, observers) would not fire
            return None
        if q.has_uninitialized_params or next(iter(q.overrides), None) is not None or not q.per_tensor:
            return None
        scale, offset = q.scale, q.offset
        if scale.numel() != 1 or scale.dtype != torch.float32 or (offset is not None and (offset.numel() != 1 or offset.dtype != torch.float32)):
            return None
        if torch.is_grad_enabled() and (scale.requires_grad or (offset is not None and offset.requires_grad)):
            return None
        container = q.quantized_dtype or deq
        if container not in (torch.int8, *_FLOATS) or q.num_bits != int(q.num_bits):
            return None
        return dict(out_dtype=container, out_scale=scale, out_offset=offset, out_num_bits=q.num_bits, requant_from=deq)

    def _wrap(self, like: Any, codes: torch.Tensor, output_quantizer: Any, deq: torch.dtype) -> Any:
        """The QuantizedTensor ``output_quantizer(real_valued_output)`` would have returned, around the fused launch's codes."""
        s = self.surface
        params = output_quantizer.quantization_parameters().with_changes(dequantize_dtype=deq)
        return s.quantized_tensor(codes, s.quantization_context(output_quantizer.quantization_function, params))

    def _finish(self, out: torch.Tensor, fused: dict[str, Any] | None, like: Any, output_quantizer: Any, deq: torch.dtype) -> Any:
        if fused is not None:
            return self._wrap(like, out, output_quantizer, deq)
        return output_quantizer(out) if output_quantizer is not None else out

    # ---- linear ---------------------------------------------------------------------------------------------------------
    def _wq_covers(self, x_dtype: torch.dtype, weight: Any, tokens: int) -> int | None:
        """The group size when the bf16 x weight-code GEMM takes (activation dtype, weight tiling, K): coverage is the library's
        own rule (``ffq_linear_wq_supported``, asked for int8 codes); every token count >= 1 is covered."""
        group = self.weight_group(weight)
        if group is None or x_dtype not in ops._TAGS:
            return None
        n, k = weight.shape
        lib = _native.library()
        ok = lib.ffq_linear_wq_supported(ops._tag(x_dtype), ops._tag(torch.int8), ops._tag(x_dtype), max(int(tokens), 1), n, k, group, 0)
        return group if ok else None

    def supported_linear(self, input: Any = None, weight: Any = None, bias: Any = None, **_: Any) -> bool:
        if not (self.static_affine(input) and self.static_affine(weight)):
            return False
        if not _on_backend(input, weight) or weight.dim() != 2 or input.dim() < 1:
            return False
        if input.shape[-1] != weight.shape[1] or input.numel() == 0 or not self._bits_ok(input, weight):
            return False
        deq = self._deq_dtype(input)
        if deq not in _FLOATS or (self._params(weight).dequantize_dtype or deq) != deq:
            return False  # the float fallback would raise on mixed dtypes; let it
        if isinstance(bias, self.surface.quantized_tensor) and not self.static_affine(bias):
            return False
        if self.row_mode(weight) is not None:  # the int8 GEMM
            return weight.shape[1] % 16 == 0 and self.row_mode(input) is not None
        # grouped weights (PerBlock(in, G): W4-g128 x A8): the bf16 GEMM on the dequantized input
        return _WEIGHT_ONLY_KERNEL and self._wq_covers(deq, weight, input.numel() // input.shape[-1]) is not None

    def linear(self, input: Any, weight: Any, bias: Any = None, *, output_quantizer: Any = None, strict_quantization: bool | None = None) -> Any:
        s = self.surface
        if strict_quantization and output_quantizer is None:
            raise s.error("'output_quantizer' must be provided if strict_quantization=True")
        deq = self._deq_dtype(input)
        if isinstance(bias, s.quantized_tensor):
            bias = bias.dequantize()
        (xs, xo), (ws, wo) = self._scale_offset(input), self._scale_offset(weight)
        if self.row_mode(weight) is None:
            if getattr(input, "_ffq_earlier", None) is not None:  # (see below: this branch reads the codes as they are)
                from fastforward_amd.quantization.affine._memo import RECENT

                RECENT.settle(input)
            # group-wise weight parameters cannot leave the contraction: A2 of the input (the reference's own first step,
            # fallback.py:94-100), then the GEMM that dequantizes the weight codes on their way into the matrix cores
            out = ops.linear_wq(input.dequantize(), self._int8_codes(weight), ws, wo, group=self.weight_group(weight), bias=bias, out_dtype=deq)
            if out is None:
                out = torch.nn.functional.linear(input.dequantize(), weight.dequantize(), bias)
            return output_quantizer(out) if output_quantizer is not None else out
        fused = self._requant(output_quantizer, deq)
        if getattr(input, "_ffq_earlier", None) is not None:
            # the input quantizer left it to the device whether its A1 ran (``sibling_quantizers(undecided=True)``): the launch that
            # takes the earlier sibling's codes along, else the codes in force written first
            from fastforward_amd.quantization.affine._memo import RECENT

            earlier = RECENT.earlier_of(input)
            if fused is None and bias is None and earlier is not None:
                out = ops.linear_w8a8_earlier(self._int8_codes(input), earlier, self._int8_codes(weight), xs, xo, ws, wo, out_dtype=deq)
                if out is not None:
                    return self._finish(out, None, input, output_quantizer, deq)
            RECENT.settle(input)
        out = ops.linear_w8a8(self._int8_codes(input), self._int8_codes(weight), xs, xo, ws, wo, bias=bias, **(fused or dict(out_dtype=deq)))
        return self._finish(out, fused, input, output_quantizer, deq)

    linear.reads_undecided_codes = True  # type: ignore[attr-defined]  # (nn/functional.py: nothing to settle before this kernel)

    def supported_weight_only(self, input: Any = None, weight: Any = None, bias: Any = None, **_: Any) -> bool:
        s = self.surface
        if not _WEIGHT_ONLY_KERNEL or isinstance(input, s.quantized_tensor) or not isinstance(input, torch.Tensor) or not self.static_affine(weight):
            return False
        if not _on_backend(input, weight) or weight.dim() != 2 or input.dim() < 1 or input.numel() == 0:
            return False
        if input.shape[-1] != weight.shape[1] or not self._bits_ok(weight) or (self._params(weight).dequantize_dtype or input.dtype) != input.dtype:
            return False
        if isinstance(bias, s.quantized_tensor) and not self.static_affine(bias):
            return False
        return self._wq_covers(input.dtype, weight, input.numel() // input.shape[-1]) is not None

    def weight_only_linear(self, input: torch.Tensor, weight: Any, bias: Any = None, *, output_quantizer: Any = None, strict_quantization: bool | None = None) -> Any:
        s = self.surface
        if strict_quantization:  # the reference's messages, fallback.py:83-92
            if output_quantizer is None:
                raise s.error("'output_quantizer' must be provided if strict_quantization=True")
            raise s.error("Expected 'input' to be an instance of 'QuantizedTensor' because strict_quantization=True.")
        if isinstance(bias, s.quantized_tensor):
            bias = bias.dequantize()
        ws, wo = self._scale_offset(weight)
        out = ops.linear_wq(input, self._int8_codes(weight), ws, wo, group=self.weight_group(weight), bias=bias, out_dtype=input.dtype)
        if out is None:  # a shape the kernel does not cover after all: the reference's path
            out = torch.nn.functional.linear(input, weight.dequantize(), bias)
        return output_quantizer(out) if output_quantizer is not None else out

    # ---- mm / matmul / bmm: the same fallback pattern in the reference (_gen/fallback.py:699-798: dequantize both operands,
    # float matmul, output quantizer), the same int8 contraction here. The right operand arrives as [K, N]; the GEMM contracts
    # K-contiguous rows, so its codes are transposed once (1 B/elem; free when the operand is itself a transposed view of a
    # K-contiguous tensor, e.g. ``k.transpose(-1, -2)``). Per-tensor or per-COLUMN parameters on the right operand
    # (PerChannel(-1): one pair per output column), per-tensor or per-row on the left; everything else takes the float fallback.
    def _pair_ok(self, a: Any, b: Any) -> bool:
        if not (self.static_affine(a) and self.static_affine(b)) or not _on_backend(a, b) or not self._bits_ok(a, b):
            return False
        deq = self._deq_dtype(a)
        return deq in _FLOATS and (self._params(b).dequantize_dtype or deq) == deq

    def supported_mm(self, input: Any = None, other: Any = None, mat2: Any = None, **_: Any) -> bool:
        right = other if other is not None else mat2
        if not self._pair_ok(input, right):
            return False
        if right.dim() > 2:  # matmul with an N-d right operand: the batched launch when the leading dims agree (no broadcasting)
            return (input.dim() == right.dim() and tuple(input.shape[:-2]) == tuple(right.shape[:-2]) and input.shape[-1] == right.shape[-2]
                    and right.shape[-2] % 16 == 0 and input.numel() > 0 and right.numel() > 0 and math.prod(input.shape[:-2]) <= 65535
                    and self.row_mode(input) == "tensor" and self.row_mode(right) == "tensor")
        if right.dim() != 2 or input.dim() < 1 or input.shape[-1] != right.shape[0] or right.shape[0] % 16 or input.numel() == 0 or right.numel() == 0:
            return False
        return self.row_mode(input) is not None and self.col_mode(right) is not None

    def mm(self, input: Any, other: Any = None, *, mat2: Any = None, output_quantizer: Any = None, strict_quantization: bool | None = None) -> Any:
        right = other if other is not None else mat2
        if strict_quantization and output_quantizer is None:
            raise self.surface.error("'output_quantizer' must be provided if strict_quantization=True")
        deq = self._deq_dtype(input)
        (xs, xo), (ws, wo) = self._scale_offset(input), self._scale_offset(right)
        if right.dim() > 2:  # [..., M, K] x [..., K, N]: flatten the leading dims into one batch
            lead = tuple(input.shape[:-2])
            x_codes = self._int8_codes(input).reshape(-1, input.shape[-2], input.shape[-1])
            w_codes = self._int8_codes(right).reshape(-1, right.shape[-2], right.shape[-1]).transpose(1, 2).contiguous()
            fused = self._requant(output_quantizer, deq)
            out = ops.bmm_w8a8(x_codes, w_codes, xs, xo, ws, wo, **(fused or dict(out_dtype=deq)))
            return self._finish(out.reshape(*lead, input.shape[-2], right.shape[-1]), fused, input, output_quantizer, deq)
        w_codes = self._int8_codes(right).t().contiguous()  # [N, K]
        fused = self._requant(output_quantizer, deq)
        out = ops.linear_w8a8(self._int8_codes(input), w_codes, xs, xo, ws, wo, bias=None, **(fused or dict(out_dtype=deq)))
        return self._finish(out, fused, input, output_quantizer, deq)

    def supported_bmm(self, input: Any = None, mat2: Any = None, **_: Any) -> bool:
        if not self._pair_ok(input, mat2):
            return False
        if input.dim() != 3 or mat2.dim() != 3 or input.shape[0] != mat2.shape[0] or input.shape[2] != mat2.shape[1]:
            return False
        if input.shape[2] % 16 or input.numel() == 0 or mat2.numel() == 0 or input.shape[0] > 65535:
            return False
        # one parameter pair for each operand: the batch shares it, every matrix of the batch is one GEMM
        return self.row_mode(input) == "tensor" and self.row_mode(mat2) == "tensor"

    def bmm(self, input: Any, mat2: Any, *, output_quantizer: Any = None, strict_quantization: bool | None = None) -> Any:
        if strict_quantization and output_quantizer is None:
            raise self.surface.error("'output_quantizer' must be provided if strict_quantization=True")
        deq = self._deq_dtype(input)
        (xs, xo), (ws, wo) = self._scale_offset(input), self._scale_offset(mat2)
        x_codes, w_codes = self._int8_codes(input), self._int8_codes(mat2).transpose(1, 2).contiguous()  # [B, N, K]
        fused = self._requant(output_quantizer, deq)
        out = ops.bmm_w8a8(x_codes, w_codes, xs, xo, ws, wo, **(fused or dict(out_dtype=deq)))  # ONE launch for the whole batch
        return self._finish(out, fused, input, output_quantizer, deq)

    def register_all(self, register_fn: Callable[[str, Any, Any], Any], predicate_type: type) -> dict[str, Any]:
        """Register every kernel through `register_fn(op_name, predicate, kernel)`; returns the registration hooks by name.
        Order matters for ``linear``: the dispatcher tries the NEWEST registration of a priority first, and the two
        predicates are disjoint (quantized vs plain input), so either order dispatches the same."""
        return {
            "linear": register_fn("linear", predicate_type(self.supported_linear), self.linear),
            "linear(weight-only)": register_fn("linear", predicate_type(self.supported_weight_only), self.weight_only_linear),
            "mm": register_fn("mm", predicate_type(self.supported_mm), self.mm),
            "matmul": register_fn("matmul", predicate_type(self.supported_mm), self.mm),
            "bmm": register_fn("bmm", predicate_type(self.supported_bmm), self.bmm),
        }


# ---- this package's own dispatcher -------------------------------------------------------------------------------------------
KERNELS = DispatcherKernels(own_surface)
fused_linear = KERNELS.linear
fused_linear_weight_only = KERNELS.weight_only_linear
fused_mm = KERNELS.mm
fused_bmm = KERNELS.bmm
fused_linear_predicate = Predicate(KERNELS.supported_linear)
fused_linear_weight_only_predicate = Predicate(KERNELS.supported_weight_only)
fused_mm_predicate = Predicate(KERNELS.supported_mm)
fused_bmm_predicate = Predicate(KERNELS.supported_bmm)
_registration = register("linear", fused_linear_predicate, fused_linear)
_registration_weight_only = register("linear", fused_linear_weight_only_predicate, fused_linear_weight_only)
_registration_mm = register("mm", fused_mm_predicate, fused_mm)
_registration_matmul = register("matmul", fused_mm_predicate, fused_mm)
_registration_bmm = register("bmm", fused_bmm_predicate, fused_bmm)
