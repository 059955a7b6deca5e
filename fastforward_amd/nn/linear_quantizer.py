"""Static affine quantizer module (reference: src/fastforward/nn/linear_quantizer.py).

``LinearQuantizer`` owns lazily-shaped ``scale`` / ``offset`` parameters (:147-173) that are
materialised by the first ``quantization_range = (min, max)`` (:327-357) and quantizes through
``AffineQuantizationFunction`` (:107-117, :235-249), i.e. through the HIP kernel A1.

Difference to the reference that matters for speed, not for results: the range setter computes
(scale, offset) with the device kernel A5 directly into the parameters, so setting a range never
synchronises the host with the GPU (the reference does ``min_range.min() >= 0`` on the host,
quantization/affine/range.py:100).
"""

from __future__ import annotations

import abc

from typing import Any, Callable

import torch

from fastforward_amd import ops
from fastforward_amd.common import ensure_tensor
from fastforward_amd.nn.quantizer import Quantizer
from fastforward_amd.quantization import affine as affine_quant
from fastforward_amd.quantization.affine._memo import RECENT
from fastforward_amd.quantization import granularity as granularities
from fastforward_amd.quantization.function import QuantizationContext, QuantizationFunction
from fastforward_amd.quantized_tensor import QuantizedTensor


class AbstractAffineQuantizer(Quantizer, abc.ABC):
    """Common state of affine quantizers: bit-width, granularity, storage dtype (reference :25-117)."""

    def __init__(
        self,
        num_bits: int,
        *,
        granularity: granularities.Granularity | None = None,
        quantized_dtype: torch.dtype | None = None,
    ) -> None:
        super().__init__()
        self.num_bits = num_bits
        self.granularity = granularity or granularities.PerTensor()
        self.quantized_dtype = quantized_dtype

    @property
    def per_channel(self) -> bool:
        return granularities.is_per_channel(self.granularity)

    @property
    def per_tensor(self) -> bool:
        return granularities.is_per_tensor(self.granularity)

    @property
    def integer_minimum(self) -> float:
        return affine_quant.integer_minimum(self.num_bits)

    @property
    def integer_maximum(self) -> float:
        return affine_quant.integer_maximum(self.num_bits)

    @property
    def has_uninitialized_params(self) -> bool:
        lazy = torch.nn.parameter.UninitializedParameter
        return any(isinstance(p, lazy) for p in self.parameters())

    def extra_repr(self) -> str:
        own = f"num_bits={self.num_bits}, granularity={self.granularity}"
        inherited = super().extra_repr()
        return f"{inherited}, {own}" if inherited else own

    @property
    @abc.abstractmethod
    def quantization_function(self) -> type[QuantizationFunction[Any]]: ...

    @abc.abstractmethod
    def quantization_parameters(self) -> Any: ...

    def quantization_context(self) -> QuantizationContext[Any]:
        return QuantizationContext(self.quantization_function, self.quantization_parameters())

    def quantize(self, data: torch.Tensor) -> torch.Tensor:
        return self.quantization_function.quantize(data, self.quantization_parameters())


class LinearQuantizer(AbstractAffineQuantizer):
    """Affine quantizer with per-tensor / per-channel / per-block / per-tile parameters.

    Args mirror the reference (:120-146): ``symmetric`` (default True), ``allow_one_sided``
    (default True: a symmetric quantizer whose range minimum is >= 0 everywhere switches to the
    unsigned grid by means of an offset buffer), ``granularity``, ``quantized_dtype`` (container of
    the codes; default: the data dtype), ``param_dtype`` (default fp32) and ``device``.
    """

    def __init__(
        self,
        num_bits: int,
        *,
        symmetric: bool = True,
        allow_one_sided: bool = True,
        granularity: granularities.Granularity | None = None,
        quantized_dtype: torch.dtype | None = None,
        param_dtype: torch.dtype | None = None,
        device: torch.device | str = "cpu",
    ) -> None:
        super().__init__(num_bits=num_bits, granularity=granularity, quantized_dtype=quantized_dtype)
        self.allow_one_sided = allow_one_sided
        self.scale = torch.nn.UninitializedParameter(device=device, dtype=param_dtype)
        if symmetric and not allow_one_sided:
            self.register_parameter("offset", None)
        elif symmetric:
            # a buffer, not a Parameter: only ever 0 or -int_min                 (reference :164-170)
            self.register_buffer("offset", torch.nn.UninitializedBuffer(device=device, dtype=param_dtype))
        else:
            self.offset = torch.nn.UninitializedParameter(device=device, dtype=param_dtype)

    @property
    def symmetric(self) -> bool:
        """True when there is no learnable offset (offset is absent or a buffer; reference :175-191)."""
        return "offset" in self._buffers or self.offset is None

    def reset_parameters(self) -> None:
        with torch.no_grad():
            self.scale = torch.nn.UninitializedParameter(device=self.scale.device, dtype=self.scale.dtype)
            if self.offset is None:
                return
            kind = torch.nn.UninitializedParameter if isinstance(self.offset, torch.nn.Parameter) else torch.nn.UninitializedBuffer
            self.offset = kind(device=self.offset.device, dtype=self.offset.dtype)

    def _initialize_parameters(self, parameter_dimensionality: int) -> None:
        if not self.has_uninitialized_params:
            return
        shape = torch.Size([parameter_dimensionality])
        with torch.no_grad():
            self.scale.materialize(shape)
            self.scale.fill_(1.0)
            if self.offset is not None:
                self.offset.materialize(shape)
                self.offset.fill_(0.0)

    def extra_repr(self) -> str:
        own = f"symmetric={self.symmetric}"
        inherited = super().extra_repr()
        return f"{inherited}, {own}" if inherited else own

    def quantization_parameters(self) -> affine_quant.StaticAffineQuantParams:
        return affine_quant.StaticAffineQuantParams(
            scale=self.scale,
            offset=self.offset,
            granularity=self.granularity,
            num_bits=self.num_bits,
            quantized_dtype=self.quantized_dtype,
        )

    @property
    def quantization_function(self) -> type[QuantizationFunction[Any]]:
        return affine_quant.AffineQuantizationFunction

    def quantize(self, data: torch.Tensor) -> torch.Tensor:
        try:
            return super().quantize(data)
        except ValueError as e:
            if not self.has_uninitialized_params:
                raise
            name = type(self).__name__
            raise ValueError(
                f"Tried to quantize a tensor using an uninitialized quantizer (of type {name}). This "
                "quantizer is initialized after its quantization_range is specified. This can be done "
                f"explicitly by using the {name}.quantization_range setter or using a range setting method."
            ) from e

    def operator_for_range(
        self, min_range: torch.Tensor, max_range: torch.Tensor, data_shape: torch.Size
    ) -> Callable[[torch.Tensor], QuantizedTensor]:
        """A quantization operator for an explicit range, independent of self (reference :280-299)."""
        scale, offset = self._parameters_for_range(min_range, max_range)
        context = affine_quant.quantization_context(
            scale=scale, offset=offset, num_bits=self.num_bits, granularity=self.granularity, output_dtype=self.quantized_dtype
        )
        return lambda data: context.quantization_fn.quantize(data, context.quantization_params)

    def _parameters_for_range(self, min_range: torch.Tensor, max_range: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor | None]:
        return affine_quant.parameters_for_range(
            min_range=min_range,
            max_range=max_range,
            num_bits=self.num_bits,
            symmetric=self.symmetric,
            allow_one_sided=self.allow_one_sided,
        )

    @property
    def quantization_range(self) -> tuple[torch.Tensor | float | None, torch.Tensor | float | None]:
        if self.has_uninitialized_params:
            return None, None
        return affine_quant.quantization_range(self.scale, self.offset, self.num_bits)

    @quantization_range.setter
    def quantization_range(self, quant_range: tuple[torch.Tensor | float, torch.Tensor | float]) -> None:
        try:
            lo, hi = (ensure_tensor(t, device=self.scale.device) for t in quant_range)
        except ValueError as e:
            raise ValueError(f"Tried to set quantization range with {len(quant_range)}-tuple. A 2-tuple is expected") from e
        except TypeError as e:
            raise ValueError("Tried to set quantization range with a single value. A 2-tuple is expected") from e
        if self.has_uninitialized_params:
            self._initialize_parameters(lo.numel())
        self._write_parameters_for_range(lo, hi)

    def update_range_from_data(self, data: torch.Tensor, tile: Any, running_min: torch.Tensor, running_max: torch.Tensor,
                               status: torch.Tensor | None) -> bool:
        """A RunningMinMax estimator step on `data` in ONE backend call: merge its per-tile extrema into `running_min` /
        `running_max` (in place) and set this quantizer's range from the merged values — exactly
        ``quantization_range = (running_min, running_max)`` after the merge (reference minmax.py:236-239). False when the
        parameters cannot be written in place (other device, layout or count): the caller then takes the two steps."""
        with torch.no_grad():
            if not data.is_cuda:  # the fused entry exists for the device path; host tensors (oracle tests) take the two steps
                return False
            # a subclass that overrides the range setter or the parameter write (to clamp, log or keep derived state) must see
            # every range: the fused entry writes scale / offset through raw pointers, so it is taken only when both are this class's own
            cls = type(self)
            if cls.quantization_range is not LinearQuantizer.quantization_range or cls._write_parameters_for_range is not LinearQuantizer._write_parameters_for_range:
                return False
            if self.has_uninitialized_params:
                self._initialize_parameters(running_min.numel())
            direct = (
                self.scale.device == data.device and self.scale.numel() == running_min.numel() and self.scale.is_contiguous()
                and (self.offset is None or (self.offset.device == data.device and self.offset.is_contiguous() and self.offset.numel() == running_min.numel()))
            )
            if not direct:
                return False
            source, source_tile = data, tile
            if RECENT.inside_scope and running_min.numel() == 1 and type(data) is torch.Tensor and tuple(tile) == tuple(data.shape):
                # sibling estimators (q / k / v, gate / up: the same activation): ONE reduction over the tensor, every
                # estimator merges its two numbers — min / max of [min(x), max(x)] are min(x) and max(x), NaN and Inf included
                pair = RECENT.extrema(data)
                if pair is None:
                    pair = torch.empty(2, dtype=data.dtype, device=data.device)
                    ops.minmax_by_tile(data, tile, into=(pair[0:1], pair[1:2]))
                    RECENT.remember_extrema(data, pair)
                source, source_tile = pair, (2,)
            ops.running_minmax_step(source, source_tile, running_min, running_max, status, self.num_bits, self.symmetric, self.allow_one_sided,
                                    self.scale.data, None if self.offset is None else self.offset.data)
            torch.autograd.graph.increment_version(self.scale)  # written through raw pointers: tell the version counters
            if self.offset is not None:
                torch.autograd.graph.increment_version(self.offset)
            return True

    def is_plain(self) -> bool:
        """No subclass has replaced a piece of the quantize / range-setting path: the fused entries, which do those pieces'
        work in one kernel, give what the pieces would."""
        cls = type(self)
        return (
            cls.quantization_range is LinearQuantizer.quantization_range and cls._write_parameters_for_range is LinearQuantizer._write_parameters_for_range
            and cls.quantize is LinearQuantizer.quantize and cls.quantization_parameters is LinearQuantizer.quantization_parameters
            and cls.quantization_function is LinearQuantizer.quantization_function and cls.forward is Quantizer.forward
        )

    def wrap_codes(self, raw: torch.Tensor, data_dtype: torch.dtype) -> QuantizedTensor:
        """The QuantizedTensor ``self.quantize(data)`` returns, around codes `raw` that somebody else produced from `data` with this
        quantizer's current parameters (function.py: _static_quantize stamps the dequantize dtype at quantize time)."""
        params = self.quantization_parameters()
        stamped = params.with_changes(dequantize_dtype=params.dequantize_dtype or data_dtype)
        return QuantizedTensor(raw, QuantizationContext(self.quantization_function, stamped))

    def update_range_and_quantize(self, data: torch.Tensor, tile: Any, running_min: torch.Tensor, running_max: torch.Tensor,
                                  status: torch.Tensor | None) -> QuantizedTensor | None:
        """:meth:`update_range_from_data` AND ``self.quantize(data)`` in one pass over `data` (``ops.running_minmax_quantize``): the
        two things a RunningMinMax estimator's override does per call (reference range_setting/common.py:218-238). The result is the
        QuantizedTensor ``quantize`` returns with the parameters the step wrote. None — nothing touched — where the one-pass kernel
        or its preconditions do not apply; the caller then takes the two steps."""
        with torch.no_grad():
            if not self.is_plain() or not data.is_cuda or type(data) not in (torch.Tensor, torch.nn.Parameter) or running_min.numel() <= 1:
                return None
            from fastforward_amd import flags

            if flags.get_export_mode() or (torch.is_grad_enabled() and data.requires_grad):
                return None
            container = self.quantized_dtype or data.dtype
            if self.has_uninitialized_params:
                self._initialize_parameters(running_min.numel())
            if self.scale.device != data.device or self.scale.numel() != running_min.numel() or (self.offset is not None and self.offset.numel() != running_min.numel()):
                return None
            offset_out = self.offset.data if self.offset is not None else torch.empty(running_min.numel(), dtype=torch.float32, device=data.device)
            raw = ops.running_minmax_quantize(data, tile, running_min, running_max, status, self.num_bits, self.symmetric, self.allow_one_sided,
                                              self.scale.data, offset_out, container)
            if raw is None:
                return None
            torch.autograd.graph.increment_version(self.scale)  # written through raw pointers: tell the version counters
            if self.offset is not None:
                torch.autograd.graph.increment_version(self.offset)
            return self.wrap_codes(raw, data.dtype)

    def _write_parameters_for_range(self, lo: torch.Tensor, hi: torch.Tensor) -> None:
        """A5 straight into ``scale`` / ``offset`` — no host round trip (reference :350-357 + range.py)."""
        with torch.no_grad():
            if hi.device != lo.device:
                hi = hi.to(lo.device)
            direct = (
                self.scale.device == lo.device
                and self.scale.numel() == lo.numel()
                and self.scale.is_contiguous()
                and (self.offset is None or (self.offset.device == lo.device and self.offset.is_contiguous()))
            )
            if direct:
                ops.parameters_for_range(
                    lo, hi, self.num_bits, self.symmetric, self.allow_one_sided,
                    scale_out=self.scale.data,
                    offset_out=None if self.offset is None else self.offset.data,
                    want_offset=False,
                )
                # the kernel wrote through raw pointers: tell autograd's version counters, which every cache keyed on
                # `_version` (llama.FusedForward's weight codes, zero-offset and shared-range tables) relies on
                torch.autograd.graph.increment_version(self.scale)
                if self.offset is not None:
                    torch.autograd.graph.increment_version(self.offset)
                return
            scale, offset = ops.parameters_for_range(
                lo, hi, self.num_bits, self.symmetric, self.allow_one_sided, want_offset=self.offset is not None
            )
            self.scale.copy_(scale.reshape(self.scale.shape) if scale.numel() == self.scale.numel() else scale)
            if self.offset is not None and offset is not None:
                self.offset.copy_(offset.reshape(self.offset.shape) if offset.numel() == self.offset.numel() else offset)
