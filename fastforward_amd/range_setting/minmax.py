"""Min/max range estimators (reference: src/fastforward/range_setting/minmax.py).

``RunningMinMaxEstimator.estimate_step`` is hot-path row A4: per-tile min and max of the batch,
merged into the running extrema, followed by the range setter (A5). The reference performs two
full-tensor reductions and two host synchronisations per quantizer per step (:229-234 and
affine/range.py:100). Here one kernel produces both extrema and a device-side status word replaces
the ``isinf().any()`` sync:

  * default (``sync_free=False``): behaves exactly like the reference — the status word is read
    after the kernel and ``NotImplementedError("Infinite")`` is raised in the same step, before the
    running extrema are touched;
  * ``sync_free=True``: the running extrema are merged in place on the device, nothing is read
    back, and the error is raised when the estimator is removed (end of ``estimate_ranges``). This is
    the mode the multi-GPU calibration uses: a whole calibration run enqueues without a host wait.
"""

from __future__ import annotations

import logging

from typing import Any, Iterator

import torch

from fastforward_amd import ops
from fastforward_amd.forward_override import OverrideHandle
from fastforward_amd.nn.quantized_module import named_quantizers
from fastforward_amd.nn.quantizer import Quantizer
from fastforward_amd.range_setting.common import RangeEstimator, RangeSettable, SimpleEstimatorStep

logger = logging.getLogger(__name__)


class _MinMaxState(SimpleEstimatorStep, torch.nn.Module):
    min: torch.Tensor | None
    max: torch.Tensor | None

    def __init__(self, quantizer: RangeSettable, disable_quantization: bool = False) -> None:
        super().__init__(disable_quantization=disable_quantization)
        lo, hi = quantizer.quantization_range
        self.register_buffer("min", lo)
        self.register_buffer("max", hi)

    def initialize_parameters(self, quantizer: RangeSettable, data: torch.Tensor) -> None:
        """+inf / -inf in the data dtype, one per tile (reference :202-213)."""
        shape = (quantizer.granularity.parameter_dimensionality(data.shape),)
        if self.min is None:
            self.min = data.new_full(shape, float("inf"))
        if self.max is None:
            self.max = data.new_full(shape, float("-inf"))

    def extra_repr(self) -> str:
        return f"min={self.min}, max={self.max}"


class RunningMinMaxEstimator(_MinMaxState):
    """Running min / running max over all batches seen (reference :179-247)."""

    def __init__(self, quantizer: RangeSettable, disable_quantization: bool = False, sync_free: bool = False) -> None:
        super().__init__(quantizer, disable_quantization=disable_quantization)
        self.sync_free = sync_free
        self.status: torch.Tensor | None = None  # int32[1] on the data's device

    def raise_if_infinite(self) -> None:
        """Deferred form of the reference's ``isinf().any()`` check (:233-234)."""
        if self.status is not None and int(self.status.item()) & ops.FLAG_INF:
            raise NotImplementedError("Infinite")

    def forward(self, quantizer: Any, callback: Any, args: tuple[Any, ...], kwargs: dict[str, Any]) -> torch.Tensor:
        """``estimate_step`` then the quantizer's forward (reference common.py:218-238) — as ONE pass over the data where this
        estimator is the quantizer's only override, works sync-free and the quantizer offers ``update_range_and_quantize`` (per-channel /
        per-token / per-block tilings of a plain LinearQuantizer on the device): a weight is read once per calibration step
        instead of twice. Same running state, parameters, status flags and codes as the two steps."""
        # a subclass that overrides estimate_step / initialize_parameters (logging, clamping, an EMA on this class's state) must see
        # every batch: the one-pass route is this class's own two steps fused, nothing else's
        own_steps = (type(self).estimate_step is RunningMinMaxEstimator.estimate_step
                     and type(self).initialize_parameters is RunningMinMaxEstimator.initialize_parameters)
        if own_steps and self.sync_free and not self._disable_quantization and not kwargs and len(args) == 1 and isinstance(args[0], torch.Tensor):
            fused = getattr(quantizer, "update_range_and_quantize", None)
            overrides = getattr(quantizer, "_quantizer_overrides", None)
            data = args[0]
            if fused is not None and overrides is not None and len(overrides) == 1 and next(iter(overrides.values())) is self:
                if not self._initialized:
                    self.setup_estimator(data)
                    self._initialized = True
                self.initialize_parameters(quantizer, data)
                assert self.min is not None and self.max is not None
                if self.min.dtype == data.dtype and self.max.dtype == data.dtype and self.min.device == data.device and self.min.numel() > 1:
                    tile = quantizer.granularity.tile_size(data.shape)
                    tile = data.shape if isinstance(tile, str) else tile
                    if self.status is None or self.status.device != data.device:
                        self.status = torch.zeros(1, dtype=torch.int32, device=data.device)
                    out = fused(data, tile, self.min, self.max, self.status)
                    if out is not None:
                        return out
        return super().forward(quantizer, callback, args, kwargs)

    def estimate_step(self, quantizer: RangeSettable, data: torch.Tensor) -> None:
        self.initialize_parameters(quantizer, data)
        assert self.min is not None and self.max is not None
        with torch.no_grad():
            raw = data.detach()
            tile = quantizer.granularity.tile_size(raw.shape)
            tile = raw.shape if isinstance(tile, str) else tile
            if self.status is None or self.status.device != raw.device:
                self.status = torch.zeros(1, dtype=torch.int32, device=raw.device)
            same_kind = self.min.dtype == raw.dtype and self.max.dtype == raw.dtype and self.min.device == raw.device
            if self.sync_free and same_kind:
                # reduction, running merge, status flags AND the range setter in one entry point (one launch per tensor for a
                # per-tensor quantizer) where the quantizer's parameters can be written in place; else the two steps below
                update = getattr(quantizer, "update_range_from_data", None)
                # (the caller's own tensor object where autograd is not involved: sibling estimators recognise a shared input by it)
                if update is not None and update(raw if data.requires_grad else data, tile, self.min, self.max, self.status):
                    return
                ops.minmax_by_tile(raw, tile, running_min=self.min, running_max=self.max, status_flags=self.status)
            elif same_kind:
                lo, hi = self.min.clone(), self.max.clone()
                ops.minmax_by_tile(raw, tile, running_min=lo, running_max=hi, status_flags=self.status)
                self.raise_if_infinite()
                self.min, self.max = lo, hi
            else:
                # running values restored from a quantizer's fp32 range while the data is half
                # precision: merge under torch's promotion like the reference     (:236-237)
                batch_lo, batch_hi = ops.minmax_by_tile(raw, tile, status_flags=self.status)
                if not self.sync_free:
                    self.raise_if_infinite()
                self.min = torch.min(self.min, batch_lo.to(self.min.device))
                self.max = torch.max(self.max, batch_hi.to(self.max.device))
        quantizer.quantization_range = (self.min, self.max)


class SmoothedMinMaxEstimator(_MinMaxState):
    """Exponential moving average of per-batch min / max (reference :22-92)."""

    def __init__(self, quantizer: RangeSettable, gamma: float = 1.0, disable_quantization: bool = False) -> None:
        super().__init__(quantizer, disable_quantization=disable_quantization)
        self.gamma = gamma

    def estimate_step(self, quantizer: RangeSettable, data: torch.Tensor) -> None:
        self.initialize_parameters(quantizer, data)
        assert self.min is not None and self.max is not None
        with torch.no_grad():
            raw = data.detach()
            tile = quantizer.granularity.tile_size(raw.shape)
            tile = raw.shape if isinstance(tile, str) else tile
            batch_lo, batch_hi = ops.minmax_by_tile(raw, tile)
            if self.min.isinf().any() or self.max.isinf().any():  # first batch: adopt it (:81-84)
                self.min, self.max = batch_lo, batch_hi
            else:
                self.min = self.gamma * batch_lo + (1 - self.gamma) * self.min
                self.max = self.gamma * batch_hi + (1 - self.gamma) * self.max
        quantizer.quantization_range = (self.min, self.max)


class _MinMaxRangeEstimator(RangeEstimator[OverrideHandle, Quantizer]):
    skip_unsupported_quantizers: bool = False

    def _make(self, module: Quantizer) -> Any:
        raise NotImplementedError

    def prepare(self, module: Quantizer) -> OverrideHandle:
        if not isinstance(module, RangeSettable):
            raise TypeError(
                f"{type(module).__name__} does not implement {RangeSettable.__module__}.{RangeSettable.__qualname__}."
            )
        return module.register_override(self._make(module))

    def cleanup(self, module: Quantizer, metadata: OverrideHandle) -> None:
        estimator = metadata.remove()
        if isinstance(estimator, RunningMinMaxEstimator) and estimator.sync_free:
            estimator.raise_if_infinite()

    def split_module(self, module: torch.nn.Module) -> Iterator[Quantizer]:
        for _, quantizer in named_quantizers(module, recurse=True):
            if isinstance(quantizer, RangeSettable) or not self.skip_unsupported_quantizers:
                yield quantizer
            else:
                logger.warning(
                    "%s does not implement RangeSettable. Therefore it is not included in %s range setting.",
                    type(quantizer).__name__, type(self).__name__,
                )


class RunningMinMaxRangeEstimator(_MinMaxRangeEstimator):
    """``ff.range_setting.running_minmax`` (reference :250-306)."""

    def __init__(self, disable_quantization: bool = False, skip_unsupported_quantizers: bool = False, sync_free: bool = False) -> None:
        self.disable_quantization = disable_quantization
        self.skip_unsupported_quantizers = skip_unsupported_quantizers
        self.sync_free = sync_free

    def _make(self, module: Quantizer) -> RunningMinMaxEstimator:
        return RunningMinMaxEstimator(module, disable_quantization=self.disable_quantization, sync_free=self.sync_free)  # type: ignore[arg-type]


class SmoothedMinMaxRangeEstimator(_MinMaxRangeEstimator):
    """``ff.range_setting.smoothed_minmax`` (reference :95-176)."""

    def __init__(self, gamma: float = 1.0, disable_quantization: bool = False, skip_unsupported_quantizers: bool = False) -> None:
        self.gamma = gamma
        self.disable_quantization = disable_quantization
        self.skip_unsupported_quantizers = skip_unsupported_quantizers

    def _make(self, module: Quantizer) -> SmoothedMinMaxEstimator:
        return SmoothedMinMaxEstimator(module, gamma=self.gamma, disable_quantization=self.disable_quantization)  # type: ignore[arg-type]


running_minmax = RunningMinMaxRangeEstimator
smoothed_minmax = SmoothedMinMaxRangeEstimator
