"""Minimum-error range estimators (reference: src/fastforward/range_setting/min_error.py).

``mse_grid`` / ``min_error_grid`` search, per quantizer, a grid of candidate ranges for the one whose
quantize -> dequantize result is closest to the data. The reference evaluates the candidates one by one:
``num_candidates`` x (quantize, dequantize, error) eager passes per quantizer per step (:218-231).

For the default error (:func:`mse_error`) on a ``LinearQuantizer`` this module does the same search with the
candidates' parameters computed by A5 and ONE pass of the device kernel ``ffq_grid_sqerror_by_tile`` over the
batch for all candidates (csrc/ffq_griderror.hip). Custom error functions, quantizers without
``operator_for_range`` parameters, and tilings outside the kernel's range take the reference's loop.
"""

from __future__ import annotations

import dataclasses
import logging

from math import floor, sqrt
from typing import Any, Callable, Iterator

import torch

from fastforward_amd import ops
from fastforward_amd.forward_override import OverrideHandle
from fastforward_amd.nn.quantized_module import named_quantizers
from fastforward_amd.nn.quantizer import Quantizer
from fastforward_amd.quantization.tiled_tensor import tiles_to_rows
from fastforward_amd.range_setting.common import RangeEstimator, SimpleEstimatorStep, SupportsRangeBasedOperator

logger = logging.getLogger(__name__)


def mse_error(quantized_data: torch.Tensor, unquantized_data: torch.Tensor) -> torch.Tensor:
    """Mean squared error per row (reference :62-72)."""
    return torch.mean((quantized_data - unquantized_data) ** 2, dim=1)


@dataclasses.dataclass
class _UniformSearchGrid:
    """Candidate (min, max) thresholds of shape ``[num_candidates, parameter_dimensionality]`` (reference :75-145)."""

    absolute_margin: float = 0.5
    relative_margin: float = 1.0

    def __call__(self, tiled_data_sample: torch.Tensor, symmetric: bool, parameter_dimensionality: int, num_candidates: int) -> tuple[torch.Tensor, torch.Tensor]:
        assert tiled_data_sample.ndim == 2 and tiled_data_sample.shape[0] == parameter_dimensionality
        rel, absm = self.relative_margin, self.absolute_margin
        max_data = rel * tiled_data_sample.max(dim=1).values + absm
        min_data = rel * tiled_data_sample.min(dim=1).values - absm
        negative_data = bool(min_data.min() < 0)
        kw: dict[str, Any] = {"dtype": tiled_data_sample.dtype, "device": tiled_data_sample.device}
        if not negative_data:
            min_threshold = torch.zeros((num_candidates, parameter_dimensionality), **kw)
            steps = torch.linspace(1 / num_candidates, 1, num_candidates, **kw)
            max_threshold = steps.unsqueeze(1) * max_data.unsqueeze(0)
        elif not symmetric:
            margin = 0.6
            n_min = floor(sqrt(num_candidates))
            n_max = n_min + num_candidates - n_min**2
            steps_min = torch.linspace(1, margin, n_min, **kw)
            steps_max = torch.linspace(margin, 1, n_max, **kw)
            min_threshold = steps_min.unsqueeze(1) * (rel * min_data.unsqueeze(0) + absm)
            max_threshold = steps_max.unsqueeze(1) * (rel * max_data.unsqueeze(0) + absm)
            min_threshold = min_threshold.repeat(n_max, 1)
            max_threshold = max_threshold.repeat_interleave(n_min, dim=0)
        else:
            steps = torch.linspace(1 / num_candidates, 1, num_candidates, **kw)
            max_abs = torch.max(torch.abs(min_data), torch.abs(max_data))
            max_threshold = steps.unsqueeze(1) * max_abs.unsqueeze(0)
            min_threshold = -max_threshold
        return min_threshold, max_threshold


def uniform_search_grid(absolute_margin: float = 0.5, relative_margin: float = 1.0) -> _UniformSearchGrid:
    return _UniformSearchGrid(absolute_margin=absolute_margin, relative_margin=relative_margin)


class _MinAvgErrorGridEstimator(SimpleEstimatorStep, torch.nn.Module):
    """Override installed on one quantizer: accumulates the error of every candidate over the batches seen and
    sets the quantizer to the best one (reference :171-231)."""

    min_threshold: torch.Tensor
    max_threshold: torch.Tensor
    cumulative_error: torch.Tensor

    def __init__(
        self,
        quantizer: SupportsRangeBasedOperator,
        error_fn: Callable[[torch.Tensor, torch.Tensor], torch.Tensor] = mse_error,
        num_candidates: int = 100,
        search_grid_generator: Callable[..., tuple[torch.Tensor, torch.Tensor]] = _UniformSearchGrid(),
        update_range_policy: Callable[["_MinAvgErrorGridEstimator", int], bool] | None = None,
        disable_quantization: bool = False,
    ) -> None:
        super().__init__(disable_quantization=disable_quantization)
        self._quantizer = quantizer
        self.error_fn = error_fn
        self.num_candidates = num_candidates
        self.search_grid_generator = search_grid_generator
        self._estimation_steps = 0
        self.update_range_policy = update_range_policy
        self._candidate_params: tuple[torch.Tensor, torch.Tensor | None] | None = None
        self.used_fused_kernel = False

    def setup_estimator(self, data: torch.Tensor) -> None:
        self._estimation_steps = 0
        granularity = self._quantizer.granularity
        dims = granularity.parameter_dimensionality(data.shape)
        tile = granularity.tile_size(data.shape)
        tile = data.shape if isinstance(tile, str) else tile
        tiled = tiles_to_rows(data.detach(), tile)
        self.min_threshold, self.max_threshold = self.search_grid_generator(tiled, self._quantizer.symmetric, dims, self.num_candidates)
        self.cumulative_error = torch.zeros_like(self.min_threshold)
        self._candidate_params = None

    def _update_quantizer_ranges(self, quantizer: SupportsRangeBasedOperator) -> None:
        best = self.cumulative_error.min(dim=0).indices
        idx = torch.arange(self.min_threshold.shape[1], device=best.device)
        quantizer.quantization_range = (self.min_threshold[best, idx], self.max_threshold[best, idx])

    # ---- fused path -----------------------------------------------------------------------------------
    def _fused_parameters(self, quantizer: Any) -> tuple[torch.Tensor, torch.Tensor | None] | None:
        """(scales, offsets) [candidates, tiles] of all candidates — what ``operator_for_range`` would use (A5 per
        candidate, because its one-sided decision is global over the tiles of ONE candidate)."""
        if self._candidate_params is not None:
            return self._candidate_params
        needed = ("num_bits", "symmetric", "allow_one_sided", "granularity")
        if self.error_fn is not mse_error or not all(hasattr(quantizer, n) for n in needed) or not hasattr(quantizer, "_parameters_for_range"):
            return None
        # The reference evaluates candidates range(num_candidates) (:219) although the asymmetric grid holds
        # floor(sqrt(n)) * (floor(sqrt(n)) + n - floor(sqrt(n))**2) rows — more than n unless n is a perfect square;
        # rows past num_candidates keep a cumulative error of 0. Same here.
        n_cand, n_tiles = min(self.num_candidates, self.min_threshold.shape[0]), self.min_threshold.shape[1]
        device = self.min_threshold.device
        scales = torch.empty((n_cand, n_tiles), dtype=torch.float32, device=device)
        offsets = torch.empty((n_cand, n_tiles), dtype=torch.float32, device=device)
        for i in range(n_cand):
            ops.parameters_for_range(
                self.min_threshold[i], self.max_threshold[i], quantizer.num_bits, quantizer.symmetric, quantizer.allow_one_sided,
                scale_out=scales[i], offset_out=offsets[i], want_offset=False,
            )
        has_offset = getattr(quantizer, "offset", None) is not None
        self._candidate_params = (scales, offsets if has_offset else None)
        return self._candidate_params

    def _fused_step(self, quantizer: Any, data: torch.Tensor) -> bool:
        params = self._fused_parameters(quantizer)
        if params is None or data.dtype not in (torch.float32, torch.bfloat16, torch.float16):
            return False
        tile = quantizer.granularity.tile_size(data.shape)
        tile = data.shape if isinstance(tile, str) else tile
        sums = ops.grid_sqerror_by_tile(data, params[0], params[1], tile, quantizer.num_bits)
        if sums is None:
            return False
        tile_numel = data.numel() // params[0].shape[1]
        err = (sums / tile_numel).to(self.cumulative_error.dtype)  # torch.mean: fp32 accumulation, result in the data dtype
        self.cumulative_error[: err.shape[0]] += err
        return True

    def estimate_step(self, quantizer: Any, data: torch.Tensor) -> None:
        with torch.no_grad():
            raw = data.detach()
            self.used_fused_kernel = self._fused_step(quantizer, raw)
            if not self.used_fused_kernel:
                tile = quantizer.granularity.tile_size(raw.shape)
                tile = raw.shape if isinstance(tile, str) else tile
                tiled = tiles_to_rows(raw, tile)
                for i in range(self.num_candidates):
                    op = quantizer.operator_for_range(self.min_threshold[i], self.max_threshold[i], raw.shape)
                    tiled_q = tiles_to_rows(op(raw).dequantize(), tile)
                    self.cumulative_error[i] += self.error_fn(tiled_q, tiled)
        self._estimation_steps += 1
        if not self.update_range_policy or self.update_range_policy(self, self._estimation_steps):
            self._update_quantizer_ranges(quantizer)


class MinErrorGridRangeEstimator(RangeEstimator[OverrideHandle, Quantizer]):
    """``ff.range_setting.mse_grid`` / ``min_error_grid`` (reference :234-318)."""

    def __init__(
        self,
        error_fn: Callable[[torch.Tensor, torch.Tensor], torch.Tensor] = mse_error,
        num_candidates: int = 100,
        search_grid_generator: Callable[..., tuple[torch.Tensor, torch.Tensor]] = _UniformSearchGrid(),
        update_range_policy: Callable[[_MinAvgErrorGridEstimator, int], bool] | None = None,
        skip_unsupported_quantizers: bool = False,
    ) -> None:
        self._error_fn = error_fn
        self._num_candidates = num_candidates
        self._search_grid_generator = search_grid_generator
        self._update_range_policy = update_range_policy
        self._skip_unsupported_quantizers = skip_unsupported_quantizers

    def prepare(self, module: Quantizer) -> OverrideHandle:
        if not isinstance(module, SupportsRangeBasedOperator):
            name = f"{SupportsRangeBasedOperator.__module__}.{SupportsRangeBasedOperator.__qualname__}"
            raise TypeError(f"{type(module).__name__} does not implement {name}.")
        return module.register_override(
            _MinAvgErrorGridEstimator(
                module, error_fn=self._error_fn, num_candidates=self._num_candidates,
                search_grid_generator=self._search_grid_generator, update_range_policy=self._update_range_policy,
            )
        )

    def cleanup(self, module: Quantizer, metadata: OverrideHandle) -> None:
        del module
        metadata.remove()

    def split_module(self, module: torch.nn.Module) -> Iterator[Quantizer]:
        for _, quantizer in named_quantizers(module, recurse=True):
            if isinstance(quantizer, SupportsRangeBasedOperator) or not self._skip_unsupported_quantizers:
                yield quantizer
            else:
                logger.warning(
                    "%s does not implement SupportsRangeBasedOperator. Therefore it is not included in %s range setting.",
                    type(quantizer).__name__, type(self).__name__,
                )


min_error_grid = MinErrorGridRangeEstimator
mse_grid = MinErrorGridRangeEstimator
