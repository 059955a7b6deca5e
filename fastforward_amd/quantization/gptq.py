"""GPTQ driver around the one-launch block kernel ``ffq_gptq_block`` (csrc/ffq_gptq.hip).

Replaces the reference's ``gptq()`` (src/fastforward/quantization/gptq.py:24-146 and its helpers :149-381; that file is itself
derived from IST-DASLab/gptq, Apache-2.0 — Frantar et al., "GPTQ: Accurate Post-Training Quantization for Generative
Pre-trained Transformers", 2022). The algorithm: visit the weight's columns in order; snap a column to the quantization grid,
divide the rounding error by the column's pivot of the upper Cholesky factor ``U`` of the damped inverse input covariance,
and subtract ``error x U[column, later columns]`` from the columns not yet visited. Results must equal the reference's
bit for bit on the CPU (fixture G13), so every floating-point statement keeps the reference's operation order; the
organisation below is this package's own:

* ``_InputCovariance``  — streaming ``2/n sum x x^T`` of the layer's inputs (reference ``calculate_hessian`` :285-321);
* ``_ParameterGrid``    — EVERY granularity as one rule: with ``(tr, tc) = granularity.tile_size(weight.shape)`` the
  parameters form a ``[rows / tr, columns / tc]`` grid, and column ``c`` sees ``grid[:, c // tc]`` repeated ``tr`` times
  (the reference enumerates the granularities one by one, :149-235);
* ``_Sweep``            — the working matrices and the two ways to process a block of columns: one launch of the HIP
  kernel when a row's parameters do not depend on the column (per-tensor / per-output-channel: rows are independent, a
  lane owns a row), else the column-by-column loop built from this package's quantize / dequantize ops.
"""

from __future__ import annotations

import logging
import math

from typing import Any, Callable, Iterable

import torch

import fastforward_amd as ff
import fastforward_amd.quantization.affine as affine_quant

from fastforward_amd import ops
from fastforward_amd.quantization import granularity as granularities

logger = logging.getLogger(__name__)

Dataset = Iterable[tuple[tuple[Any, ...], dict[str, Any]]]


class _InputCovariance:
    """``H = (2 / n) sum_t x_t x_t^T`` over all tokens seen, kept as a running mean in float64 (the per-batch product is a
    float32 matmul, as in the reference)."""

    def __init__(self, features: int, device: torch.device) -> None:
        self.matrix = torch.zeros((features, features), dtype=torch.float64, device=device)
        self.tokens = 0

    def update(self, activation: torch.Tensor) -> None:
        batch, seq, hidden = activation.shape
        columns = activation.to(device=self.matrix.device, dtype=torch.float32).reshape(batch * seq, hidden).transpose(0, 1)
        fresh = columns.shape[1]
        self.matrix.mul_(self.tokens / (self.tokens + fresh))
        self.tokens += fresh
        columns = columns * math.sqrt(2.0 / self.tokens)
        self.matrix.add_(columns @ columns.transpose(0, 1))

    def finish(self) -> torch.Tensor:
        """float32 matrix; input features that never fired get a unit diagonal so the factorisation exists."""
        silent = torch.diag(self.matrix) == 0
        self.matrix[silent, silent] = 1
        return self.matrix.float()


def _upper_factor_of_inverse(covariance: torch.Tensor, perc_damp: float) -> torch.Tensor:
    """``U`` with ``U^T U = (H + damp I)^-1``, upper triangular: damp by ``perc_damp x mean(diag)``, Cholesky, inverse
    from the factor, Cholesky of the inverse (the same three LAPACK calls as the reference :363-381)."""
    n = covariance.shape[0]
    ridge = perc_damp * torch.mean(torch.diag(covariance))
    where = torch.arange(n, device=covariance.device)
    covariance[where, where] += ridge
    inverse = torch.cholesky_inverse(torch.linalg.cholesky(covariance))
    return torch.linalg.cholesky(inverse, upper=True)


class _ParameterGrid:
    """The weight quantizer's scale / offset as a ``[rows / tr, columns / tc]`` grid over the weight."""

    def __init__(self, quantizer: "ff.nn.LinearQuantizer", weight_shape: torch.Size) -> None:
        granularity = quantizer.granularity
        if isinstance(granularity, granularities.PerBlock) and not granularity.strict_blocks:
            raise ValueError("GPTQ does not support PerBlock with strict_blocks=False.")
        if not isinstance(granularity, (granularities.PerTensor, granularities.PerChannel, granularities.PerBlock, granularities.PerTile)):
            raise TypeError(f"Unsupported granularity: {type(granularity).__name__}")
        rows, columns = weight_shape
        tile = granularity.tile_size(weight_shape)
        self.tile_rows, self.tile_cols = (rows, columns) if isinstance(tile, str) else (int(tile[0]), int(tile[1]))
        self.row_groups, self.col_groups = rows // self.tile_rows, columns // self.tile_cols
        self.quantizer = quantizer

    @property
    def column_independent(self) -> bool:
        """A row's parameters are the same in every column: the block kernel's case."""
        return self.col_groups == 1

    def _column_of(self, table: torch.Tensor, column: int) -> torch.Tensor:
        picked = table.view(self.row_groups, self.col_groups)[:, column // self.tile_cols]
        return picked.repeat_interleave(self.tile_rows) if self.tile_rows > 1 else picked

    def operator_for_column(self, column: int) -> Callable[[torch.Tensor], torch.Tensor]:
        """values [rows] -> quantize-dequantized values [rows], with the parameters column `column` of the weight sees."""
        q = self.quantizer
        scale = self._column_of(q.scale, column).contiguous()
        offset = None if q.offset is None else self._column_of(q.offset, column).contiguous()
        context = affine_quant.quantization_context(
            scale=scale, offset=offset, num_bits=q.num_bits, granularity=granularities.PerChannel(0), output_dtype=q.quantized_dtype
        )

        def quantize_dequantize(values: torch.Tensor) -> torch.Tensor:
            as_matrix = values.unsqueeze(1).contiguous()  # [rows, 1]: one parameter pair per row
            return context.quantization_fn.quantize(as_matrix, context.quantization_params).dequantize().flatten()

        return quantize_dequantize

    def refit_group(self, group: int, weights: torch.Tensor) -> None:
        """Grouped quantization: the scales of column group `group` are re-estimated on the error-corrected weights the
        sweep has reached (reference :91-99)."""
        first = group * self.tile_cols
        per_tile = weights[:, first : first + self.tile_cols].reshape(self.row_groups, -1)
        update_partial_range(
            self.quantizer, per_tile.min(dim=-1).values, per_tile.max(dim=-1).values,
            param_view_shape=(self.row_groups, self.col_groups), param_view_index=(slice(None), group),
        )


class _Sweep:
    """Working state of one layer: permuted weights, their snapped values, the scaled errors, the factor ``U``."""

    def __init__(self, weights: torch.Tensor, factor: torch.Tensor, order: torch.Tensor, grid: _ParameterGrid, refit_groups: bool) -> None:
        self.weights, self.factor, self.order, self.grid = weights, factor, order, grid
        self.snapped = torch.zeros_like(weights)
        self.errors = torch.zeros_like(weights)
        self.refit_groups = refit_groups

    def block_in_one_launch(self, start: int, width: int) -> bool:
        q = self.grid.quantizer
        return ops.gptq_block(self.weights, self.snapped, self.errors, start, width, self.factor, q.scale, q.offset, q.num_bits)

    def block_column_by_column(self, start: int, stop: int) -> None:
        local = self.weights[:, start:stop].clone()
        pivots = self.factor[start:stop, start:stop]
        for j in range(stop - start):
            column = start + j
            if self.refit_groups and column % self.grid.tile_cols == 0:
                self.grid.refit_group(column // self.grid.tile_cols, self.weights)
            snap = self.grid.operator_for_column(int(self.order[column].item()))
            self.snapped[:, column] = snap(local[:, j])
            self.errors[:, column] = (local[:, j] - self.snapped[:, column]) / pivots[j, j]
            local[:, j + 1 :] -= self.errors[:, column].unsqueeze(1) @ pivots[j : j + 1, j + 1 :]

    def push_errors_right(self, start: int, stop: int) -> None:
        self.weights[:, stop:] -= self.errors[:, start:stop] @ self.factor[start:stop, stop:]


def gptq(
    module: "ff.nn.QuantizedLinear",
    dataset: Dataset,
    block_size: int = 128,
    perc_damp: float = 0.01,
    actorder: bool = False,
    layer_name: str = "",
    fused: bool = True,
) -> None:
    """Replace ``module.weight`` by GPTQ-optimised values on its weight quantizer's grid (reference signature :24-32;
    ``fused=False`` forces the column loop where the block kernel would apply)."""
    quantizer = module.weight_quantizer
    if not isinstance(quantizer, ff.nn.LinearQuantizer):
        raise ValueError(f"weight_quantizer must be a LinearQuantizer, got {type(quantizer).__name__}.")
    shape = module.weight.shape
    grid = _ParameterGrid(quantizer, shape)
    weights = module.weight.data.clone().float()
    with ff.estimate_ranges(quantizer, ff.range_setting.smoothed_minmax):
        quantizer(weights)

    covariance = _InputCovariance(shape[1], module.weight.device)
    for (activation,), _ in dataset:
        covariance.update(activation)
    hessian = covariance.finish()
    columns = shape[1]
    order = torch.argsort(torch.diag(hessian), descending=True) if actorder else torch.arange(columns, device=hessian.device)
    weights = weights[:, order].contiguous()
    factor = _upper_factor_of_inverse(hessian[order][:, order], perc_damp).contiguous()

    grouped = isinstance(quantizer.granularity, (granularities.PerBlock, granularities.PerTile))
    sweep = _Sweep(weights, factor, order, grid, refit_groups=grouped and grid.col_groups > 1 and not actorder)
    kernel_applies = fused and grid.column_independent and grid.tile_rows in (1, shape[0]) and block_size <= 128 and quantizer.scale.dtype == torch.float32
    for start in range(0, columns, block_size):
        stop = min(start + block_size, columns)
        if not (kernel_applies and sweep.block_in_one_launch(start, stop - start)):
            sweep.block_column_by_column(start, stop)
        sweep.push_errors_right(start, stop)

    back = torch.argsort(order)
    with torch.no_grad():  # an in-place write autograd sees (the reference: module.weight.copy_): every cache keyed on
        module.weight.copy_(sweep.snapped[:, back].view(shape).to(module.weight.dtype))  # `_version` notices the new weights
    logger.info("[GPTQ][wbits=%d][%s] loss=%.6f", quantizer.num_bits, layer_name, torch.mean(torch.abs(sweep.errors[:, back])).item())


# ---- the two helpers the reference exposes beside gptq() (its tests call them: tests/quantization/test_gptq.py:47-140) ----
def column_quantizer(weight_quantizer: "ff.nn.LinearQuantizer", weight_shape: torch.Size, col_index: int) -> Callable[[torch.Tensor], torch.Tensor]:
    """Quantize-dequantize operator for column ``col_index`` of a ``weight_shape`` weight (reference :149-235)."""
    return _ParameterGrid(weight_quantizer, torch.Size(weight_shape)).operator_for_column(col_index)


def update_partial_range(
    weight_quantizer: "ff.nn.LinearQuantizer",
    min_range: torch.Tensor,
    max_range: torch.Tensor,
    *,
    param_view_shape: tuple[int, ...],
    param_view_index: Any,
) -> None:
    """A5 for a subset of the quantizer's parameters: ``(scale, offset)`` of ``(min_range, max_range)`` written at
    ``param_view_index`` of the parameters viewed as ``param_view_shape``; a two-sided range clears a stale offset
    (reference :238-282)."""
    scale, offset = affine_quant.parameters_for_range(
        min_range, max_range, num_bits=weight_quantizer.num_bits, symmetric=weight_quantizer.symmetric,
        allow_one_sided=weight_quantizer.allow_one_sided,
    )
    with torch.no_grad():  # indexed writes through views of the parameters themselves (not .data): their version counters move,
        weight_quantizer.scale.view(param_view_shape)[param_view_index] = scale.to(weight_quantizer.scale.dtype)  # which the code caches rely on
        if weight_quantizer.offset is not None:
            target = weight_quantizer.offset.view(param_view_shape)
            target[param_view_index] = 0.0 if offset is None else offset.to(target.dtype)
