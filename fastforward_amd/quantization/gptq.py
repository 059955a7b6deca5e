"""GPTQ for quantized linears (reference: src/fastforward/quantization/gptq.py).

``gptq(module, dataset)`` snaps a ``QuantizedLinear``'s weight to its quantization grid column by column while
pushing each column's error onto the columns not yet quantized, using the inverse Hessian of the layer's inputs.
The reference runs the column loop in eager mode: about five launches per column on ``[out_features]`` vectors,
20 k launches for a 4096-column weight. For per-tensor and per-output-channel weight quantizers (one scale/offset
per ROW, the same for every column) the whole loop over a block of columns is one launch of
``ffq_gptq_block`` (csrc/ffq_gptq.hip): rows are independent, a lane owns a row. Column-dependent granularities
(per-input-channel, per-block with grouped-scale recomputation, per-tile) take the reference's loop, built from
this package's quantize / dequantize ops.
"""

from __future__ import annotations

import logging
import math

from typing import Any, Callable, Iterable, cast

import torch

import fastforward_amd as ff
import fastforward_amd.quantization.affine as affine_quant

from fastforward_amd import ops
from fastforward_amd.quantization import granularity as granularities

logger = logging.getLogger(__name__)


def _per_row_parameters(granularity: granularities.Granularity) -> bool:
    """One (scale, offset) per row of the weight, identical for every column."""
    if isinstance(granularity, granularities.PerTensor):
        return True
    return isinstance(granularity, granularities.PerChannel) and tuple(granularity.channel_dims) == (0,)


def gptq(
    module: "ff.nn.QuantizedLinear",
    dataset: Iterable[tuple[tuple[Any, ...], dict[str, Any]]],
    block_size: int = 128,
    perc_damp: float = 0.01,
    actorder: bool = False,
    layer_name: str = "",
    fused: bool = True,
) -> None:
    """Quantize a QuantizedLinear in place using GPTQ (reference :25-146). ``fused=False`` forces the column loop."""
    if not isinstance(module.weight_quantizer, ff.nn.LinearQuantizer):
        raise ValueError(f"weight_quantizer must be a LinearQuantizer, got {type(module.weight_quantizer).__name__}.")
    granularity = module.weight_quantizer.granularity
    if isinstance(granularity, granularities.PerBlock) and not granularity.strict_blocks:
        raise ValueError("GPTQ does not support PerBlock with strict_blocks=False.")

    original_weight_shape = module.weight.shape
    weights = module.weight.data.clone().float()
    columns = weights.shape[1]
    weight_quantizer = module.weight_quantizer

    with ff.estimate_ranges(weight_quantizer, ff.range_setting.smoothed_minmax):
        weight_quantizer(weights)

    hessian = calculate_hessian(module, dataset)
    column_order = torch.argsort(torch.diag(hessian), descending=True) if actorder else torch.arange(columns, device=hessian.device)
    weights = weights[:, column_order].contiguous()
    hessian = hessian[column_order][:, column_order]

    quantized_weights = torch.zeros_like(weights)
    errors = torch.zeros_like(weights)
    hessian_inverse = invert_hessian(hessian, perc_damp).contiguous()

    # For grouped quantization only: recompute each group's scale on its error-corrected weights (reference :91-99).
    recompute_scales = False
    col_block_size = num_row_blocks = num_col_blocks = 0
    if isinstance(granularity, (granularities.PerBlock, granularities.PerTile)):
        row_block_size, col_block_size = granularity.tile_size(weights.shape)
        num_row_blocks = weights.shape[0] // row_block_size
        num_col_blocks = weights.shape[1] // col_block_size
        recompute_scales = num_col_blocks > 1 and not actorder

    use_kernel = fused and _per_row_parameters(granularity) and block_size <= 128 and weight_quantizer.scale.dtype == torch.float32
    for i in range(0, columns, block_size):
        block_end = min(i + block_size, columns)
        done = False
        if use_kernel:
            done = ops.gptq_block(weights, quantized_weights, errors, i, block_end - i, hessian_inverse,
                                  weight_quantizer.scale, weight_quantizer.offset, weight_quantizer.num_bits)
        if not done:
            weights_block = weights[:, i:block_end].clone()
            hessinv_block = hessian_inverse[i:block_end, i:block_end]
            for j in range(block_end - i):
                global_col = i + j
                if recompute_scales and global_col % col_block_size == 0:
                    col_block_idx = global_col // col_block_size
                    group_weights = weights[:, global_col : global_col + col_block_size]
                    reshaped = group_weights.reshape(num_row_blocks, -1)
                    update_partial_range(
                        weight_quantizer, reshaped.min(dim=-1).values, reshaped.max(dim=-1).values,
                        param_view_shape=(num_row_blocks, num_col_blocks), param_view_index=(slice(None), col_block_idx),
                    )
                orig_col = int(column_order[i + j].item())
                quant_deq = column_quantizer(weight_quantizer, weights.shape, orig_col)
                quantized_weights[:, i + j] = quant_deq(weights_block[:, j])
                errors[:, i + j] = (weights_block[:, j] - quantized_weights[:, i + j]) / hessinv_block[j, j]
                weights_block[:, j + 1 :] -= errors[:, i + j].unsqueeze(1) @ hessinv_block[j : j + 1, j + 1 :]
        weights[:, block_end:] -= errors[:, i:block_end] @ hessian_inverse[i:block_end, block_end:]

    restore_order = torch.argsort(column_order)
    quantized_weights = quantized_weights[:, restore_order]
    errors = errors[:, restore_order]
    module.weight.data.copy_(quantized_weights.view(original_weight_shape).to(module.weight.dtype))
    loss = torch.mean(torch.abs(errors)).item()
    logger.info("[GPTQ][wbits=%d][%s] loss=%.6f", module.weight_quantizer.num_bits, layer_name, loss)


def column_quantizer(weight_quantizer: "ff.nn.LinearQuantizer", weight_shape: torch.Size, col_index: int) -> Callable[[torch.Tensor], torch.Tensor]:
    """Quantize-dequantize operator for ONE column: the quantizer's granularity restated as one (scale, offset) per
    row for that column (reference :149-235)."""
    out_features, in_features = weight_shape
    scale: torch.Tensor = weight_quantizer.scale
    offset: torch.Tensor | None = weight_quantizer.offset
    granularity = weight_quantizer.granularity

    if isinstance(granularity, granularities.PerTensor):
        scale = scale.expand(out_features)
        offset = offset.expand(out_features) if offset is not None else None
    elif isinstance(granularity, granularities.PerChannel) and tuple(granularity.channel_dims) == (0,):
        scale = scale.reshape(out_features)
        offset = offset.reshape(out_features) if offset is not None else None
    elif isinstance(granularity, granularities.PerChannel) and tuple(granularity.channel_dims) == (1,):
        scale = scale[col_index].expand(out_features)
        offset = offset[col_index].expand(out_features) if offset is not None else None
    elif isinstance(granularity, granularities.PerChannel) and tuple(granularity.channel_dims) == (0, 1):
        scale = scale.view(out_features, in_features)[:, col_index]
        offset = offset.view(out_features, in_features)[:, col_index] if offset is not None else None
    elif isinstance(granularity, granularities.PerBlock) and not granularity.strict_blocks:
        raise ValueError("GPTQ does not support PerBlock with strict_blocks=False.")
    elif isinstance(granularity, (granularities.PerBlock, granularities.PerTile)):
        row_block_size, col_block_size = granularity.tile_size(weight_shape)
        num_row_blocks = out_features // row_block_size
        num_col_blocks = in_features // col_block_size
        col_block_idx = col_index // col_block_size
        scale = scale.view(num_row_blocks, num_col_blocks)[:, col_block_idx].repeat_interleave(row_block_size)
        offset = offset.view(num_row_blocks, num_col_blocks)[:, col_block_idx].repeat_interleave(row_block_size) if offset is not None else None
    else:
        raise TypeError(f"Unsupported granularity: {type(granularity).__name__}")

    ctx = affine_quant.quantization_context(
        scale=scale.contiguous(), offset=None if offset is None else offset.contiguous(), num_bits=weight_quantizer.num_bits,
        granularity=granularities.PerChannel(0), output_dtype=weight_quantizer.quantized_dtype,
    )

    def _quant_fn(col: torch.Tensor) -> torch.Tensor:
        q = ctx.quantization_fn.quantize(col.unsqueeze(1).contiguous(), ctx.quantization_params)
        return q.dequantize().flatten()

    return _quant_fn


def update_partial_range(
    weight_quantizer: "ff.nn.LinearQuantizer",
    min_range: torch.Tensor,
    max_range: torch.Tensor,
    *,
    param_view_shape: tuple[int, ...],
    param_view_index: Any,
) -> None:
    """Write scale/offset for a subset of parameter positions from a (min, max) range (reference :238-282)."""
    scale, offset = affine_quant.parameters_for_range(
        min_range, max_range, num_bits=weight_quantizer.num_bits, symmetric=weight_quantizer.symmetric,
        allow_one_sided=weight_quantizer.allow_one_sided,
    )
    scale_view = weight_quantizer.scale.data.view(param_view_shape)
    scale_view[param_view_index] = scale.to(scale_view.dtype)
    if weight_quantizer.offset is not None:
        offset_view = weight_quantizer.offset.data.view(param_view_shape)
        if offset is not None:
            offset_view[param_view_index] = offset.to(offset_view.dtype)
        else:
            offset_view[param_view_index] = 0.0


def calculate_hessian(layer: "ff.nn.QuantizedLinear", activations: Iterable[tuple[tuple[Any, ...], dict[str, Any]]]) -> torch.Tensor:
    """Running mean of ``2 x xᵀ`` over the layer's inputs, float64 accumulation (reference :285-321)."""
    device = layer.weight.device
    in_features = layer.weight.shape[1]
    hessian = torch.zeros((in_features, in_features), device=device, dtype=torch.float64)
    n_samples = 0
    for (activation,), _ in activations:
        activation = cast(torch.Tensor, activation).to(device=device, dtype=torch.float32)
        bsz, seq_len, hidden = activation.shape
        x = activation.reshape(bsz * seq_len, hidden).transpose(0, 1)
        hessian.mul_(n_samples / (n_samples + x.shape[1]))
        n_samples += x.shape[1]
        x = x * math.sqrt(2.0 / n_samples)
        hessian.add_(x @ x.transpose(0, 1))
    dead = torch.diag(hessian) == 0
    hessian[dead, dead] = 1
    return hessian.float()


def invert_hessian(hessian: torch.Tensor, perc_damp: float) -> torch.Tensor:
    """Upper Cholesky factor of the damped inverse Hessian (reference :363-381)."""
    dampening = perc_damp * torch.mean(torch.diag(hessian))
    diag = torch.arange(hessian.shape[0], device=hessian.device)
    hessian[diag, diag] += dampening
    hessian = torch.linalg.cholesky(hessian)
    hessian = torch.cholesky_inverse(hessian)
    return torch.linalg.cholesky(hessian, upper=True)
