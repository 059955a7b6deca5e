"""Fold weight quantizers into the weights (reference: src/fastforward/quantization/fuse.py:53-273).

``fuse_qdq_weights(model)`` overwrites every weight that has an initialised weight quantizer with its
quantize -> dequantize value (A1 then A2 on the device), so a later forward either re-quantizes a grid-snapped
weight (a no-op for affine quantizers whose parameters did not change) or, with ``stub_quantizers=True``, skips
the weight quantizer altogether. Together with ``llama.FusedForward(cache_weight_codes=True)`` — int8 codes kept
per (weight, scale, offset) version and fed straight to the int8 GEMM — this is the "weights are quantized once"
mode of SURVEY §8(f) row 1; the default everywhere stays the reference's re-quantize-every-forward.
"""

from __future__ import annotations

import collections

from typing import Iterator

import torch

import fastforward_amd as ff

from fastforward_amd.exceptions import QuantizationError
from fastforward_amd.nn.quantizer import Quantizer, QuantizerStub, default_tags

WeightQuantizerTarget = tuple[torch.nn.Module, str, Quantizer]


class ConventionDiscovery:
    """Weight quantizers by convention: a child quantizer tagged `tag` next to a Parameter named `weight_attr`
    (how QuantizedLinear wires its weight quantizer; reference :53-88)."""

    def __init__(self, weight_attr: str = "weight", *, tag: str = "parameter/weight") -> None:
        self._weight_attr = weight_attr
        self._tag = tag

    def __call__(self, model: torch.nn.Module) -> Iterator[WeightQuantizerTarget]:
        for parent in model.modules():
            weight = getattr(parent, self._weight_attr, None)
            if not isinstance(weight, torch.nn.Parameter):
                continue
            for _, child in parent.named_children():
                if not isinstance(child, Quantizer) or child.is_stub():
                    continue
                meta = child.quant_metadata
                tagged = meta is not None and self._tag in meta
                # quantizers installed by attribute assignment inherit the slot's metadata; fall back to the slot name
                by_name = meta is None and self._tag == repr(default_tags.weight_quantizer) and child is getattr(parent, "weight_quantizer", None)
                if tagged or by_name:
                    yield parent, self._weight_attr, child


def find_weight_quantizers(model: torch.nn.Module, *, discovery: ConventionDiscovery | None = None) -> list[WeightQuantizerTarget]:
    """The targets :func:`fuse_qdq_weights` would act on (reference :244-273)."""
    return list((discovery or ConventionDiscovery())(model))


def _check_tied(model: torch.nn.Module, targets: list[WeightQuantizerTarget]) -> None:
    """A weight tied across modules must be snapped to ONE grid (reference :140-196)."""
    by_storage: dict[tuple[int, torch.Size], list[WeightQuantizerTarget]] = collections.defaultdict(list)
    for module, attr, quantizer in targets:
        weight = getattr(module, attr)
        by_storage[(weight.data_ptr(), weight.shape)].append((module, attr, quantizer))
    names = {id(m): n for n, m in model.named_modules()}
    for group in by_storage.values():
        if len(group) == 1:
            continue
        weight = getattr(group[0][0], group[0][1])
        reference: torch.Tensor | None = None
        for module, attr, quantizer in group:
            with ff.strict_quantization(False):
                qdq = quantizer(weight).dequantize()
            if reference is None:
                reference = qdq
            elif not torch.equal(reference, qdq):
                listed = ", ".join(f"{names.get(id(m), type(m).__name__)}.{a}" for m, a, _ in group)
                raise QuantizationError(
                    f"Cannot fuse QDQ weights: the weight shared by [{listed}] is tied across modules whose weight "
                    "quantizers snap it to different grids, so fusing would leave it correct for at most one of them."
                )


def fuse_qdq_weights(model: torch.nn.Module, *, stub_quantizers: bool = False, discovery: ConventionDiscovery | None = None) -> None:
    """Replace a model's weights with their quantize-dequantize values in place (reference :199-242)."""
    targets = find_weight_quantizers(model, discovery=discovery)
    _check_tied(model, targets)
    for module, attr, quantizer in targets:
        weight = getattr(module, attr)
        with ff.strict_quantization(False):
            qdq = quantizer(weight).dequantize()
        if qdq is weight:
            continue
        with torch.no_grad():
            weight.copy_(qdq)
        if stub_quantizers:
            for name, sibling in module.named_children():
                if sibling is quantizer:
                    setattr(module, name, QuantizerStub(_metadata=quantizer.quant_metadata))
                    break
