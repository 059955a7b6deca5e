"""Shared helpers of the test-suite: run one fixture case through whichever backend is active."""

from __future__ import annotations

import torch

import fastforward_amd as ff

from datagen import dtype_from_name, make_data  # tests/golden/datagen.py


def granularity_of(spec):
    kind = spec[0]
    if kind == "tensor":
        return ff.PerTensor()
    if kind == "channel":
        return ff.PerChannel(tuple(spec[1]) if isinstance(spec[1], (list, tuple)) else spec[1])
    if kind == "block":
        return ff.PerBlock(block_dims=tuple(spec[1]), block_sizes=tuple(spec[2]), per_channel_dims=tuple(spec[3]))
    if kind == "tile":
        return ff.PerTile(tuple(spec[1]))
    raise ValueError(spec)


def case_input(case) -> torch.Tensor:
    if case.get("data") is not None:
        return case["data"]
    return make_data(case["seed"], tuple(case["shape"]), dtype_from_name(case["dtype"]), case["kind"])


def to_device(t, device):
    return None if t is None else t.to(device)


def same_with_nan(a: torch.Tensor, b: torch.Tensor, signed_zero: bool = True) -> bool:
    """Bit-for-bit equality of values, NaN == NaN (sign/payload of NaN is not part of the contract).

    With signed_zero (default) -0.0 and +0.0 are different values.
    """
    if a.shape != b.shape:
        return False
    if a.dtype.is_floating_point or b.dtype.is_floating_point:
        a, b = a.double(), b.double()
        nan = a.isnan() & b.isnan()
        eq = a == b
        if signed_zero:
            eq = eq & (torch.signbit(a) == torch.signbit(b))
        return bool((eq | nan).all())
    return bool(torch.equal(a, b))


def mismatch_report(a: torch.Tensor, b: torch.Tensor, limit: int = 5) -> str:
    a, b = a.double().flatten(), b.double().flatten()
    bad = ~((a == b) | (a.isnan() & b.isnan()))
    idx = bad.nonzero().flatten()[:limit]
    return f"{int(bad.sum())} of {a.numel()} differ; first: " + ", ".join(f"[{int(i)}] {a[i].item()} vs {b[i].item()}" for i in idx)


def quantize_case(case, device, quantized_dtype=None):
    """Quantize the case's input with its parameters through fastforward_amd; returns the QuantizedTensor."""
    x = case_input(case).to(device)
    gran = granularity_of(case["granularity"])
    return ff.quantization.affine.quantize_per_granularity(
        x, to_device(case["scale"], device), to_device(case["offset"], device), gran, case["num_bits"], quantized_dtype
    )
