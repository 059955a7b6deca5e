"""Parity checks against the golden fixtures, written once and run against either backend.

tests/test_oracle_golden.py runs them with the C oracle on CPU tensors (this is what pins the
oracle to the reference); tests/test_parity_gpu.py runs the very same checks with the HIP library
on cuda tensors.
"""

from __future__ import annotations

import torch

import fastforward_amd as ff

from conftest import golden
from fastforward_amd import ops
from helpers import case_input, granularity_of, mismatch_report, quantize_case, same_with_nan, to_device


def check_known_answers(device):
    g1 = golden("g1_known_answers.pt")
    for name in ("symmetric_2bit", "asymmetric_2bit"):
        c = g1[name]
        q = ff.quantization.affine.quantize_per_tensor(c["data"].to(device), to_device(c["scale"], device), to_device(c["offset"], device), c["num_bits"])
        assert same_with_nan(q.raw_data.cpu(), c["codes"]), name
        assert same_with_nan(q.dequantize().cpu(), c["dequantized"]), name
    # the literal vectors written in the reference's tests
    assert torch.equal(g1["symmetric_2bit"]["codes"], g1["symmetric_2bit"]["expected_codes_in_reference_test"])
    assert torch.equal(g1["asymmetric_2bit"]["codes"][0], g1["asymmetric_2bit"]["expected_row_in_reference_test"])
    # one-sided: offset 8 for 4 bits (tests/nn/test_linear_quantizer.py:400-418)
    c = g1["one_sided_4bit"]
    quantizer = ff.nn.LinearQuantizer(4, symmetric=True, allow_one_sided=True, device=device)
    quantizer.quantization_range = (c["range_min"].to(device), c["range_max"].to(device))
    assert torch.equal(quantizer.offset.cpu(), torch.tensor([8.0]))
    assert same_with_nan(quantizer.scale.detach().cpu(), c["scale"])
    q = quantizer(c["data"].to(device))
    assert same_with_nan(q.raw_data.cpu(), c["codes"]) and same_with_nan(q.dequantize().cpu(), c["dequantized"])


def check_edges(device):
    for c in golden("g2_edges.pt"):
        qdt = None if c["quantized_dtype"] is None else getattr(torch, c["quantized_dtype"])
        q = quantize_case(c, device, qdt)
        assert q.raw_data.dtype == c["codes"].dtype, c["name"]
        assert same_with_nan(q.raw_data.cpu(), c["codes"]), f'{c["name"]}: {mismatch_report(q.raw_data.cpu(), c["codes"])}'
        assert same_with_nan(q.dequantize().cpu(), c["dequantized"]), f'{c["name"]}: {mismatch_report(q.dequantize().cpu(), c["dequantized"])}'


def check_sweeps(device, name_filter=None):
    n = 0
    for c in golden("g3_sweeps.pt"):
        if name_filter and not name_filter(c["name"]):
            continue
        n += 1
        x_dtype = getattr(torch, c["dtype"])
        for qdt in (None, torch.int8):
            q = quantize_case(c, device, qdt)
            assert q.raw_data.dtype == (qdt or x_dtype)
            got = q.raw_data.cpu()
            assert same_with_nan(got.to(torch.int8) if qdt is None else got, c["codes"]), f'{c["name"]} [{qdt}]: {mismatch_report(got, c["codes"])}'
            if c["dequantized"] is not None:
                deq = q.dequantize().cpu()
                assert deq.dtype == x_dtype
                # an integer container cannot hold the -0.0 a float container keeps, so the sign of a
                # zero result differs between the two containers in the reference as well
                assert same_with_nan(deq, c["dequantized"], signed_zero=qdt is None), f'{c["name"]} [{qdt}] dequantize: {mismatch_report(deq, c["dequantized"])}'
    assert n > 0


def check_dtype_sweep(device):
    cases = golden("g3_dtype_sweep.pt")
    assert len(cases) > 300
    for c in cases:
        qdt = getattr(torch, c["quantized_dtype"])
        q = quantize_case(c, device, qdt)
        got = q.raw_data.cpu()
        assert got.dtype == c["codes"].dtype
        assert same_with_nan(got, c["codes"]), f'{c["name"]}: {mismatch_report(got, c["codes"])}'
        deq = q.dequantize().cpu()
        assert deq.dtype == c["dequantized"].dtype, c["name"]
        assert same_with_nan(deq, c["dequantized"]), f'{c["name"]} dequantize: {mismatch_report(deq, c["dequantized"])}'


def check_ranges(device):
    for c in golden("g4_ranges.pt"):
        scale, offset = ff.quantization.affine.parameters_for_range(
            c["min"].to(device), c["max"].to(device), c["num_bits"], symmetric=c["symmetric"], allow_one_sided=c["allow_one_sided"]
        )
        assert same_with_nan(scale.cpu(), c["scale"]), (c, scale)
        assert (offset is None) == (c["offset"] is None), c
        if offset is not None:
            assert same_with_nan(offset.cpu(), c["offset"]), (c, offset)


def check_running_minmax(device, sync_free=False):
    for c in golden("g5_minmax.pt"):
        quantizer = ff.nn.LinearQuantizer(c["num_bits"], symmetric=c["symmetric"], granularity=granularity_of(c["granularity"]), device=device)
        model = torch.nn.ModuleList([quantizer])
        with ff.estimate_ranges(model, ff.range_setting.running_minmax, sync_free=sync_free):
            for batch, expected in zip(c["batches"], c["codes_per_step"]):
                got = quantizer(batch.to(device)).raw_data.cpu()
                assert same_with_nan(got, expected), mismatch_report(got, expected)
        assert same_with_nan(quantizer.scale.detach().cpu(), c["scale"])
        if c["offset"] is not None:
            assert same_with_nan(quantizer.offset.detach().cpu(), c["offset"])


def check_int4(device):
    c = golden("g8_int4.pt")
    gran = ff.PerBlock(block_dims=1, block_sizes=128, per_channel_dims=0)
    q = ff.quantization.affine.quantize_per_granularity(c["weight"].to(device), c["scale"].to(device), to_device(c["offset"], device), gran, 4, torch.int8)
    assert torch.equal(q.raw_data.cpu(), c["codes"])
    assert same_with_nan(q.dequantize().cpu(), c["dequantized"])
    packed = ops.pack_int4(q.raw_data, block=32)
    assert torch.equal(packed.cpu().reshape(-1, 16), c["q4_0_nibbles_block32"])
    for block in (32, 128, 256):
        for dtype in (torch.int8, torch.bfloat16, torch.float32, torch.int32):
            codes = q.raw_data.to(dtype)
            assert torch.equal(ops.unpack_int4(ops.pack_int4(codes, block=block), codes.shape, dtype, block=block), codes)


def linear_tolerances(dtype):
    # the reference's own tolerances for half-precision results (tests/quantization/test_tiled_affine.py:43-55)
    return (1e-1, 1.3e-2) if dtype in (torch.bfloat16, torch.float16) else (1e-4, 1.3e-5)


def check_linear(device):
    for c in golden("g6_linear.pt"):
        dtype = c["x"].dtype
        lin = torch.nn.Linear(c["weight"].shape[1], c["weight"].shape[0], bias=c["bias"] is not None).to(dtype)
        with torch.no_grad():
            lin.weight.copy_(c["weight"])
            if c["bias"] is not None:
                lin.bias.copy_(c["bias"])
        model = torch.nn.Sequential(lin).to(device)
        ff.quantize_model(model)
        lin.weight_quantizer = ff.nn.LinearQuantizer(8, granularity=ff.PerChannel(0), device=device)
        lin.input_quantizer = ff.nn.LinearQuantizer(8, symmetric=False, device=device)
        x = c["x"].to(device)
        with ff.strict_quantization(False):
            with ff.estimate_ranges(model, ff.range_setting.running_minmax):
                model(x)
            y = model(x)
            xq, wq = lin.input_quantizer(x), lin.weight_quantizer(lin.weight)
        # integer side: bit-exact
        assert same_with_nan(lin.input_quantizer.scale.detach().cpu(), c["x_scale"])
        assert same_with_nan(lin.input_quantizer.offset.detach().cpu(), c["x_offset"])
        assert same_with_nan(lin.weight_quantizer.scale.detach().cpu(), c["w_scale"])
        assert torch.equal(xq.raw_data.cpu().to(torch.int8), c["x_codes"])
        assert torch.equal(wq.raw_data.cpu().to(torch.int8), c["w_codes"])
        # float side: the fused int8 kernel against the reference's bf16/fp32 eager output
        atol, rtol = linear_tolerances(dtype)
        torch.testing.assert_close(y.detach().cpu().float(), c["y"].float(), atol=atol, rtol=rtol)
        # ... and against the exact (float64) value of the same dequantized operands: 2^-7 relative
        # for bf16 (operand + output rounding), tight for fp32
        tol = 2.0**-7 if dtype == torch.bfloat16 else 1e-5
        torch.testing.assert_close(y.detach().cpu().float(), c["y_float64"], atol=tol, rtol=tol)
