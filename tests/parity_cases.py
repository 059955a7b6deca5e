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
    # fused A1+A7 / A7+A2 (one pass each) == the reference's codes, nibbles and dequantized values
    w, scale, offset = c["weight"].to(device), c["scale"].to(device), to_device(c["offset"], device)
    tile = gran.tile_size(w.shape)
    fused = ops.quantize_pack_int4(w, scale, tile, offset, block=32)
    assert torch.equal(fused.cpu().reshape(-1, 16), c["q4_0_nibbles_block32"])
    assert same_with_nan(ops.unpack_dequantize_int4(fused, scale, w.shape, tile, offset, block=32, output_dtype=w.dtype).cpu(), c["dequantized"])
    for block in (32, 128, 256):
        for dtype in (torch.bfloat16, torch.float32):
            x = c["weight"].to(dtype).to(device)
            for tl, sc, of in ((tile, scale, offset), (tuple(x.shape), scale[:1], None), ((1, x.shape[1]), scale[: x.shape[0]], offset[: x.shape[0]] if offset is not None else None)):
                packed = ops.quantize_pack_int4(x, sc, tl, of, block=block)
                assert torch.equal(packed, ops.pack_int4(ops.quantize_by_tile(x, sc, tl, 4, torch.int8, of), block=block))
                back = ops.unpack_dequantize_int4(packed, sc, x.shape, tl, of, block=block, output_dtype=dtype)
                assert same_with_nan(back.cpu(), ops.dequantize_by_tile(ops.unpack_int4(packed, x.shape, torch.int8, block=block), sc, tl, of, dtype).cpu())


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


def check_weight_only_linear(device):
    """Fixture G15: the reference's fallback.linear with a quantized weight and a plain bf16 input (strict off).

    Integer side bit-exact (scale, offset, codes). Float side: the kernel multiplies EXACTLY the reference's dequantized
    bf16 weight (A2 in the operand load) and accumulates in fp32; F.linear's own fp32 summation order is unspecified, so
    the stated tolerance is one rounding of the output: |y - y64| <= 2^-8 |y64| (half a bf16 ulp) + 1e-4 against the
    float64 value of the same operands, and one bf16 ulp against the reference's own bf16 output."""
    from helpers import granularity_of

    for c in golden("g15_weight_only_linear.pt"):
        lin = torch.nn.Linear(c["weight"].shape[1], c["weight"].shape[0], bias=c["bias"] is not None).to(torch.bfloat16)
        with torch.no_grad():
            lin.weight.copy_(c["weight"])
            if c["bias"] is not None:
                lin.bias.copy_(c["bias"])
        model = torch.nn.Sequential(lin).to(device)
        ff.quantize_model(model)
        lin.weight_quantizer = ff.nn.LinearQuantizer(c["num_bits"], granularity=granularity_of(c["granularity"]), symmetric=c["symmetric"],
                                                     quantized_dtype=torch.int8, device=device)
        x = c["x"].to(device)
        with torch.no_grad(), ff.strict_quantization(False):
            with ff.estimate_ranges(model, ff.range_setting.running_minmax):
                model(x)
            wq = lin.weight_quantizer(lin.weight)
            with ff.fused_linear.weight_only_kernel(False):  # the A/B arm: nobody claims it, the reference's path runs (A2 + F.linear)
                assert ff.dispatcher.dispatch("linear", input=x, weight=wq, bias=lin.bias) is None
                y_reference_path = model(x)
            # the hand-written bf16 x weight-code GEMM: the default at every token count since round 4
            assert ff.dispatcher.dispatch("linear", input=x, weight=wq, bias=lin.bias) is ff.fused_linear.fused_linear_weight_only, c["name"]
            y = model(x)
        torch.testing.assert_close(y_reference_path.detach().cpu().float(), c["y"].float(), rtol=2.0**-7, atol=2e-4)
        assert same_with_nan(lin.weight_quantizer.scale.detach().cpu(), c["w_scale"]), c["name"]
        if c["w_offset"] is not None:
            assert same_with_nan(lin.weight_quantizer.offset.detach().cpu(), c["w_offset"]), c["name"]
        assert torch.equal(wq.raw_data.cpu(), c["w_codes"]), c["name"]
        assert y.dtype == torch.bfloat16 and y.shape == c["y"].shape
        got = y.detach().cpu().float()
        torch.testing.assert_close(got, c["y_float64"], rtol=2.0**-8, atol=1e-4, msg=lambda m: f'{c["name"]}: {m}')
        torch.testing.assert_close(got, c["y"].float(), rtol=2.0**-7, atol=2e-4, msg=lambda m: f'{c["name"]} vs reference output: {m}')


def check_linear_large(device):
    """Fixture G18: the reference's W8A8 QuantizedLinear at a size the 256 x 256-tile persistent GEMM takes (2048 tokens,
    N = 2048, K = 512). Parameters and all codes (through their row sums) bit-exact; the output within the reference's own
    half-precision tolerance of its bf16 eager result and within 2^-7 of the float64 value of the same operands."""
    from datagen import make_data

    c = golden("g18_linear_large.pt")
    x = make_data(c["x_seed"], tuple(c["x_shape"]), torch.bfloat16, "normal").to(device)
    w = (make_data(c["w_seed"], tuple(c["w_shape"]), torch.float32, "normal") * c["w_factor"]).to(torch.bfloat16)
    lin = torch.nn.Linear(w.shape[1], w.shape[0], bias=False).to(torch.bfloat16)
    with torch.no_grad():
        lin.weight.copy_(w)
    model = torch.nn.Sequential(lin).to(device)
    ff.quantize_model(model)
    lin.weight_quantizer = ff.nn.LinearQuantizer(8, granularity=ff.PerChannel(0), quantized_dtype=torch.int8, device=device)
    lin.input_quantizer = ff.nn.LinearQuantizer(8, symmetric=False, quantized_dtype=torch.int8, device=device)
    with torch.no_grad(), ff.strict_quantization(False):
        with ff.estimate_ranges(model, ff.range_setting.running_minmax):
            model(x)
        y = model(x)
        xq, wq = lin.input_quantizer(x), lin.weight_quantizer(lin.weight)
    assert same_with_nan(lin.input_quantizer.scale.detach().cpu(), c["x_scale"]) and same_with_nan(lin.input_quantizer.offset.detach().cpu(), c["x_offset"])
    assert same_with_nan(lin.weight_quantizer.scale.detach().cpu(), c["w_scale"])
    xc, wc = xq.raw_data.reshape(-1, w.shape[1]).cpu().to(torch.int64), wq.raw_data.cpu().to(torch.int64)
    assert torch.equal(xc.sum(1), c["x_code_row_sums"]) and int(xc.abs().sum()) == c["x_code_abs_sum"]
    assert torch.equal(wc.sum(1), c["w_code_row_sums"]) and int(wc.abs().sum()) == c["w_code_abs_sum"]
    got = y.detach().reshape(-1, w.shape[0]).cpu()[c["rows"]].double()
    atol, rtol = linear_tolerances(torch.bfloat16)
    torch.testing.assert_close(got.float(), c["y_rows"].float(), atol=atol, rtol=rtol)  # the reference's bf16 result, its own tolerance
    # The kernel contracts the integer codes exactly and scales once; the reference rounds both dequantized operands to
    # bf16 first. (a) Against the float64 value of the EXACT operands the output is one bf16 rounding away:
    x_exact = (xc.double()[c["rows"]] + torch.round(c["x_offset"].double())) * c["x_scale"].double()
    w_exact = wc.double() * c["w_scale"].double()[:, None]
    y_exact = x_exact @ w_exact.t()
    one_rounding = 2.0**-8 * y_exact.abs() + 1e-4
    if ops._native.library().is_device:
        assert bool(((got - y_exact).abs() <= one_rounding).all()), float((got - y_exact).abs().max())
    else:  # the oracle restates the reference (operands rounded to bf16 first): one rounding away from THAT value
        assert bool(((got - c["y_rows_float64"].double()).abs() <= one_rounding).all())
    # (b) against the float64 value of the reference's bf16-ROUNDED operands (the fixture's y_rows_float64) the distance is
    # bounded by the rounding of the operands themselves: each product is off by at most 2^-8 relative (two bf16 roundings)
    operand_rounding = 2.0**-8 * (x_exact.abs() @ w_exact.abs().t())
    assert bool(((got - c["y_rows_float64"].double()).abs() <= operand_rounding + 2.0**-8 * y_exact.abs() + 1e-4).all())


def _ulps_bf16(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """Distance in bf16 units-in-the-last-place between two bf16 tensors (finite values)."""
    def key(t):
        bits = t.contiguous().view(torch.int16).to(torch.int32) & 0xFFFF
        return torch.where(bits >= 0x8000, 0x8000 - bits, bits)  # monotone in the value
    return (key(a) - key(b)).abs()


def check_producers(device):
    """Fixture G10: the reference's RMSNorm (behind a residual add), SiLU*up and rotary embedding in
    bf16, each followed by a static per-tensor 8-bit quantizer.

    Contract (include/ffq.h): adds, multiplies and the rotary embedding are exact; the normalised value
    (two bf16 roundings: up to 2 ulp) and silu (1 ulp) may differ from the reference's CPU result on rare elements (fp32 summation
    order of the mean, last bit of rsqrt / exp differ between platforms); codes are exactly A1 of the
    value the call produced, hence equal to the reference's codes wherever that value is equal.
    """
    g = golden("g10_producers.pt")
    for name, c in g.items():
        if not name.startswith("rmsnorm"):
            continue
        qp = c["quantized"]
        quantizers = [(qp["scale"].to(device), qp["offset"].to(device))] * 2 + [(qp["scale"].to(device) * 2, None)]
        total, norm, codes = ops.add_rmsnorm_quantize(c["x"].to(device), c["delta"].to(device), c["weight"].to(device), c["eps"], quantizers, want_norm=True)
        assert torch.equal(total.cpu(), c["sum"]), name
        ulps = _ulps_bf16(norm.cpu(), c["normalised"])
        assert int(ulps.max()) <= 2 and float((ulps > 0).float().mean()) < 0.01, f"{name}: {int(ulps.max())} ulp, {float((ulps > 0).float().mean()):.4f} differ"
        same = ulps == 0
        assert torch.equal(codes[0].cpu()[same], qp["codes"][same]), name
        assert torch.equal(codes[0], codes[1]), name
        # the codes are A1 of the value this call produced
        for (s, o), got in zip(quantizers, codes):
            want = ops.quantize_by_tile(norm, s, norm.shape, 8, torch.int8, o)
            assert torch.equal(got, want), name
        # without the residual: sum is the input itself, nothing else changes
        total2, norm2, codes2 = ops.add_rmsnorm_quantize(total, None, c["weight"].to(device), c["eps"], quantizers[:1], want_norm=True)
        assert total2.data_ptr() == total.data_ptr() and torch.equal(norm2, norm) and torch.equal(codes2[0], codes[0]), name
        # no quantizers, no normalised output requested
        total3, norm3, codes3 = ops.add_rmsnorm_quantize(c["x"].to(device), c["delta"].to(device), c["weight"].to(device), c["eps"])
        assert torch.equal(total3, total) and norm3 is None and codes3 == []

    c = g["silu_mul"]
    qp = c["quantized"]
    quantizers = [(qp["scale"].to(device), qp["offset"].to(device))]
    product, codes = ops.silu_mul_quantize(c["gate"].to(device), c["up"].to(device), quantizers, want_product=True)
    ulps = _ulps_bf16(product.cpu(), c["product"])
    assert int(ulps.max()) <= 1 and float((ulps > 0).float().mean()) < 0.01, f"silu_mul: {int(ulps.max())} ulp"
    same = ulps == 0
    assert torch.equal(codes[0].cpu()[same], qp["codes"][same])
    assert torch.equal(codes[0], ops.quantize_by_tile(product, quantizers[0][0], product.shape, 8, torch.int8, quantizers[0][1]))
    none, codes_only = ops.silu_mul_quantize(c["gate"].to(device), c["up"].to(device), quantizers)
    assert none is None and torch.equal(codes_only[0], codes[0])

    c = g["rope"]
    q, k = c["q"].clone().to(device), c["k"].clone().to(device)
    ops.rope_(q, k, c["cos"].to(device), c["sin"].to(device), c["head_dim"])
    assert torch.equal(q.cpu(), c["q_rotated"]) and torch.equal(k.cpu(), c["k_rotated"])

    # argument errors follow the other entry points
    import pytest

    x = torch.zeros(2, 24, dtype=torch.bfloat16, device=device)
    with pytest.raises(NotImplementedError):
        ops.add_rmsnorm_quantize(x, None, torch.ones(24, dtype=torch.bfloat16, device=device), 1e-5)  # 24 % 16 != 0
    with pytest.raises(NotImplementedError):
        ops.add_rmsnorm_quantize(x.float(), None, torch.ones(24, device=device), 1e-5)
    with pytest.raises(RuntimeError):
        ops.silu_mul_quantize(x, x[:1])


def check_matmul_family(device):
    """mm / matmul / bmm on quantized operands (reference fallback: src/fastforward/_gen/fallback.py:699-798 — dequantize both,
    float op, output quantizer): the dispatcher takes the int8 contraction where the GEMM covers the granularities and the
    float fallback elsewhere; both must agree with the float64 product of the dequantized operands like the linear does."""
    torch.manual_seed(41)
    F_ = ff.nn.functional

    def quantized(t, **kw):
        q = ff.nn.LinearQuantizer(8, quantized_dtype=torch.int8, device=device, **kw)
        with ff.estimate_ranges(q, ff.range_setting.running_minmax):
            return q(t.to(device))

    def close(got, a, b, op):
        want = op(a.dequantize().detach().double().cpu(), b.dequantize().detach().double().cpu())
        assert got.dtype == torch.bfloat16 and got.shape == want.shape
        torch.testing.assert_close(got.cpu().double(), want, atol=2.0**-7 * float(want.detach().abs().max()), rtol=2.0**-7)

    x3 = torch.randn(5, 24, 64).to(torch.bfloat16)
    x2 = torch.randn(40, 64).to(torch.bfloat16)
    w = (torch.randn(64, 48) * 0.1).to(torch.bfloat16)
    with ff.strict_quantization(False):
        for x in (x3, x2):
            for wkw in ({}, {"granularity": ff.PerChannel(-1)}, {"granularity": ff.PerChannel(-1), "symmetric": False}):
                a, b = quantized(x, symmetric=False), quantized(w, **wkw)
                assert ff.dispatcher.dispatch("matmul", input=a, other=b) is not None
                close(F_.matmul(a, b), a, b, torch.matmul)
                close(torch.matmul(a, b), a, b, torch.matmul)          # the __torch_function__ route
                if x.dim() == 2:
                    close(F_.mm(a, b), a, b, torch.mm)
        # per-row parameters on the RIGHT operand (one pair per k): not a GEMM epilogue — float fallback, same answer
        a, b = quantized(x2, symmetric=False), quantized(w, granularity=ff.PerChannel(0))
        assert ff.dispatcher.dispatch("mm", input=a, mat2=b) is None
        close(F_.mm(a, b), a, b, torch.mm)
        # bmm: one parameter pair per operand
        a, b = quantized(torch.randn(3, 10, 32).to(torch.bfloat16), symmetric=False), quantized(torch.randn(3, 32, 20).to(torch.bfloat16))
        assert ff.dispatcher.dispatch("bmm", input=a, mat2=b) is not None
        close(F_.bmm(a, b), a, b, torch.bmm)
        # ... as ONE launch for the whole batch (round 4; a Python loop of launches before): every pair equals its own mm call bit for bit,
        # asymmetric on both sides, ragged sizes, and the output quantizer rides in the batched launch's epilogue
        a, b = quantized(torch.randn(7, 150, 96).to(torch.bfloat16), symmetric=False), quantized((torch.randn(7, 96, 70) * 0.3).to(torch.bfloat16), symmetric=False)
        whole = F_.bmm(a, b)
        close(whole, a, b, torch.bmm)
        pa, pb = a.quantization_context.quantization_params, b.quantization_context.quantization_params
        for i in (0, 3, 6):
            one = ff.ops.linear_w8a8(a.raw_data[i], b.raw_data[i].t().contiguous(), pa.scale, pa.offset, pb.scale, pb.offset, out_dtype=torch.bfloat16)
            assert torch.equal(whole[i], one)
        bq = ff.nn.LinearQuantizer(8, symmetric=False, quantized_dtype=torch.int8, device=device)
        with ff.estimate_ranges(bq, ff.range_setting.running_minmax):
            bq(whole)
        fused_codes = F_.bmm(a, b, output_quantizer=bq)
        assert isinstance(fused_codes, ff.QuantizedTensor) and torch.equal(fused_codes.raw_data, bq(whole).raw_data)
        # matmul with an N-d RIGHT operand (attention-shaped [B, H, S, D] x [B, H, D, S]): the batched launch, no broadcasting
        a4, b4 = quantized(torch.randn(2, 3, 20, 32).to(torch.bfloat16), symmetric=False), quantized(torch.randn(2, 3, 32, 24).to(torch.bfloat16))
        assert ff.dispatcher.dispatch("matmul", input=a4, other=b4) is not None
        close(F_.matmul(a4, b4), a4, b4, torch.matmul)
        close(torch.matmul(a4, b4), a4, b4, torch.matmul)
        assert ff.dispatcher.dispatch("matmul", input=a4, other=quantized(torch.randn(3, 32, 24).to(torch.bfloat16))) is None  # broadcasting: the float fallback
        # an output quantizer is applied to the result
        out_q = ff.nn.LinearQuantizer(8, symmetric=False, quantized_dtype=torch.int8, device=device)
        a, b = quantized(x2, symmetric=False), quantized(w)
        with ff.estimate_ranges(out_q, ff.range_setting.running_minmax):
            y = F_.mm(a, b, output_quantizer=out_q)
        assert isinstance(y, ff.QuantizedTensor)
    import pytest

    with pytest.raises(ff.exceptions.QuantizationError):
        F_.mm(quantized(x2), quantized(w), strict_quantization=True)  # strict: an output quantizer is required


def check_rowsum_fusion(device):
    """ABI version 3: weight codes + their row sums in one pass, and the GEMM entry points that take them — the same codes
    as A1, the exact integer sums, and bit-identical GEMM results with and without the sums handed in."""
    import pytest

    torch.manual_seed(31)
    for rows, cols, with_offset in ((256, 2048, False), (128, 1024, True), (384, 3072, True)):
        w = (torch.randn(rows, cols) * 0.02).to(torch.bfloat16).to(device)
        scale = (w.float().abs().amax(1) / 127).clamp_min(1e-8)
        offset = (torch.randn(rows, device=device) * 3) if with_offset else None
        got = ops.quantize_rows_rowsum(w, scale, offset)
        assert got is not None
        codes, rowsum = got
        assert torch.equal(codes, ops.quantize_by_tile(w, scale, (1, cols), 8, torch.int8, offset))
        assert torch.equal(rowsum.cpu(), codes.cpu().int().sum(1).to(torch.int32))
        x = torch.randint(-128, 128, (3, 40, cols), dtype=torch.int8, device=device)
        xs, xo = torch.tensor([0.02], device=device), torch.tensor([5.0], device=device)
        w_off = None if offset is None else offset
        a = ops.linear_w8a8(x, codes, xs, xo, scale, w_off)
        b = ops.linear_w8a8(x, codes, xs, xo, scale, w_off, w_rowsum=rowsum)
        assert torch.equal(a, b)
        if not ops._native.library().is_device:  # the oracle is the checker: it refuses sums that are not the codes' sums
            with pytest.raises(ValueError, match="row sum"):
                ops.linear_w8a8(x, codes, xs, xo, scale, w_off, w_rowsum=rowsum + 1)
    # the one-launch MLP front half with the sums of both weight matrices
    k, n = 1024, 256
    g = (torch.randn(n, k) * 0.02).to(torch.bfloat16).to(device)
    u = (torch.randn(n, k) * 0.02).to(torch.bfloat16).to(device)
    gs, us = g.float().abs().amax(1) / 127, u.float().abs().amax(1) / 127
    (gc, grs), (uc, urs) = ops.quantize_rows_rowsum(g, gs, None), ops.quantize_rows_rowsum(u, us, None)
    x = torch.randint(-128, 128, (2, 96, k), dtype=torch.int8, device=device)
    xs, xo = torch.tensor([0.03], device=device), torch.tensor([-7.0], device=device)
    ds, do = torch.tensor([0.01], device=device), torch.tensor([-30.0], device=device)
    plain = ops.mlp_gate_up_w8a8(x, gc, uc, xs, xo, gs, us, ds, do)
    with_sums = ops.mlp_gate_up_w8a8(x, gc, uc, xs, xo, gs, us, ds, do, gate_rowsum=grs, up_rowsum=urs)
    assert plain is not None and torch.equal(plain, with_sums)
    # outside the one-pass kernel's range: the caller takes quantize_by_tile
    assert ops.quantize_rows_rowsum(torch.zeros(4, 1000, dtype=torch.bfloat16, device=device), torch.ones(4, device=device), None) is None
    assert ops.quantize_rows_rowsum(torch.zeros(4, 1024, device=device), torch.ones(4, device=device), None) is None


def attention_reference64(q, k, v, head_dim, causal):
    """softmax(q k^T / sqrt(d) [+ causal mask]) v in float64 on the bf16 inputs: [batch, seq, heads * head_dim]."""
    b, s_, _ = q.shape
    heads, kv_heads = q.shape[2] // head_dim, k.shape[2] // head_dim
    qs = q.double().view(b, s_, heads, head_dim).transpose(1, 2)
    ks = k.double().view(b, s_, kv_heads, head_dim).transpose(1, 2).repeat_interleave(heads // kv_heads, dim=1)
    vs = v.double().view(b, s_, kv_heads, head_dim).transpose(1, 2).repeat_interleave(heads // kv_heads, dim=1)
    w = qs @ ks.transpose(2, 3) * head_dim**-0.5
    if causal:
        w = w.masked_fill(torch.ones(s_, s_, dtype=torch.bool, device=q.device).triu(1), float("-inf"))
    return (torch.softmax(w, dim=-1) @ vs).transpose(1, 2).reshape(b, s_, -1)


# Attention is a floating-point kernel. The reference's eager chain rounds scores, scaled scores, probabilities and
# the context to bf16 (attention.py:60-88) — with |score| up to ~30 in G14 a bf16 score carries an error of 2^-4, so
# the reference itself sits up to 7e-2 away from the float64 value of the same bf16 inputs. The flash-style launch keeps
# scores and the context sum in fp32 and rounds only the un-normalised probabilities and the result: it must stay
# within ATTENTION_ATOL of the float64 value (|context| <= ~4.5 here: one bf16 ulp of the largest outputs) and may
# never be less accurate than the reference is.
ATTENTION_ATOL = 2.0**-6


def check_attention(device, exact_chain=False):
    """G14: the reference's attention chain + o_proj input quantizer. `exact_chain`: the implementation follows the
    eager chain op by op (the oracle), so it must reproduce the fixture on nearly every element."""
    g = golden("g14_attention.pt")
    for name, c in g.items():
        q, k, v = c["q"].to(device), c["k"].to(device), c["v"].to(device)
        qp = c["quantized"]
        quantizer = (qp["scale"].to(device), qp["offset"].to(device))
        ctx, codes = ops.attention(q, k, v, c["head_dim"], causal=c["causal"], quantizer=quantizer)
        want64 = attention_reference64(c["q"], c["k"], c["v"], c["head_dim"], c["causal"])
        err_got = float((ctx.cpu().double() - want64).abs().max())
        err_ref = float((c["context"].double() - want64).abs().max())
        ulps = _ulps_bf16(ctx.cpu(), c["context"])
        if not exact_chain:
            assert err_got <= ATTENTION_ATOL and err_got <= err_ref + 2.0**-9, f"{name}: {err_got:.3e} from float64, reference {err_ref:.3e}"
        else:
            # (fp32 summation order inside the two matmuls is the only freedom: a rare 1-ulp flip of a score or a probability)
            worst = float((ctx.cpu().float() - c["context"].float()).abs().max())
            assert worst <= 2.0**-7 and float((ulps > 0).float().mean()) < 0.005, f"{name}: {worst:.3e} off, {float((ulps > 0).float().mean()):.4f} differ"
        # the codes are exactly A1 of the context this call produced; where the context equals the reference's, so do the codes
        assert torch.equal(codes, ops.quantize_by_tile(ctx, quantizer[0], ctx.shape, 8, torch.int8, quantizer[1])), name
        same = ulps == 0
        assert torch.equal(codes.cpu()[same], qp["codes"][same]), name
        off = (codes.cpu().int() - qp["codes"].int()).abs()
        bound = int((err_got + err_ref) / float(qp["scale"])) + 1
        assert int(off.max()) <= bound, f"{name}: codes off by {int(off.max())} > {bound}"
        # outputs are optional
        only_ctx, none = ops.attention(q, k, v, c["head_dim"], causal=c["causal"])
        assert none is None and torch.equal(only_ctx, ctx), name
        none, only_codes = ops.attention(q, k, v, c["head_dim"], causal=c["causal"], quantizer=quantizer, want_context=False)
        assert none is None and torch.equal(only_codes, codes), name
    import pytest

    x = torch.zeros(1, 64, 256, dtype=torch.bfloat16, device=device)
    with pytest.raises(NotImplementedError):
        ops.attention(x, x, x, 64)            # head_dim 64
    with pytest.raises(NotImplementedError):
        ops.attention(x[:, :48], x[:, :48], x[:, :48], 128)   # seq 48
    with pytest.raises(RuntimeError):
        ops.attention(x, x[:, :32], x[:, :32], 128)


def _backward_terms64(c):
    """Per-element gradient terms of fixture case `c` in float64 (an independent statement of the formulas),
    as rows per tile: (dscale terms, doffset terms)."""
    from fastforward_amd.quantization.tiled_tensor import tiles_to_rows

    tile = torch.Size(c["tile"])
    s = c["scale"].double()[:, None]
    o = (c["offset"] if c["offset"] is not None else torch.zeros_like(c["scale"])).double()[:, None]
    x, g = tiles_to_rows(c["data"].float(), tile), tiles_to_rows(c["grad"].double(), tile)
    lo = -(2.0 ** (c["num_bits"] - 1))
    hi = -lo - 1
    o = torch.round(o)  # _infer_offset rounds the offset once, for every use (reference :140-141)
    u = (x / c["scale"][:, None] - o.float()).double()  # the fp32 value the kernel sees
    q = torch.round(u)
    below, clip = q < lo, (q < lo) | (q > hi)
    ds = torch.where(clip, torch.where(below, lo, hi) + o, q - u) * g
    do = torch.where(clip, s * g, torch.zeros_like(g))
    return ds, do


def check_backward(device):
    """Fixture G11: fastforward::quantize_by_tile_backward of the reference. dinput is exact; the per-tile sums
    are fp32 sums in an implementation-defined order: |got - want| <= 4e-6 * sum|terms| (about 30 ulp of the
    largest partial sum), the bound the reference's own fp32 summation satisfies against float64."""
    for c in golden("g11_backward.pt"):
        offset = None if c["offset"] is None else c["offset"].to(device)
        dinput, dscale, doffset = ops.quantize_by_tile_backward(c["data"].to(device), c["grad"].to(device), c["scale"].to(device), c["tile"], float(c["num_bits"]), offset)
        assert dinput.dtype == c["dinput"].dtype and same_with_nan(dinput.cpu(), c["dinput"]), c["name"]
        ds64, do64 = _backward_terms64(c)
        tol_s = 4e-6 * ds64.abs().sum(1) + 1e-30
        assert dscale.shape == c["dscale"].shape and bool(((dscale.cpu().double() - c["dscale"].double()).abs() <= tol_s).all()), c["name"]
        assert bool(((dscale.cpu().double() - ds64.sum(1)).abs() <= tol_s).all()), c["name"]
        if c["offset"] is None:
            assert doffset.numel() == 0, c["name"]
        else:
            tol_o = 4e-6 * do64.abs().sum(1) + 1e-30
            assert bool(((doffset.cpu().double() - c["doffset"].double()).abs() <= tol_o).all()), c["name"]
            assert bool(((doffset.cpu().double() - do64.sum(1)).abs() <= tol_o).all()), c["name"]


def check_mse_grid(device):
    """Fixture G12: the reference's mse_grid estimator. The search grid is exact; every candidate's cumulative error
    agrees within the summation tolerance of a mean (fp32: 1e-5 relative; bf16 results: one bf16 ulp per step); the
    selected range is the reference's, or — where two candidates tie within that tolerance — one whose error is."""
    from fastforward_amd.range_setting.min_error import _MinAvgErrorGridEstimator

    for c in golden("g12_mse_grid.pt"):
        quantizer = ff.nn.LinearQuantizer(c["num_bits"], granularity=granularity_of(c["granularity"]), symmetric=c["symmetric"], device=device)
        with ff.estimate_ranges(quantizer, ff.range_setting.mse_grid, num_candidates=c["num_candidates"]):
            est = next(o for o in quantizer.overrides if isinstance(o, _MinAvgErrorGridEstimator))
            for x in c["batches"]:
                quantizer(x.to(device))
            # every tiling is covered by a kernel of the library (strided channels / N-d tiles: the by-tile kernel)
            assert est.used_fused_kernel, c["name"]
            assert same_with_nan(est.min_threshold.cpu(), c["min_threshold"]) and same_with_nan(est.max_threshold.cpu(), c["max_threshold"]), c["name"]
            got, want = est.cumulative_error.cpu().double(), c["cumulative_error"].double()
            rtol = 2e-2 if c["cumulative_error"].dtype == torch.bfloat16 else 1e-5
            assert torch.allclose(got, want, rtol=rtol, atol=1e-12), f'{c["name"]}: {float(((got - want).abs() / want.abs().clamp_min(1e-30)).max())}'
        # the chosen candidate: identical, or a tie within tolerance
        best_ref = c["cumulative_error"].double().min(dim=0)
        chosen = est.cumulative_error.cpu().double().min(dim=0).indices
        err_of_chosen = c["cumulative_error"].double().gather(0, chosen[None, :])[0]
        assert bool((err_of_chosen <= best_ref.values * (1 + 2 * rtol) + 1e-12).all()), c["name"]
        same_choice = chosen == best_ref.indices
        if bool(same_choice.all()):
            assert same_with_nan(quantizer.scale.detach().cpu(), c["scale"]), c["name"]
            if c["offset"] is not None:
                assert same_with_nan(quantizer.offset.detach().cpu(), c["offset"]), c["name"]


def check_smoothed_minmax(device):
    """Fixture G16: the reference's SmoothedMinMax estimator (range_setting/minmax.py:67-92) over five batches — codes of
    every step, parameters after every step and at the end, bit for bit."""
    n = 0
    for c in golden("g16_smoothed_minmax.pt"):
        quantizer = ff.nn.LinearQuantizer(c["num_bits"], symmetric=c["symmetric"], granularity=granularity_of(c["granularity"]), device=device)
        with ff.estimate_ranges(torch.nn.ModuleList([quantizer]), ff.range_setting.smoothed_minmax, gamma=c["gamma"]):
            for batch, expected, (scale, offset) in zip(c["batches"], c["codes_per_step"], c["params_per_step"]):
                got = quantizer(batch.to(device)).raw_data.cpu()
                assert same_with_nan(got, expected), f'{c["granularity"]} gamma={c["gamma"]}: {mismatch_report(got, expected)}'
                assert same_with_nan(quantizer.scale.detach().cpu(), scale)
                if offset is not None:
                    assert same_with_nan(quantizer.offset.detach().cpu(), offset)
        assert same_with_nan(quantizer.scale.detach().cpu(), c["scale"])
        if c["offset"] is not None:
            assert same_with_nan(quantizer.offset.detach().cpu(), c["offset"])
        n += 1
    assert n == 36


def check_gguf_blocks(device):
    # Fixture G17: records produced by the reference's own pack_q4_0_blocks / pack_q8_0_blocks (_packing.py:23-79),
    # including out-of-range codes, -128 and scales that round / overflow / underflow in fp16 — byte for byte
    g17 = golden("g17_gguf_blocks.pt")
    assert torch.equal(ops.pack_q4_0_blocks(g17["codes4"].to(device), g17["scales"].to(device)).cpu(), g17["q4_0"])
    assert torch.equal(ops.pack_q8_0_blocks(g17["codes8"].to(device), g17["scales"].to(device)).cpu(), g17["q8_0"])

    """The reference's assertions for its GGUF block-32 packers (tests/export/stages/gguf/test_packing.py), re-expressed:
    record widths, positive fp16 scale, +8 nibble offset with low/high halves, llama.cpp's dequantization formulas
    reproduce scale * code, -128 is clipped to -127 in Q8_0."""
    import numpy as np

    def dequant_q4_0(b):
        d = b[:, :2].copy().view(np.float16).astype(np.float32).reshape(-1, 1)
        qs = b[:, 2:]
        return d * np.concatenate([(qs & 0x0F).astype(np.int16) - 8, (qs >> 4).astype(np.int16) - 8], axis=1).astype(np.float32)

    def dequant_q8_0(b):
        d = b[:, :2].copy().view(np.float16).astype(np.float32).reshape(-1, 1)
        return d * b[:, 2:].view(np.int8).astype(np.float32)

    codes = torch.zeros(2, 32, dtype=torch.int8)
    codes[0, 0], codes[0, 1], codes[1, :] = -8, 7, 3
    scales = torch.tensor([0.5, 2.0])
    packed = ops.pack_q4_0_blocks(codes.to(device), scales.to(device)).cpu()
    assert packed.shape == (2, 18) and packed.dtype == torch.uint8
    np.testing.assert_allclose(packed.numpy()[:, :2].copy().view(np.float16).astype(np.float32).reshape(-1), scales.numpy(), rtol=1e-3)
    assert int(packed[0, 2]) & 0x0F == 0 and int(packed[0, 2]) >> 4 == 8 and int(packed[0, 3]) & 0x0F == 15
    assert bool((packed[1, 2:] == (11 | (11 << 4))).all())
    rng = np.random.default_rng(0)
    c4 = torch.from_numpy(rng.integers(-8, 8, size=(16, 32)).astype(np.int8))
    s4 = torch.from_numpy(rng.uniform(0.05, 2.0, size=16).astype(np.float32))
    np.testing.assert_allclose(dequant_q4_0(ops.pack_q4_0_blocks(c4.to(device), s4.to(device)).cpu().numpy()), s4.numpy()[:, None] * c4.numpy().astype(np.float32), atol=1e-2)
    rng = np.random.default_rng(1)
    c8 = torch.from_numpy(rng.integers(-127, 128, size=(16, 32)).astype(np.int8))
    s8 = torch.from_numpy(rng.uniform(0.001, 0.5, size=16).astype(np.float32))
    np.testing.assert_allclose(dequant_q8_0(ops.pack_q8_0_blocks(c8.to(device), s8.to(device)).cpu().numpy()), s8.numpy()[:, None] * c8.numpy().astype(np.float32), atol=1e-2)
    clipped = ops.pack_q8_0_blocks(torch.full((1, 32), -128, dtype=torch.int8, device=device), torch.tensor([0.1], device=device)).cpu()
    assert clipped.shape == (1, 34) and int(clipped[:, 2:].numpy().view(np.int8).min()) == -127
    for fn, width in ((ops.pack_q4_0_blocks, 18), (ops.pack_q8_0_blocks, 34)):
        assert fn(torch.zeros(1, 32, dtype=torch.int8, device=device), torch.ones(1, device=device)).shape == (1, width)
    # the exact bytes, against a direct statement of the layout, on a block count that is not a multiple of the launch block
    n = 1000
    gen = torch.Generator().manual_seed(8)
    codes = torch.randint(-128, 128, (n, 32), generator=gen, dtype=torch.int8)
    scales = torch.rand(n, generator=gen) * 0.3 + 1e-3
    d = scales.to(torch.float16).view(torch.uint8).reshape(n, 2)
    q = (codes.to(torch.int16) + 8).clamp(0, 15).to(torch.uint8)
    assert torch.equal(ops.pack_q4_0_blocks(codes.to(device), scales.to(device)).cpu(), torch.cat([d, q[:, :16] | (q[:, 16:] << 4)], dim=1))
    assert torch.equal(ops.pack_q8_0_blocks(codes.to(device), scales.to(device)).cpu(), torch.cat([d, codes.clamp(-127, 127).view(torch.uint8)], dim=1))


def _gptq_granularity(name):
    return {"channel0": ff.PerChannel(0), "tensor": ff.PerTensor(), "group16": ff.PerBlock(block_dims=1, block_sizes=16, per_channel_dims=0),
            "channel1": ff.PerChannel(1)}[name]


def run_gptq_case(c, device, fused):
    from fastforward_amd.quantization.gptq import gptq

    layer = torch.nn.Linear(c["weight"].shape[1], c["weight"].shape[0], bias=False)
    with torch.no_grad():
        layer.weight.copy_(c["weight"])
    ff.quantize_model(layer)
    layer.to(device)
    layer.weight_quantizer = ff.nn.LinearQuantizer(c["num_bits"], granularity=_gptq_granularity(c["granularity"]), symmetric=c["symmetric"], device=device)
    dataset = [((a.to(device),), {}) for a in c["activations"]]
    with torch.no_grad(), ff.strict_quantization(False):
        gptq(layer, dataset, block_size=c["block_size"], actorder=c["actorder"], fused=fused)
    return layer


def check_gptq(device, exact):
    """Fixture G13: the reference's gptq(). On the CPU (same torch Cholesky / matmul as the reference ran) the result is
    the reference's weight bit for bit, through the one-launch block kernel and through the column loop; on the GPU
    the Hessian inverse comes from a different LAPACK, so the comparison there is fused kernel == column loop (same
    inputs, exact) and closeness to the fixture."""
    for c in golden("g13_gptq.pt"):
        fused = run_gptq_case(c, device, fused=True)
        loop = run_gptq_case(c, device, fused=False)
        assert same_with_nan(fused.weight.detach().cpu(), loop.weight.detach().cpu()), c["name"]
        assert same_with_nan(fused.weight_quantizer.scale.detach().cpu(), loop.weight_quantizer.scale.detach().cpu()), c["name"]
        got = fused.weight.detach().cpu()
        if exact:
            assert same_with_nan(got, c["result"]), f'{c["name"]}: {mismatch_report(got, c["result"])}'
            assert same_with_nan(fused.weight_quantizer.scale.detach().cpu(), c["scale"]), c["name"]
        elif c["granularity"] in ("channel0", "tensor", "channel1"):
            # a different rounding in the Hessian inverse can move a weight to a neighbouring grid point (grouped scales
            # are themselves recomputed from the error-corrected weights, so their grids move too: not compared)
            step = float(c["scale"].max())
            assert float((got - c["result"]).abs().max()) <= 2.01 * step and float((got != c["result"]).float().mean()) < 0.05, c["name"]
        if c["granularity"] in ("channel0", "tensor"):  # fixed per-row grids: every weight sits on its grid
            with ff.strict_quantization(False):
                assert same_with_nan(fused.weight_quantizer(fused.weight).dequantize().detach().cpu(), got), c["name"]
