"""Sibling quantizers whose parameters only the device can compare (include/ffq.h: ffq_quantize_by_tile_unless_same,
ffq_linear_w8a8_earlier; the `xq_up` of ffq_mlp_gate_up_w8a8_estimating): q_proj / k_proj / v_proj and gate_proj / up_proj
quantize the same hidden state with their own input quantizers (reference nn/linear.py:32-39), and while range estimators move
the parameters on every step the later quantizers' A1 launches run only where the parameters differ from the first one's.

Ground truth: every quantizer quantizing for itself (ops.quantize_by_tile) and every linear reading its own codes
(ops.linear_w8a8) — launches pinned elsewhere (fixtures G1-G3, tests/test_gemm_gpu.py); here only WHICH codes are read is at
stake, so every comparison is bit for bit. Unwritten codes are modelled by a constant fill that no quantizer would produce.
"""

import ctypes

import pytest
import torch

from fastforward_amd import _native, ops
from fastforward_amd.quantization.affine._memo import RECENT

pytestmark = pytest.mark.gpu
DEV = "cuda"
GARBAGE = 77


@pytest.fixture(autouse=True)
def _backend(hip_backend):
    yield


def _t(v):
    return None if v is None else torch.tensor([v], device=DEV, dtype=torch.float32)


PAIRS = {
    # name: ((scale, offset), (earlier scale, earlier offset), same?)
    "equal": ((0.031, -2.6), (0.031, -2.6), True),
    "offsets_round_alike": ((0.031, 3.2), (0.031, 2.9), True),
    "no_offset_and_one_that_rounds_to_zero": ((0.031, None), (0.031, 0.3), True),
    "tie_rounds_to_even": ((0.031, 2.5), (0.031, 1.6), True),
    "scale_one_ulp_apart": ((0.031, -2.6), (float(torch.nextafter(torch.tensor(0.031), torch.tensor(1.0))), -2.6), False),
    "offsets_round_apart": ((0.031, 2.4), (0.031, 2.6), False),
    "nan_offsets": ((0.031, float("nan")), (0.031, float("nan")), False),
    "negative_zero_scale_bits": ((0.0, 1.0), (-0.0, 1.0), False),
}


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("case", list(PAIRS))
def test_a1_runs_unless_the_earlier_parameters_are_the_same(case, dtype):
    (s, o), (es, eo), same = PAIRS[case]
    g = torch.Generator(device=DEV).manual_seed(len(case))
    x = (torch.randn(3, 37, 16, device=DEV, generator=g) * 3).to(dtype)
    scale, offset, e_scale, e_offset = _t(s), _t(o), _t(es), _t(eo)
    lib = _native.library()
    out = torch.full(x.shape, GARBAGE, dtype=torch.int8, device=DEV)
    stream = torch.cuda.current_stream().cuda_stream
    p = lambda t: ctypes.c_void_p(None if t is None else t.data_ptr())  # noqa: E731
    lib.check(lib.ffq_quantize_by_tile_unless_same(p(x), ops._tag(x.dtype), p(scale), p(offset), x.numel(), 8.0, p(e_scale), p(e_offset), p(out), stream))
    if same:
        assert bool((out == GARBAGE).all())
    else:
        assert torch.equal(out, ops.quantize_by_tile(x, scale, x.shape, 8, torch.int8, offset))
    # the wrapper: codes where they differ; outside the launch's coverage nothing is launched
    got = ops.quantize_by_tile_unless_same(x, scale, offset, 8, e_scale, e_offset)
    if not same:
        assert torch.equal(got, ops.quantize_by_tile(x, scale, x.shape, 8, torch.int8, offset))
    assert ops.quantize_by_tile_unless_same(x.reshape(-1)[:40], scale, offset, 8, e_scale, e_offset) is None  # not whole chunks
    assert ops.quantize_by_tile_unless_same(x, torch.cat([scale, scale]), offset, 8, e_scale, e_offset) is None  # not per tensor


@pytest.mark.parametrize("weight_offsets", ["none", "all_round_to_zero", "live"])
@pytest.mark.parametrize("case", ["equal", "offsets_round_alike", "scale_one_ulp_apart", "offsets_round_apart"])
def test_linear_reads_the_codes_in_force(case, weight_offsets):
    (s, o), (es, eo), same = PAIRS[case]
    g = torch.Generator(device=DEV).manual_seed(len(case) + len(weight_offsets))
    m, n, k = 2000, 2048, 512  # ragged M; 8 x 8 tiles of 256 x 256
    x = (torch.randn(m, k, device=DEV, generator=g) * 2).to(torch.bfloat16)
    scale, offset, e_scale, e_offset = _t(s), _t(o), _t(es), _t(eo)
    first = ops.quantize_by_tile(x, e_scale, x.shape, 8, torch.int8, e_offset)
    own = ops.quantize_by_tile(x, scale, x.shape, 8, torch.int8, offset)
    wq = torch.randint(-128, 128, (n, k), device=DEV, dtype=torch.int8, generator=g)
    sw = torch.rand(n, device=DEV, generator=g) * 1e-3 + 1e-4
    ow = None
    if weight_offsets == "all_round_to_zero":
        ow = torch.zeros(n, device=DEV) + 0.25
    if weight_offsets == "live":
        ow = torch.zeros(n, device=DEV)
        ow[5] = -3.0
    want = ops.linear_w8a8(own, wq, scale, offset, sw, ow, None, out_dtype=torch.bfloat16)
    assert ops.linear_w8a8_takes_earlier(m, n, k)
    # the later quantizer's launch, as the product issues it: `maybe` is unwritten where the parameters are the same
    maybe = ops.quantize_by_tile_unless_same(x, scale, offset, 8, e_scale, e_offset)
    if same:
        maybe.fill_(GARBAGE)
        assert torch.equal(own, first)
    for _ in range(2):
        got = ops.linear_w8a8_earlier(maybe, (first, e_scale, e_offset), wq, scale, offset, sw, ow, out_dtype=torch.bfloat16)
        assert torch.equal(got.view(torch.int16), want.view(torch.int16))
    # the earlier codes are this linear's codes whenever the parameters are the same — written or not
    got = ops.linear_w8a8_earlier(own, (first, e_scale, e_offset), wq, scale, offset, sw, ow, out_dtype=torch.bfloat16)
    assert torch.equal(got.view(torch.int16), want.view(torch.int16))
    # outside the persistent kernel's shapes: nothing launched, the caller settles
    assert not ops.linear_w8a8_takes_earlier(100, n, k)
    assert ops.linear_w8a8_earlier(maybe[:100], (first[:100], e_scale, e_offset), wq, scale, offset, sw, ow) is None


def test_linear_with_earlier_codes_replays_in_a_hipgraph():
    g = torch.Generator(device=DEV).manual_seed(3)
    m, n, k = 2048, 2048, 256
    x = (torch.randn(m, k, device=DEV, generator=g) * 2).to(torch.bfloat16)
    scale, offset, e_scale, e_offset = _t(0.02), _t(1.0), _t(0.02), _t(1.0)
    wq = torch.randint(-128, 128, (n, k), device=DEV, dtype=torch.int8, generator=g)
    sw = torch.rand(n, device=DEV, generator=g) * 1e-3 + 1e-4
    first = ops.quantize_by_tile(x, e_scale, x.shape, 8, torch.int8, e_offset)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            maybe = ops.quantize_by_tile_unless_same(x, scale, offset, 8, e_scale, e_offset)
            out = ops.linear_w8a8_earlier(maybe, (first, e_scale, e_offset), wq, scale, offset, sw, None, out_dtype=torch.bfloat16)
    for new_scale in (0.02, 0.025, 0.02):  # the parameters move between replays: the same graph takes either side
        scale.fill_(new_scale)
        maybe.fill_(GARBAGE)
        graph.replay()
        torch.cuda.synchronize()
        own = ops.quantize_by_tile(x, scale, x.shape, 8, torch.int8, offset)
        want = ops.linear_w8a8(own, wq, scale, offset, sw, None, None, out_dtype=torch.bfloat16)
        assert torch.equal(out.view(torch.int16), want.view(torch.int16))
        assert bool((maybe == GARBAGE).all()) == (new_scale == 0.02)


@pytest.mark.parametrize("route", ["equal_parameters", "equal_parameters_a_weight_offset", "different_scale"])
def test_gate_up_while_estimating_reads_gate_codes_where_up_left_none(route):
    g = torch.Generator(device=DEV).manual_seed(len(route))
    m, n, k = 2000, 2048, 512
    x = (torch.randn(m, k, device=DEV, generator=g) * 2).to(torch.bfloat16)
    sg, og = _t(0.031), _t(-2.6)
    su, ou = (_t(0.04), _t(-2.6)) if route == "different_scale" else (_t(0.031), _t(-3.4))  # (-3.4 rounds to -3 like -2.6)
    xg = ops.quantize_by_tile(x, sg, x.shape, 8, torch.int8, og)
    xu = ops.quantize_by_tile(x, su, x.shape, 8, torch.int8, ou)
    wg = torch.randint(-128, 128, (n, k), device=DEV, dtype=torch.int8, generator=g)
    wu = torch.randint(-128, 128, (n, k), device=DEV, dtype=torch.int8, generator=g)
    swg = torch.rand(n, device=DEV, generator=g) * 1e-3 + 1e-4
    swu = torch.rand(n, device=DEV, generator=g) * 1e-3 + 1e-4
    owg = owu = None
    if route == "equal_parameters_a_weight_offset":  # the two-launch route although the input quantizers agree
        owg, owu = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
        owu[n - 1] = 2.0
    gate = ops.linear_w8a8(xg, wg, sg, og, swg, owg, None, out_dtype=torch.bfloat16)
    up = ops.linear_w8a8(xu, wu, su, ou, swu, owu, None, out_dtype=torch.bfloat16)
    want = ops.silu_mul_quantize(gate, up, (), want_product=True)[0]
    maybe = ops.quantize_by_tile_unless_same(x, su, ou, 8, sg, og)
    if route != "different_scale":
        maybe.fill_(GARBAGE)
    got = ops.mlp_gate_up_w8a8_estimating(xg, maybe, wg, wu, (sg, og), (su, ou), (swg, owg), (swu, owu))
    assert torch.equal(got.view(torch.int16), want.view(torch.int16))


def test_settle_writes_the_codes_in_force():
    class Holder:  # (what settle needs of a QuantizedTensor)
        def __init__(self, raw):
            self.raw_data = raw

    g = torch.Generator(device=DEV).manual_seed(11)
    x = (torch.randn(64, 256, device=DEV, generator=g) * 2).to(torch.bfloat16)
    for (s, o), (es, eo), same in PAIRS.values():
        scale, offset, e_scale, e_offset = _t(s), _t(o), _t(es), _t(eo)
        first = ops.quantize_by_tile(x, e_scale, x.shape, 8, torch.int8, e_offset)
        maybe = ops.quantize_by_tile_unless_same(x, scale, offset, 8, e_scale, e_offset)
        if same:
            maybe.fill_(GARBAGE)
        holder = Holder(maybe)
        RECENT.mark_undecided(holder, (first, e_scale, e_offset), scale, offset)
        assert RECENT.earlier_of(holder) is not None
        RECENT.settle(holder)
        assert RECENT.earlier_of(holder) is None
        assert torch.equal(maybe, ops.quantize_by_tile(x, scale, x.shape, 8, torch.int8, offset))
    RECENT.clear()
