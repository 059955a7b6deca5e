"""The skinny form of the weight-only linear (csrc/ffq_wskinny.hip: 1 <= M <= 128 token rows, the contraction as a stream over the
weight codes) against
  * the float64 product of THE SAME operands — the reference's own operand is A2's bf16 weight (fallback.py:86-112), which
    ops.dequantize_by_tile produces bit for bit — within one output rounding,
  * exact values where every partial sum is exact in fp32 (small integers x power-of-two scales): any summation order, any split,
  * the 256-row-tile kernel on the same operands (ffq_force_generic_kernels selects it),
at T in {1, 7, 16} (skinny), {17 ... 512} (the 128-column tiles of csrc/ffq_wmid.hip since round 6; exactness also at 129, 300, 512 under every split), for int8 containers and packed nibbles, per-tensor /
per-channel / group-128 parameters, offsets, bias, f32 output, ragged N, one to three weight matrices in one launch, every forced split,
repeated launches (a race hunt over the ticketed split-K reduction) and a hipGraph replay.
"""

import pytest
import torch

from fastforward_amd import _native, ops

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(autouse=True)
def _backend(hip_backend):
    yield


def _dequantized(w, scale, offset, group, k):
    tile = (1, group)
    return ops.dequantize_by_tile(w, scale, tile, offset, torch.bfloat16) if scale.numel() > 1 else ops.dequantize_by_tile(w, scale, w.shape, offset, torch.bfloat16)


def _check(x, w, scale, offset, group, got):
    wd = _dequantized(w, scale, offset, group, x.shape[-1]).double()
    want = x.double() @ wd.t()
    err = (got.double() - want).abs()
    bound = 2.0 ** -8 * want.abs() + 1e-4 * float(want.abs().max() + 1e-30)
    assert bool((err <= bound).all()), f"max err {float(err.max())} at |y| {float(want.abs().max())}"


@pytest.mark.parametrize("tokens", [1, 4, 7, 16, 17, 33, 64, 100, 128, 300, 512])
@pytest.mark.parametrize("n,k", [(4096, 4096), (1024, 4096), (1000, 1024), (4096, 14336), (14336, 1152)], ids=str)
def test_weight_only_linear_at_few_rows_matches_float64_of_the_same_operands(tokens, n, k):
    g = torch.Generator(device=DEV).manual_seed(tokens * 7 + n)
    x = torch.randn(tokens, k, device=DEV, generator=g).to(torch.bfloat16)
    w8 = torch.randint(-128, 128, (n, k), device=DEV, dtype=torch.int8, generator=g)
    s_row = torch.rand(n, device=DEV, generator=g) * 1e-2 + 1e-3
    o_row = torch.round(torch.randn(n, device=DEV, generator=g) * 5)
    for scale, offset, group in ((s_row, None, k), (s_row, o_row, k), (s_row[:1].clone(), None, k)):
        got = ops.linear_wq(x, w8, scale, offset, group=group)
        assert got is not None
        _check(x, w8, scale, offset, group, got)
    if k % 128 == 0:
        w4 = torch.randint(-8, 8, (n, k), device=DEV, dtype=torch.int8, generator=g)
        s_g = torch.rand(n * (k // 128), device=DEV, generator=g) * 1e-1 + 1e-2
        o_g = torch.round(torch.randn(n * (k // 128), device=DEV, generator=g) * 2)
        packed = ops.pack_int4(w4, block=128)
        for offset in (None, o_g):
            from_codes = ops.linear_wq(x, w4, s_g, offset, group=128)
            from_nibbles = ops.linear_wq(x, packed, s_g, offset, group=128, pack_block=128)
            _check(x, w4, s_g, offset, 128, from_codes)
            assert torch.equal(from_codes, from_nibbles), "packed nibbles and int8 containers of the same codes disagree"


@pytest.mark.parametrize("tokens", [1, 7, 17, 64, 128, 129, 300, 512])
def test_skinny_form_is_exact_where_the_sum_is_order_independent(tokens):
    """Small-integer activations x integer codes x power-of-two scales: every product and partial sum is exact in fp32, so the result
    must equal the float64 value bit for bit — whatever the split, for every storage form, bias and f32 output included — and the
    256-row-tile kernel must give the same tensor."""
    g = torch.Generator(device=DEV).manual_seed(tokens)
    lib = _native.library()
    for n, k in ((4096, 4096), (768, 2048), (200, 512)):
        x = torch.randint(-4, 5, (tokens, k), device=DEV, generator=g).to(torch.bfloat16)
        w4 = torch.randint(-8, 8, (n, k), device=DEV, dtype=torch.int8, generator=g)
        s = torch.full((n * (k // 128),), 0.25, device=DEV)
        o = torch.round(torch.randn(n * (k // 128), device=DEV, generator=g) * 2)
        bias = torch.randint(-8, 9, (n,), device=DEV, generator=g).to(torch.bfloat16)
        for offset in (None, o):
            wd = (w4.double().view(n, k // 128, 128) + (0 if offset is None else offset.double().view(n, k // 128, 1))).view(n, k) * 0.25
            exact = x.double() @ wd.t()
            plan = int(lib.ffq_linear_wq_split(tokens, n, k, 0))
            for split in sorted({0, 1, 2, plan, k // 256}):
                got = ops.linear_wq(x, w4, s, offset, group=128, split=split)
                assert torch.equal(got, exact.to(torch.bfloat16)), (n, k, split)
            packed = ops.pack_int4(w4, block=128)
            assert torch.equal(ops.linear_wq(x, packed, s, offset, group=128, pack_block=128), exact.to(torch.bfloat16))
            assert torch.equal(ops.linear_wq(x, w4, s, offset, group=128, out_dtype=torch.float32), exact.float())
            assert torch.equal(ops.linear_wq(x, w4, s, offset, group=128, bias=bias, out_dtype=torch.float32), (exact + bias.double()).float())
            previous = lib.ffq_force_generic_kernels(1)
            try:
                tiles = ops.linear_wq(x, w4, s, offset, group=128)
            finally:
                lib.ffq_force_generic_kernels(previous)
            assert torch.equal(tiles, exact.to(torch.bfloat16))


def test_q_k_v_in_one_skinny_launch_equal_three_launches():
    g = torch.Generator(device=DEV).manual_seed(5)
    k = 4096
    for tokens in (1, 64):
        x = torch.randn(tokens, k, device=DEV, generator=g).to(torch.bfloat16)
        ws = [torch.randint(-128, 128, (n, k), device=DEV, dtype=torch.int8, generator=g) for n in (4096, 1024, 1024)]
        ss = [torch.rand(n, device=DEV, generator=g) * 1e-2 + 1e-3 for n in (4096, 1024, 1024)]
        lib = _native.library()
        # the same K split on both sides: the summation order is a function of the plan ("summation order")
        split = int(lib.ffq_linear_wq_split(tokens, 6144, k, 0))
        together = ops.linear_wq_multi(x, ws, ss, [None] * 3, split=split)
        assert together is not None
        for w, s, out in zip(ws, ss, together):
            assert torch.equal(out, ops.linear_wq(x, w, s, None, split=split))


def test_repeated_skinny_launches_are_bit_identical_also_from_a_graph():
    """The ticketed reduction: whichever wave arrives last adds the partial sums in slice order — 200 repeats on two streams and a
    hipGraph replay must reproduce the first result bit for bit, and the ticket words must be zero afterwards."""
    g = torch.Generator(device=DEV).manual_seed(3)
    k = 4096
    x = torch.randn(64, k, device=DEV, generator=g).to(torch.bfloat16)
    w = torch.randint(-128, 128, (4096, k), device=DEV, dtype=torch.int8, generator=g)
    s = torch.rand(4096, device=DEV, generator=g) * 1e-2 + 1e-3
    first = ops.linear_wq(x, w, s, None)
    for _ in range(200):
        assert torch.equal(ops.linear_wq(x, w, s, None), first)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(50):
            assert torch.equal(ops.linear_wq(x, w, s, None), first)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            out = ops.linear_wq(x, w, s, None)
    torch.cuda.current_stream().wait_stream(side)
    for _ in range(20):
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, first)
    for buf in ops._TICKETS.values():
        assert int(buf.abs().sum()) == 0
