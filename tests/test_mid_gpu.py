"""The 128-column-tile form of the weight-only linear (csrc/ffq_wmid.hip: up to 512 token rows, codes converted once per block into a
k-ordered LDS image, ticketed split-K that never waits) — reference fallback.py:86-112 (A2 of the weight, then F.linear).
tests/test_skinny_gpu.py already drives it through ops.linear_wq at 17 ... 512 rows against float64 of the same operands and, where
the sum is order-independent, against exact values under every split. Here: what is specific to this form —
  * every storage form of one weight gives the same bits: int8 containers and packed nibbles of packing blocks 32 / 64 / 128 / 256
    (GGUF's blocks take this form at EVERY row count up to 512, also where the skinny form is preferred: ADVICE r5),
  * q / k / v in one launch equal three launches under the same split,
  * the plan's scratch figures cover the launch the library actually makes (tickets zero afterwards, no silent drop to split 1),
  * repeated launches on two streams and a hipGraph replay reproduce the first result bit for bit (race hunt over the tickets),
  * the 256-row-tile kernel on the same operands (ffq_force_generic_kernels) agrees where the sums are exact.
"""

import pytest
import torch

from fastforward_amd import _native, ops

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(autouse=True)
def _backend(hip_backend):
    yield


@pytest.mark.parametrize("tokens", [1, 9, 40, 64, 65, 200, 512])
@pytest.mark.parametrize("block", [32, 64, 128, 256])
def test_every_packing_block_gives_the_bits_of_the_int8_container(tokens, block):
    g = torch.Generator(device=DEV).manual_seed(tokens + block)
    for n, k in ((1024, 2048), (300, 512)):
        x = torch.randn(tokens, k, device=DEV, generator=g).to(torch.bfloat16)
        w4 = torch.randint(-8, 8, (n, k), device=DEV, dtype=torch.int8, generator=g)
        group = 256
        s = torch.rand(n * (k // group), device=DEV, generator=g) * 0.1 + 0.01
        o = torch.round(torch.randn(n * (k // group), device=DEV, generator=g) * 2)
        packed = ops.pack_int4(w4, block=block)
        assert torch.equal(ops.unpack_int4(packed, (n, k), block=block), w4)
        for offset in (None, o):
            from_nibbles = ops.linear_wq(x, packed, s, offset, group=group, pack_block=block)
            assert from_nibbles is not None
            if tokens >= 17:  # both containers take the 128-column tiles with the same plan: the same bits
                assert torch.equal(from_nibbles, ops.linear_wq(x, w4, s, offset, group=group)), (tokens, block, n, k)
            # (below 17 rows the int8 container takes the skinny form, whose summation order is its own: compare with float64 only)
            wd = ops.dequantize_by_tile(w4, s, (1, group), offset, torch.bfloat16).double()
            ref = x.double() @ wd.t()
            err = (from_nibbles.double() - ref).abs()
            assert bool((err <= 2.0 ** -8 * ref.abs() + 1e-4 * float(ref.abs().max() + 1e-30)).all())


@pytest.mark.parametrize("tokens", [33, 300, 512])
def test_q_k_v_in_one_launch_equal_three_launches(tokens):
    g = torch.Generator(device=DEV).manual_seed(tokens)
    k = 4096
    lib = _native.library()
    x = torch.randn(tokens, k, device=DEV, generator=g).to(torch.bfloat16)
    ns = (4096, 1024, 1000)  # the last matrix may be ragged
    ws = [torch.randint(-128, 128, (n, k), device=DEV, dtype=torch.int8, generator=g) for n in ns]
    ss = [torch.rand(n, device=DEV, generator=g) * 1e-2 + 1e-3 for n in ns]
    for split in (1, 2, int(lib.ffq_linear_wq_split(tokens, sum(ns), k, 0))):
        together = ops.linear_wq_multi(x, ws, ss, [None] * 3, split=split)
        assert together is not None
        for w, s, out in zip(ws, ss, together):
            assert torch.equal(out, ops.linear_wq(x, w, s, None, split=split)), split


def test_the_plan_covers_the_launch_and_leaves_the_tickets_zero():
    """ffq_linear_wq_split / _tickets / _slab_bytes answer for the form that runs: a launch with the library's plan must not fall back to
    split 1 for lack of scratch — checked by comparing with the same split forced (a forced split FAILS when the scratch is short)."""
    g = torch.Generator(device=DEV).manual_seed(11)
    lib = _native.library()
    for tokens, n, k, block in ((512, 1024, 4096, 0), (256, 1024, 4096, 0), (128, 4096, 14336, 0), (4, 4096, 4096, 32), (1, 4096, 4096, 64)):
        x = torch.randn(tokens, k, device=DEV, generator=g).to(torch.bfloat16)
        w = torch.randint(-8, 8, (n, k), device=DEV, dtype=torch.int8, generator=g)
        s = torch.rand(n, device=DEV, generator=g) * 1e-2 + 1e-3
        codes = ops.pack_int4(w, block=block) if block else w
        plan = int(lib.ffq_linear_wq_split(tokens, n, k, 0))
        assert int(lib.ffq_linear_wq_tickets(tokens, n, k, 0)) > 0
        by_plan = ops.linear_wq(x, codes, s, None, pack_block=block)
        if block == 0:
            assert torch.equal(by_plan, ops.linear_wq(x, codes, s, None, pack_block=block, split=plan))
        else:  # the 128-column tiles take over from the skinny form: their own plan
            assert int(lib.ffq_linear_wq_slab_bytes(tokens, n, k, 0, plan)) > 0
            forced = [ops.linear_wq(x, codes, s, None, pack_block=block, split=sp) for sp in (2, 3, 4)]
            assert any(torch.equal(by_plan, f) for f in forced), "the plan's launch matches no split > 1: it ran without its split"
    torch.cuda.synchronize()
    for buf in ops._TICKETS.values():
        assert int(buf.abs().sum()) == 0


def test_repeated_launches_are_bit_identical_also_from_a_graph():
    g = torch.Generator(device=DEV).manual_seed(3)
    k = 4096
    x = torch.randn(128, k, device=DEV, generator=g).to(torch.bfloat16)
    w = torch.randint(-128, 128, (4096, k), device=DEV, dtype=torch.int8, generator=g)
    s = torch.rand(4096, device=DEV, generator=g) * 1e-2 + 1e-3
    assert int(_native.library().ffq_linear_wq_split(128, 4096, k, 0)) > 1  # the exchange is on this path (64 tiles: four K slices each)
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


@pytest.mark.parametrize("tokens", [17, 129, 512])
def test_exact_sums_agree_with_the_256_row_tiles_and_the_c_plus_plus_route(tokens):
    g = torch.Generator(device=DEV).manual_seed(tokens)
    lib = _native.library()
    n, k = 768, 2048
    x = torch.randint(-4, 5, (tokens, k), device=DEV, generator=g).to(torch.bfloat16)
    w = torch.randint(-128, 128, (n, k), device=DEV, dtype=torch.int8, generator=g)
    s = torch.full((n,), 2.0 ** -6, device=DEV)
    bias = torch.randint(-8, 9, (n,), device=DEV, generator=g).to(torch.bfloat16)
    exact = (x.double() @ (w.double() * 2.0 ** -6).t())
    got = ops.linear_wq(x, w, s, None)
    assert torch.equal(got, exact.to(torch.bfloat16))
    assert torch.equal(ops.linear_wq(x, w, s, None, bias=bias, out_dtype=torch.float32), (exact + bias.double()).float())
    previous = lib.ffq_force_generic_kernels(1)
    try:
        tiles = ops.linear_wq(x, w, s, None)
    finally:
        lib.ffq_force_generic_kernels(previous)
    assert torch.equal(tiles, got)
    if ops.NATIVE_DISPATCH:  # dispatcher -> C++ -> C ABI (csrc/ffq_torch.cpp) sizes the scratch itself
        via_op = torch.ops.fastforward_amd.linear_wq(x, w, s, None, k, None, torch.bfloat16, 0, -1, 0)
        assert torch.equal(via_op, got)


@pytest.mark.parametrize("tokens", [17, 64, 65, 128, 300, 512])
@pytest.mark.parametrize("form", ["int8 per channel", "int8 per channel + offset", "int8 per tensor", "nibbles g128", "nibbles g128 + offset", "nibbles block 256 g256"])
def test_the_lds_dma_kernel_and_the_register_staged_kernel_give_the_same_bits(tokens, form):
    """wq_mid_dma_kernel (operands by LDS-DMA into a ring, codes converted on the way into the MFMA's registers) against wq_mid_kernel
    (both operands staged through registers, codes converted into a bf16 LDS image; selected with bit 2 of ffq_force_generic_kernels): the
    same tiles, K slices and k order — bit-equal outputs for every forced split, ragged M and N, one to three matrices, bias and f32."""
    g = torch.Generator(device=DEV).manual_seed(tokens + len(form))
    lib = _native.library()
    for n, k in ((1024, 4096), (1000, 1024), (4096, 1152)):
        x = torch.randn(tokens, k, device=DEV, generator=g).to(torch.bfloat16)
        kwargs = {}
        if form.startswith("int8"):
            w = torch.randint(-128, 128, (n, k), device=DEV, dtype=torch.int8, generator=g)
            s = (torch.rand(1 if "tensor" in form else n, device=DEV, generator=g) * 1e-2 + 1e-3)
            o = torch.round(torch.randn(s.numel(), device=DEV, generator=g) * 5) if "offset" in form else None
            group = k
        else:
            block, group = (256, 256) if "256" in form else (128, 128)
            if k % group:
                continue
            w4 = torch.randint(-8, 8, (n, k), device=DEV, dtype=torch.int8, generator=g)
            w = ops.pack_int4(w4, block=block)
            s = torch.rand(n * (k // group), device=DEV, generator=g) * 0.1 + 0.01
            o = torch.round(torch.randn(s.numel(), device=DEV, generator=g) * 2) if "offset" in form else None
            kwargs = dict(pack_block=block)
        bias = torch.randn(n, device=DEV, generator=g).to(torch.bfloat16)
        for split in (0, 1, 2, 3):
            for extra in (dict(), dict(bias=bias, out_dtype=torch.float32)):
                dma = ops.linear_wq(x, w, s, o, group=group, split=split, **kwargs, **extra)
                previous = lib.ffq_force_generic_kernels(4)
                try:
                    staged = ops.linear_wq(x, w, s, o, group=group, split=split, **kwargs, **extra)
                finally:
                    lib.ffq_force_generic_kernels(previous)
                assert torch.equal(dma, staged), (form, n, k, split, list(extra))
    torch.cuda.synchronize()
    for buf in ops._TICKETS.values():
        assert int(buf.abs().sum()) == 0
