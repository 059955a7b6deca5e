"""The int8 GEMM family of csrc/ffq_linear.hip beyond the fixed BASELINE shapes: seeded random shapes, contraction
lengths that take the tail kernel, real and all-zero weight offsets, and the output quantizer inside the epilogue.

Ground truth is never another launch of the kernel under test: the accumulator is a float64 matmul of the codes on the
device (exact below 2^53), the fp32 epilogue is restated with elementwise torch ops (one IEEE operation each, the order of
csrc/ffq_linear.hip), and A1 of the re-quantizing epilogue is the A1 kernel (pinned by fixtures G1-G3) on that restated
tensor. Reference path: src/fastforward/_gen/fallback.py:77-112 (dequantize, dequantize, F.linear, output quantizer).
"""

import random

import pytest
import torch

import fastforward_amd as ff

from fastforward_amd import ops
from fastforward_amd.quantization.affine import function as affine_function

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(autouse=True)
def _backend(hip_backend):
    yield


def _accumulators(xq: torch.Tensor, wq: torch.Tensor) -> torch.Tensor:
    """sum_k xq[m, k] * wq[n, k], exact, as fp32 (|acc| < 2^24 for the sizes used here is asserted by the callers)."""
    out = torch.empty(xq.shape[0], wq.shape[0], device=xq.device, dtype=torch.float32)
    w64 = wq.double().T.contiguous()
    for r0 in range(0, xq.shape[0], 4096):
        acc = (xq[r0:r0 + 4096].double() @ w64).round()
        assert float(acc.abs().max()) < 2**24
        out[r0:r0 + 4096] = acc.float()
    return out


def _epilogue(acc, xq, wq, sx, ox, sw, ow, bias=None):
    """csrc/ffq_linear.hip: v = float(acc) + ox * rowsum_w[n];  v += ow[n] * rowsum_x[m];  v += (K * ox) * ow[n];
    y = (sx * sw[n]) * v (+ bias) — every step its own fp32 rounding."""
    k = xq.shape[1]
    v = acc
    oxr = None if ox is None else torch.round(ox).reshape(-1, 1)
    if oxr is not None:
        v = v + oxr * wq.sum(dim=1, dtype=torch.int64).float()[None, :]
    if ow is not None:
        owr = torch.round(ow).reshape(1, -1)
        v = v + owr * xq.sum(dim=1, dtype=torch.int64).float()[:, None]
        v = v + (float(k) * (oxr if oxr is not None else torch.zeros(1, 1, device=acc.device))) * owr
    y = (sx.reshape(-1, 1) * sw.reshape(1, -1)) * v
    return y if bias is None else y + bias.float()[None, :]


def _random_case(rng: random.Random, seed: int):
    m = rng.choice([rng.randint(1, 300), rng.randint(129, 6000), 256 * rng.randint(1, 40), rng.randint(4000, 20000)])
    n = rng.choice([8 * rng.randint(1, 40), 64 * rng.randint(2, 64), 128 * rng.randint(1, 40), rng.randint(1, 3000)])
    k = rng.choice([16 * rng.randint(1, 40), 64 * rng.randint(1, 64), 128 * rng.randint(2, 48)])
    g = torch.Generator(device=DEV).manual_seed(seed)
    xq = torch.randint(-128, 128, (m, k), device=DEV, dtype=torch.int8, generator=g)
    wq = torch.randint(-128, 128, (n, k), device=DEV, dtype=torch.int8, generator=g)
    return m, n, k, g, xq, wq


@pytest.mark.parametrize("block", range(10))
def test_gemm_fuzz(block):
    """120 seeded random problems (12 per block): every size class of M / N / K, with and without activation offsets,
    per-channel weight offsets that are real, all zero (the symmetric quantizer's buffer) or absent, per-token activation
    parameters, bias, bf16 / fp16 / fp32 outputs; MLP mode where the shape admits it. Every output bit is checked."""
    rng = random.Random(1000 + block)
    for case in range(12):
        m, n, k, g, xq, wq = _random_case(rng, 100 * block + case)
        per_token = rng.random() < 0.2
        sx = torch.rand(m if per_token else 1, device=DEV, generator=g) * 0.02 + 0.005
        ox = None if rng.random() < 0.2 else torch.round(torch.randn(m if per_token else 1, device=DEV, generator=g) * 20)
        sw = torch.rand(n, device=DEV, generator=g) * 1e-3 + 1e-4
        kind = rng.choice(["none", "none", "zero", "real"])
        ow = None if kind == "none" else (torch.zeros(n, device=DEV) if kind == "zero" else torch.round(torch.randn(n, device=DEV, generator=g) * 3))
        bias = torch.randn(n, device=DEV, generator=g).to(torch.bfloat16) if rng.random() < 0.25 else None
        dtype = rng.choice([torch.bfloat16, torch.bfloat16, torch.float16, torch.float32])
        acc = _accumulators(xq, wq)
        want = _epilogue(acc, xq, wq, sx, ox, sw, ow, bias).to(dtype)
        got = ops.linear_w8a8(xq, wq, sx, ox, sw, ow, bias=bias, out_dtype=dtype)
        tag = f"block {block} case {case}: M={m} N={n} K={k} {dtype} ox={'-' if ox is None else ox.numel()} ow={kind} bias={bias is not None}"
        assert torch.equal(got, want), f"{tag}: {int((got != want).sum())} of {want.numel()} outputs differ"
        if n % 128 == 0 and k % 128 == 0 and k >= 256 and not per_token:
            uq = torch.randint(-128, 128, (n, k), device=DEV, dtype=torch.int8, generator=g)
            su = torch.rand(n, device=DEV, generator=g) * 1e-3 + 1e-4
            so, oo = torch.tensor([0.02], device=DEV), torch.tensor([float(rng.randint(-10, 10))], device=DEV)
            codes = ops.mlp_gate_up_w8a8(xq, wq, uq, sx, ox, sw, su, so, oo, 8)
            assert codes is not None, tag
            gate = _epilogue(acc, xq, wq, sx, ox, sw, None).to(torch.bfloat16)
            up = _epilogue(_accumulators(xq, uq), xq, uq, sx, ox, su, None).to(torch.bfloat16)
            z = torch.nn.functional.silu(gate) * up
            assert torch.equal(codes, ops.quantize_by_tile(z, so, z.shape, 8, torch.int8, oo)), f"{tag}: MLP mode differs"


@pytest.mark.parametrize("k", [320, 448, 576])
@pytest.mark.parametrize("weight_offset", [None, "zero", "real"])
def test_contraction_lengths_that_are_not_a_multiple_of_128(k, weight_offset):
    """K = 64 (mod 128) at >= 64 tiles of 256 x 256: the persistent kernel's super-steps do not divide such a K, the
    launcher routes it to the tail kernel (round 2 had an untested 64-byte-row ping-pong kernel here); MLP mode declines."""
    m, n = 4096, 2048
    g = torch.Generator(device=DEV).manual_seed(k)
    xq = torch.randint(-128, 128, (m, k), device=DEV, dtype=torch.int8, generator=g)
    wq = torch.randint(-128, 128, (n, k), device=DEV, dtype=torch.int8, generator=g)
    sx, ox = torch.tensor([0.011], device=DEV), torch.tensor([7.0], device=DEV)
    sw = torch.rand(n, device=DEV, generator=g) * 1e-3 + 1e-4
    ow = None if weight_offset is None else (torch.zeros(n, device=DEV) if weight_offset == "zero" else torch.round(torch.randn(n, device=DEV, generator=g) * 4))
    want = _epilogue(_accumulators(xq, wq), xq, wq, sx, ox, sw, ow)
    for dtype in (torch.bfloat16, torch.float32):
        got = ops.linear_w8a8(xq, wq, sx, ox, sw, ow, out_dtype=dtype)
        assert torch.equal(got, want.to(dtype)), f"{int((got != want.to(dtype)).sum())} outputs differ"
    so, oo = torch.tensor([0.02], device=DEV), torch.tensor([3.0], device=DEV)
    assert ops.mlp_gate_up_w8a8(xq, wq, wq, sx, ox, sw, sw, so, oo, 8) is None


@pytest.mark.parametrize("k", [256, 384, 512, 640])
def test_shortest_k_loops_of_the_persistent_kernels(k):
    """2-5 super-steps per tile at 4 tiles per block (1024 tiles of 256 x 256): the K-loop of the persistent kernels runs across
    tile boundaries, so with a contraction this short nearly every LDS-DMA piece in flight belongs to ANOTHER tile than the one
    being computed — the slot hand-over (pieces issued by the non-computing wave group, waited for one phase before the first
    read) is exercised at its tightest. int8: exact accumulators (plain and gate/up mode); bf16 x weight codes: every storage
    form against one another and within a rounding of float64."""
    m, n = 8192, 8192
    g = torch.Generator(device=DEV).manual_seed(k)
    xq = torch.randint(-128, 128, (m, k), device=DEV, dtype=torch.int8, generator=g)
    wq = torch.randint(-128, 128, (n, k), device=DEV, dtype=torch.int8, generator=g)
    sx, ox = torch.tensor([0.013], device=DEV), torch.tensor([-5.0], device=DEV)
    sw = torch.rand(n, device=DEV, generator=g) * 1e-3 + 1e-4
    acc = _accumulators(xq, wq)
    want = _epilogue(acc, xq, wq, sx, ox, sw, None).to(torch.bfloat16)
    assert torch.equal(ops.linear_w8a8(xq, wq, sx, ox, sw, None, out_dtype=torch.bfloat16), want)
    uq = torch.randint(-128, 128, (n, k), device=DEV, dtype=torch.int8, generator=g)
    so, oo = torch.tensor([0.05], device=DEV), torch.tensor([2.0], device=DEV)
    codes = ops.mlp_gate_up_w8a8(xq, wq, uq, sx, ox, sw, sw, so, oo, 8)
    up = _epilogue(_accumulators(xq, uq), xq, uq, sx, ox, sw, None).to(torch.bfloat16)
    z = torch.nn.functional.silu(want) * up
    assert torch.equal(codes, ops.quantize_by_tile(z, so, z.shape, 8, torch.int8, oo))
    del acc, up, z, codes
    # the weight-code GEMM: small integers x power-of-two scales -> every partial sum is exact in fp32, any order
    x = torch.randint(-4, 5, (m, k), device=DEV, generator=g).to(torch.bfloat16)
    w4 = torch.randint(-8, 8, (n, k), device=DEV, dtype=torch.int8, generator=g)
    s4 = torch.full((n * (k // 128),), 0.25, device=DEV)
    exact = (x.double() @ (w4.double() * 0.25).t()).to(torch.bfloat16)
    for kwargs, weight in ((dict(two_pass=False), w4), (dict(two_pass=True), w4), (dict(two_pass=False, pack_block=128), ops.pack_int4(w4, block=128)),
                           (dict(two_pass=True, pack_block=128), ops.pack_int4(w4, block=128))):
        got = ops.linear_wq(x, weight, s4, None, group=128, **kwargs)
        assert torch.equal(got, exact), kwargs
    u4 = torch.randint(-8, 8, (n, k), device=DEV, dtype=torch.int8, generator=g)
    fused = ops.mlp_gate_up_wq(x, w4, u4, s4, None, s4, None, group=128)
    parts, _ = ops.silu_mul_quantize(exact, ops.linear_wq(x, u4, s4, None, group=128), (), want_product=True)
    assert torch.equal(fused, parts)


def test_repeated_launches_are_bit_identical():
    """Race hunt (tools/gemm_stress.py in small): the persistent kernels hand LDS slots between wave groups with raw barriers and
    counted waits only — 40 repeats of each launch form at a long-K and a short-K shape must reproduce the first result exactly."""
    g = torch.Generator(device=DEV).manual_seed(11)
    for m, n, k in ((16384, 4096, 14336), (8192, 8192, 256)):
        xq = torch.randint(-128, 128, (m, k), device=DEV, dtype=torch.int8, generator=g)
        wq = torch.randint(-128, 128, (n, k), device=DEV, dtype=torch.int8, generator=g)
        sx, ox = torch.tensor([0.02], device=DEV), torch.tensor([3.0], device=DEV)
        sw = torch.rand(n, device=DEV, generator=g) * 1e-3 + 1e-4
        so, oo = torch.tensor([0.05], device=DEV), torch.tensor([-2.0], device=DEV)
        xb = torch.randn(m, k, device=DEV, generator=g).to(torch.bfloat16)
        forms = [lambda: ops.linear_w8a8(xq, wq, sx, ox, sw, None, out_dtype=torch.bfloat16),
                 lambda: ops.mlp_gate_up_w8a8(xq, wq, wq, sx, ox, sw, sw, so, oo, 8),
                 lambda: ops.linear_wq(xb, wq, sw, None, two_pass=True),
                 lambda: ops.linear_wq(xb, wq, sw, None, two_pass=False),
                 lambda: ops.mlp_gate_up_wq(xb, wq, wq, sw, None, sw, None)]
        for i, fn in enumerate(forms):
            first = fn()
            assert all(torch.equal(fn(), first) for _ in range(40)), (m, n, k, i)


def test_one_wave_per_simd_form_equals_the_eight_wave_kernel():
    """The bf16-image form of the weight-only GEMM has two kernels: wq_gemm4w_kernel (one wave per SIMD, 128 x 128 accumulators per
    wave in AGPRs; taken by plain launches of whole 256 x 256 tiles without a split tail, with an even number of 64-deep super-steps,
    bf16 output) and the 8-wave wq_gemm256_kernel. Same MFMA instruction, same k order: bit-equal — one to many tiles per block
    (the K-loop running across tile boundaries), three weight matrices in one launch, down_proj at 16 k tokens (470 MB of
    activations switch the walk to column groups), and since round 6 the gate+up+SiLU*up mode on whole tiles (the same epilogue
    arithmetic in both kernels, equal to the composition of its parts); ragged shapes and an odd super-step count keep both arms on
    the 8-wave kernel (the dispatch must not send them to a kernel that assumes whole tiles)."""
    from fastforward_amd import _native

    lib = _native.library()
    g = torch.Generator(device=DEV).manual_seed(21)

    def operands(m, n, k):
        x = torch.randn(m, k, device=DEV, generator=g).to(torch.bfloat16)
        w = torch.randint(-128, 128, (n, k), device=DEV, dtype=torch.int8, generator=g)
        s = torch.rand(n, device=DEV, generator=g) * 1e-3 + 1e-4
        return x, w, s

    def both(fn):
        got = fn()
        previous = lib.ffq_force_generic_kernels(1)
        try:
            want = fn()
        finally:
            lib.ffq_force_generic_kernels(previous)
        return got, want

    for m, n, k in ((4096, 4096, 4096), (1024, 768, 256), (768, 256, 384), (2048, 1024, 640), (800, 520, 384), (1000, 1001, 640), (768, 256, 320), (8192, 2048, 512), (16384, 1024, 256), (16384, 4096, 14336)):  # (above 512 rows: below, a plain launch takes the 128-column tiles of ffq_wmid.hip)
        x, w, s = operands(m, n, k)
        got, want = both(lambda: ops.linear_wq(x, w, s, None, two_pass=True, split=1))
        assert torch.equal(got, want), (m, n, k)
        rows = slice(0, min(m, 512))  # the 8-wave kernel is checked against exact sums elsewhere; here a sample against the restated operands
        exact = x[rows].double() @ (w.double() * s.double()[:, None]).to(torch.bfloat16).double().t()
        torch.testing.assert_close(got[rows].double(), exact, rtol=2.0**-7, atol=1e-5 * float(exact.abs().max()) * k**0.5)
        del x, w, s, got, want, exact
    for m, last in ((768, 512), (700, 200)):  # whole tiles: the 4-wave kernel; ragged rows and a ragged last matrix: the 8-wave one
        x, w, s = operands(m, 512, 512)
        ws = [w, torch.randint(-128, 128, (256, 512), device=DEV, dtype=torch.int8, generator=g), torch.randint(-128, 128, (last, 512), device=DEV, dtype=torch.int8, generator=g)]
        ss = [s, torch.rand(256, device=DEV, generator=g) * 1e-3 + 1e-4, torch.rand(last, device=DEV, generator=g) * 1e-3 + 1e-4]
        got, want = both(lambda: ops.linear_wq_multi(x, ws, ss, [None] * 3, two_pass=True, split=1))
        assert all(torch.equal(a, b) for a, b in zip(got, want))
        assert all(torch.equal(a, ops.linear_wq(x, wi, si, None, two_pass=True, split=1)) for a, wi, si in zip(got, ws, ss))
    # gate + up + SiLU*up: whole tiles of 256 rows x 128 output columns take the one-wave-per-SIMD kernel's MLP mode (round 6), ragged
    # rows the 8-wave kernel's; one to many tiles per block, silu arguments inside and (x 30) outside the table's window
    for m, n, k, scale in ((700, 384, 512, 1.0), (4096, 1024, 1024, 1.0), (1024, 512, 256, 30.0), (2048, 1024, 640, 30.0), (16384, 2048, 512, 1.0), (768, 384, 384, 1.0)):
        x, w, s = operands(m, n, k)
        x = x * scale
        u = torch.randint(-128, 128, (n, k), device=DEV, dtype=torch.int8, generator=g)
        got, want = both(lambda: ops.mlp_gate_up_wq(x, w, u, s, None, s, None, two_pass=True, split=1))
        assert torch.equal(got, want), (m, n, k)
        parts, _ = ops.silu_mul_quantize(ops.linear_wq(x, w, s, None, two_pass=True, split=1), ops.linear_wq(x, u, s, None, two_pass=True, split=1), (), want_product=True)
        assert torch.equal(got, parts), (m, n, k)


def test_repeated_split_k_launches_are_bit_identical():
    """Race hunt on the split-K exchange (write-through slabs, arrival counters, `sc1` reads of the peers' pieces): 2048-token
    launches of every split the chip allows, the q / k / v launch and the MLP mode, 60 repeats each, back to back on one stream
    (every launch must leave the counters zero for the next), on a second stream with its own counters, and replayed from a
    hipGraph — all bit-equal to the first result."""
    from fastforward_amd import _native

    lib = _native.library()
    g = torch.Generator(device=DEV).manual_seed(5)
    t, k = 2048, 4096
    x = torch.randn(t, k, device=DEV, generator=g).to(torch.bfloat16)
    ws = {n: torch.randint(-128, 128, (n, k), device=DEV, dtype=torch.int8, generator=g) for n in (4096, 1024)}
    ss = {n: torch.rand(n, device=DEV, generator=g) * 1e-3 + 1e-4 for n in ws}
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    forms = []
    for n in ws:
        tiles = (t // 256) * (n // 256)
        for split in (2, 4, 8):
            if tiles * split <= cus:
                forms.append((f"N={n} S={split}", lambda n=n, split=split: ops.linear_wq(x, ws[n], ss[n], None, two_pass=False, split=split)))
        assert int(lib.ffq_linear_wq_split(t, n, k, 0)) > 1  # and the plan the library takes by itself
        forms.append((f"N={n} plan", lambda n=n: ops.linear_wq(x, ws[n], ss[n], None)))
    forms.append(("q/k/v", lambda: torch.cat(ops.linear_wq_multi(x, [ws[4096], ws[1024], ws[1024]], [ss[4096], ss[1024], ss[1024]], [None] * 3), dim=1)))
    forms.append(("mlp", lambda: ops.mlp_gate_up_wq(x, ws[4096], ws[4096], ss[4096], None, ss[4096], None)))
    side = torch.cuda.Stream()
    for name, fn in forms:
        first = fn()
        assert all(torch.equal(fn(), first) for _ in range(60)), name
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            assert all(torch.equal(fn(), first) for _ in range(10)), name
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side):
                captured = fn()
        torch.cuda.synchronize()
        for _ in range(10):
            captured.zero_()
            graph.replay()
            torch.cuda.synchronize()
            assert torch.equal(captured, first), name


def test_repeated_one_wave_per_simd_launches_are_bit_identical():
    """Race hunt on `wq_gemm4w_kernel`'s round-6 schedule (every image of a slot refilled as soon as its last reader is done, counted
    `vmcnt(8)` waits, a K-loop that runs across tile boundaries with an epilogue's stores in flight): the plain launch at a deep and a
    shallow contraction, q / k / v in one launch and the gate + up + SiLU*up mode at 4096 tokens — several tiles per block — 40 repeats
    back to back, 10 on a second stream while the first keeps the chip busy, and replayed from a hipGraph: all bit-equal to the first
    result, which equals the 8-wave kernel's."""
    from fastforward_amd import _native

    lib = _native.library()
    g = torch.Generator(device=DEV).manual_seed(9)
    t = 4096
    xs = {k: torch.randn(t, k, device=DEV, generator=g).to(torch.bfloat16) for k in (4096, 14336, 512)}
    w = lambda n, k: torch.randint(-128, 128, (n, k), device=DEV, dtype=torch.int8, generator=g)  # noqa: E731
    sc = lambda n: torch.rand(n, device=DEV, generator=g) * 1e-3 + 1e-4  # noqa: E731
    wd, sd = w(4096, 14336), sc(4096)
    wo, so = w(4096, 512), sc(4096)
    wq_, wk, wv, sq, sk, sv = w(4096, 4096), w(1024, 4096), w(1024, 4096), sc(4096), sc(1024), sc(1024)
    wg, wu, sg, su = w(14336, 4096), w(14336, 4096), sc(14336), sc(14336)
    forms = [
        ("down", lambda: ops.linear_wq(xs[14336], wd, sd, None, two_pass=True, split=1)),
        ("K=512", lambda: ops.linear_wq(xs[512], wo, so, None, two_pass=True, split=1)),
        ("q/k/v", lambda: torch.cat(ops.linear_wq_multi(xs[4096], [wq_, wk, wv], [sq, sk, sv], [None] * 3, two_pass=True, split=1), dim=1)),
        ("mlp", lambda: ops.mlp_gate_up_wq(xs[4096], wg, wu, sg, None, su, None, two_pass=True, split=1)),
    ]
    side = torch.cuda.Stream()
    for name, fn in forms:
        first = fn()
        previous = lib.ffq_force_generic_kernels(1)
        try:
            assert torch.equal(fn(), first), name  # the 8-wave kernel on the same operands
        finally:
            lib.ffq_force_generic_kernels(previous)
        assert all(torch.equal(fn(), first) for _ in range(40)), name
        side.wait_stream(torch.cuda.current_stream())
        busy = [fn() for _ in range(4)]  # keeps the first stream's CUs contended while the second one launches
        with torch.cuda.stream(side):
            assert all(torch.equal(fn(), first) for _ in range(10)), name
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side):
                captured = fn()
        torch.cuda.synchronize()
        assert all(torch.equal(b, first) for b in busy), name
        for _ in range(5):
            captured.zero_()
            graph.replay()
            torch.cuda.synchronize()
            assert torch.equal(captured, first), name


def test_all_zero_weight_offsets_cost_no_row_sums_and_change_nothing():
    """The offset BUFFER of a symmetric quantizer (reference nn/linear_quantizer.py:164-170) at a persistent-kernel shape:
    same bits as no offset at all; one non-zero entry switches the exact ow terms on (for every column)."""
    m, n, k = 4096, 4096, 1024
    g = torch.Generator(device=DEV).manual_seed(3)
    xq = torch.randint(-128, 128, (m, k), device=DEV, dtype=torch.int8, generator=g)
    wq = torch.randint(-128, 128, (n, k), device=DEV, dtype=torch.int8, generator=g)
    sx, ox = torch.tensor([0.011], device=DEV), torch.tensor([-3.0], device=DEV)
    sw = torch.rand(n, device=DEV, generator=g) * 1e-3 + 1e-4
    plain = ops.linear_w8a8(xq, wq, sx, ox, sw, None, out_dtype=torch.bfloat16)
    zeros = torch.zeros(n, device=DEV)
    assert torch.equal(ops.linear_w8a8(xq, wq, sx, ox, sw, zeros, out_dtype=torch.bfloat16), plain)
    assert torch.equal(ops.linear_w8a8(xq, wq, sx, ox, sw, zeros + 0.4, out_dtype=torch.bfloat16), plain)  # rounds to zero (A2 rounds offsets)
    one_hot = zeros.clone()
    one_hot[n - 1] = 2.0
    want = _epilogue(_accumulators(xq, wq), xq, wq, sx, ox, sw, one_hot).to(torch.bfloat16)
    got = ops.linear_w8a8(xq, wq, sx, ox, sw, one_hot, out_dtype=torch.bfloat16)
    assert torch.equal(got, want) and not torch.equal(got[:, -1], plain[:, -1]) and torch.equal(got[:, :-1], plain[:, :-1])


# ---- the output quantizer inside the GEMM's epilogue (fallback.py:110-111) ----------------------------------------------------
_REQUANT_SHAPES = [(40, 72, 256), (300, 200, 144), (2048, 2048, 512), (4099, 2304, 384)]


@pytest.mark.parametrize("m,n,k", _REQUANT_SHAPES, ids=str)
@pytest.mark.parametrize("container", [torch.int8, torch.bfloat16, torch.float32], ids=str)
@pytest.mark.parametrize("y_dtype", [torch.bfloat16, torch.float16, torch.float32], ids=str)
def test_requantizing_epilogue_is_a1_of_the_plain_output(m, n, k, container, y_dtype):
    """codes == quantize_by_tile(y) where y is the linear's output in the dtype the reference's float GEMM returns, formed
    from the EXACT accumulators (tail kernel and persistent kernel shapes, ragged edges, 8 and 4 bits, with / without offset)."""
    g = torch.Generator(device=DEV).manual_seed(m + n + k)
    xq = torch.randint(-128, 128, (m, k), device=DEV, dtype=torch.int8, generator=g)
    wq = torch.randint(-128, 128, (n, k), device=DEV, dtype=torch.int8, generator=g)
    sx, ox = torch.tensor([0.013], device=DEV), torch.tensor([5.0], device=DEV)
    sw = torch.rand(n, device=DEV, generator=g) * 1e-3 + 1e-4
    y = _epilogue(_accumulators(xq, wq), xq, wq, sx, ox, sw, None).to(y_dtype)
    spread = float(y.float().std())
    for bits, so, oo in ((8, spread / 40, torch.tensor([-17.6], device=DEV)), (4, spread / 3, None), (8, spread * 30, torch.tensor([0.5], device=DEV))):
        so = torch.tensor([so], device=DEV)
        want = ops.quantize_by_tile(y, so, y.shape, bits, container, oo)
        got = ops.linear_w8a8(xq, wq, sx, ox, sw, None, out_dtype=container, out_scale=so, out_offset=oo, out_num_bits=bits, requant_from=y_dtype)
        assert got.dtype == container and torch.equal(got, want), f"{bits} bits: {int((got != want).sum())} of {want.numel()} codes differ"
        assert int(want.float().max() - want.float().min()) >= (0 if so > spread else 7)  # the fine grids are real grids, not saturated tensors


def test_requantizing_epilogue_refuses_a_container_that_cannot_hold_the_codes():
    xq = torch.zeros(16, 64, dtype=torch.int8, device=DEV)
    one = torch.ones(1, device=DEV)
    with pytest.raises(RuntimeError, match="not enough to store"):
        ops.linear_w8a8(xq, xq, one, None, one, None, out_dtype=torch.int8, out_scale=one, out_num_bits=12)


@pytest.mark.parametrize("quantized_dtype", [torch.int8, None], ids=["int8_container", "data_dtype_container"])
def test_dispatcher_runs_a_static_output_quantizer_inside_the_gemm(quantized_dtype, monkeypatch):
    """``ff.nn.functional.linear(xq, wq, output_quantizer=q)`` with a static per-tensor LinearQuantizer: ONE GEMM launch whose
    epilogue quantizes — no quantizer forward (``quantize_affine``) runs for the output — and the QuantizedTensor it returns equals the
    two-pass result (GEMM, then the quantizer's own forward) bit for bit: codes, parameters, dequantize dtype."""
    torch.manual_seed(9)
    lin = torch.nn.Linear(512, 768, bias=True).to(DEV, torch.bfloat16)
    x = torch.randn(6, 200, 512, device=DEV, dtype=torch.bfloat16)
    wq_ = ff.nn.LinearQuantizer(8, granularity=ff.PerChannel(0), quantized_dtype=torch.int8, device=DEV)
    xq_ = ff.nn.LinearQuantizer(8, symmetric=False, quantized_dtype=torch.int8, device=DEV)
    oq = ff.nn.LinearQuantizer(8, symmetric=False, quantized_dtype=quantized_dtype, device=DEV)
    with torch.no_grad(), ff.estimate_ranges(torch.nn.ModuleList([wq_, xq_]), ff.range_setting.running_minmax):
        qw, qx = wq_(lin.weight), xq_(x)
    with torch.no_grad(), ff.strict_quantization(False):
        y = ff.nn.functional.linear(qx, qw, lin.bias)
        with ff.estimate_ranges(oq, ff.range_setting.running_minmax):
            oq(y)
        two_pass = oq(y)
        calls = []  # every A1 a quantizer's forward runs goes through this function (quantization/affine/function.py)
        real = affine_function.quantize_affine
        monkeypatch.setattr(affine_function, "quantize_affine", lambda *a, **k: calls.append(tuple(a[0].shape)) or real(*a, **k))
        fused = ff.nn.functional.linear(qx, qw, lin.bias, output_quantizer=oq)
        assert calls == [], "the output quantizer ran as its own pass"
        # with gradients possibly wanted, or an override active, the quantizer runs as a module (its own forward)
        with ff.disable_quantization(torch.nn.ModuleList([oq])):
            assert not isinstance(ff.nn.functional.linear(qx, qw, lin.bias, output_quantizer=oq), ff.QuantizedTensor)
    with ff.strict_quantization(False):
        with_grad = ff.nn.functional.linear(qx, qw, lin.bias, output_quantizer=oq)  # grad mode on: scale is a Parameter
    assert len(calls) == 1
    for got in (fused, with_grad):
        assert isinstance(got, ff.QuantizedTensor) and got.raw_data.dtype == two_pass.raw_data.dtype
        assert torch.equal(got.raw_data, two_pass.raw_data)
        gp, tp = got.quantization_context.quantization_params, two_pass.quantization_context.quantization_params
        assert gp.dequantize_dtype == tp.dequantize_dtype == torch.bfloat16 and gp.num_bits == tp.num_bits
        assert torch.equal(got.dequantize(), two_pass.dequantize())
    # mm takes the same epilogue
    with torch.no_grad(), ff.strict_quantization(False):
        qwt = ff.quantization.affine.quantize_per_tensor(lin.weight.t().contiguous(), torch.tensor([0.001], device=DEV), None, 8, torch.int8)
        y_mm = ff.nn.functional.mm(qx.reshape(-1, 512), qwt)
        assert torch.equal(ff.nn.functional.mm(qx.reshape(-1, 512), qwt, output_quantizer=oq).raw_data, oq(y_mm).raw_data)


def test_grouped_weights_with_quantized_inputs_take_the_weight_code_gemm():
    """W4 group-128 x A8 (SURVEY 8(b) Seam 2 lists PerBlock(in, 128) weights): group-wise parameters cannot leave an int8
    contraction, so the dispatcher kernel dequantizes the input codes (A2 — the reference's own first step, fallback.py:94-100)
    and runs the bf16 x weight-code GEMM: the reference's operands bit for bit, no float fallback, no vendor GEMM."""
    torch.manual_seed(11)
    tokens, n, k = 4096, 768, 512
    x = torch.randn(tokens, k, device=DEV, dtype=torch.bfloat16)
    w = (torch.randn(n, k, device=DEV) * 0.05).to(torch.bfloat16)
    wq_ = ff.nn.LinearQuantizer(4, granularity=ff.PerBlock(1, 128, 0), quantized_dtype=torch.int8, device=DEV)
    xq_ = ff.nn.LinearQuantizer(8, symmetric=False, quantized_dtype=torch.int8, device=DEV)
    with torch.no_grad(), ff.estimate_ranges(torch.nn.ModuleList([wq_, xq_]), ff.range_setting.running_minmax):
        qw, qx = wq_(w), xq_(x)
    assert ff.dispatcher.dispatch("linear", input=qx, weight=qw) is ff.fused_linear.fused_linear
    with torch.no_grad():  # 64 tokens: the same kernel (no token threshold since round 4: a launch with few tiles splits along K)
        assert ff.dispatcher.dispatch("linear", input=xq_(x[:64]), weight=qw) is ff.fused_linear.fused_linear
        with ff.strict_quantization(False):
            few = ff.nn.functional.linear(xq_(x[:64]), qw)
        want_few = xq_(x[:64]).dequantize().double() @ qw.dequantize().double().t()
        torch.testing.assert_close(few.double(), want_few, rtol=2.0**-8, atol=1e-5 * float(want_few.abs().max()))
    with torch.no_grad(), ff.strict_quantization(False):
        got = ff.nn.functional.linear(qx, qw)
        with ff.fused_linear.weight_only_kernel(False):
            assert ff.dispatcher.dispatch("linear", input=qx, weight=qw) is None
            reference_path = ff.nn.functional.linear(qx, qw)  # dequantize, dequantize, F.linear
    assert got.dtype == torch.bfloat16
    exact = qx.dequantize().double() @ qw.dequantize().double().t()
    torch.testing.assert_close(got.double(), exact, rtol=2.0**-8, atol=1e-5 * float(exact.abs().max()))
    torch.testing.assert_close(got.float(), reference_path.float(), rtol=2.0**-7, atol=2e-4 * float(exact.abs().max()))


def test_a_learnable_weight_offset_written_through_dot_data_is_seen_by_the_next_linear():
    """The zero-offset shortcut of the dispatcher remembers only the derived offset BUFFER of a symmetric quantizer; a learnable
    offset (nn.Parameter) written through ``.data`` — invisible to version counters — takes the device-side decision on every call,
    so the next linear uses the new offsets (reference fallback.py:94-100 dequantizes with the current parameters every call)."""
    torch.manual_seed(3)
    tokens, n, k = 2048, 2048, 512
    x = torch.randn(tokens, k, device=DEV, dtype=torch.bfloat16)
    w = (torch.randn(n, k, device=DEV) * 0.05).to(torch.bfloat16)
    wq_ = ff.nn.LinearQuantizer(8, symmetric=False, granularity=ff.PerChannel(0), quantized_dtype=torch.int8, device=DEV)
    xq_ = ff.nn.LinearQuantizer(8, symmetric=False, quantized_dtype=torch.int8, device=DEV)
    with torch.no_grad(), ff.estimate_ranges(torch.nn.ModuleList([wq_, xq_]), ff.range_setting.running_minmax):
        wq_(w), xq_(x)
    assert isinstance(wq_.offset, torch.nn.Parameter)
    with torch.no_grad(), ff.strict_quantization(False):
        wq_.offset.data.zero_()  # all zero: a version-keyed cache would now learn "zero" ...
        for _ in range(3):
            zero = ff.nn.functional.linear(xq_(x), wq_(w))
        version = wq_.offset._version
        wq_.offset.data.fill_(5.0)  # ... and keep it across this write
        assert wq_.offset._version == version
        qx, qw = xq_(x), wq_(w)
        got = ff.nn.functional.linear(qx, qw)
        # the int8 GEMM contracts the codes exactly: compare with the affine values themselves in float64 (no bf16 operand rounding)
        x_hat = (qx.raw_data.double() + torch.round(xq_.offset.detach().double())) * xq_.scale.detach().double()
        w_hat = (qw.raw_data.double() + torch.round(wq_.offset.detach().double())[:, None]) * wq_.scale.detach().double()[:, None]
        want = x_hat @ w_hat.t()
    assert not torch.equal(got, zero)
    torch.testing.assert_close(got.double(), want, rtol=2.0**-7, atol=1e-3 * float(want.abs().max()))


@pytest.mark.parametrize("m,n,k", [(2048, 2048, 512), (2000, 2112, 256), (4096, 4096, 1024)], ids=str)
@pytest.mark.parametrize("weight_offset", [None, "zero", "real"])
def test_gated_epilogue_is_silu_of_the_gate_times_the_plain_output(m, n, k, weight_offset):
    """ops.linear_w8a8_gated (silu(gate) * linear in the int8 GEMM's epilogue: the MLP's second projection during range estimation)
    == silu_mul_quantize(gate, linear_w8a8(...)) bit for bit — i.e. ATen's F.silu(gate) * up on the two bf16 tensors (mlp.py:36-38;
    tests/test_parity_gpu.py pins silu_mul_quantize to that chain). The gate tensor holds every bf16 pattern (NaN, Inf, denormals,
    both zeros) besides ordinary activations; ragged M, N % 256 != 0, with / without (all-zero) weight offsets."""
    g = torch.Generator(device=DEV).manual_seed(m + n + k)
    xq = torch.randint(-128, 128, (m, k), device=DEV, dtype=torch.int8, generator=g)
    wq = torch.randint(-128, 128, (n, k), device=DEV, dtype=torch.int8, generator=g)
    sx = torch.tensor([0.013], device=DEV)
    ox = torch.tensor([-7.0], device=DEV)
    sw = torch.rand(n, device=DEV, generator=g) * 1e-3 + 1e-4
    ow = None if weight_offset is None else (torch.zeros(n, device=DEV) if weight_offset == "zero" else torch.round(torch.randn(n, device=DEV, generator=g) * 3))
    gate = (torch.randn(m, n, device=DEV, generator=g) * 4).to(torch.bfloat16)
    patterns = torch.arange(65536, device=DEV, dtype=torch.int32).to(torch.int16).view(torch.bfloat16)
    gate.view(-1)[: 65536] = patterns
    gate.view(-1)[-65536:] = patterns.flip(0)
    up = ops.linear_w8a8(xq, wq, sx, ox, sw, ow, None, out_dtype=torch.bfloat16)
    want = ops.silu_mul_quantize(gate, up, (), want_product=True)[0]
    got = ops.linear_w8a8_gated(xq, wq, sx, ox, sw, ow, gate)
    assert got is not None
    eager = torch.nn.functional.silu(gate) * up
    for other in (want, eager):  # the same bits wherever the value is a number, NaN where the chain gives NaN (its payload is not pinned)
        assert torch.equal(got.isnan(), other.isnan())
        assert torch.equal(torch.where(got.isnan(), 0, got.view(torch.int16)), torch.where(other.isnan(), 0, other.view(torch.int16)))


def test_gated_epilogue_declines_what_the_whole_line_path_does_not_cover():
    g = torch.Generator(device=DEV).manual_seed(1)
    for m, n, k in ((2048, 2040, 512), (256, 256, 512), (2048, 2048, 320)):
        xq = torch.randint(-128, 128, (m, k), device=DEV, dtype=torch.int8, generator=g)
        wq = torch.randint(-128, 128, (n, k), device=DEV, dtype=torch.int8, generator=g)
        gate = torch.randn(m, n, device=DEV, generator=g).to(torch.bfloat16)
        assert ops.linear_w8a8_gated(xq, wq, torch.tensor([0.01], device=DEV), None, torch.tensor([0.02], device=DEV), None, gate) is None


@pytest.mark.parametrize("with_nan", [False, True])
def test_gated_epilogue_leaves_the_extrema_of_its_product(with_nan):
    """want_extrema: [min, max] of the product, left by the launch that wrote it == ops.minmax_by_tile over the finished tensor
    (NaN-propagating), launch after launch from the same four accumulator words; ragged M and columns beyond N % 256 included."""
    g = torch.Generator(device=DEV).manual_seed(11)
    for m, n, k in ((2048, 2048, 256), (2000, 2112, 512)):
        xq = torch.randint(-128, 128, (m, k), device=DEV, dtype=torch.int8, generator=g)
        wq = torch.randint(-128, 128, (n, k), device=DEV, dtype=torch.int8, generator=g)
        sx, ox = torch.tensor([0.02], device=DEV), torch.tensor([3.0], device=DEV)
        sw = torch.rand(n, device=DEV, generator=g) * 1e-3 + 1e-4
        ow = torch.round(torch.randn(n, device=DEV, generator=g))
        for trial in range(3):
            gate = (torch.randn(m, n, device=DEV, generator=g) * (trial + 1)).to(torch.bfloat16)
            if with_nan and trial == 1:
                gate[m - 1, n - 1] = float("nan")
            if trial == 2:
                gate[5, 7] = -0.0
            product, pair = ops.linear_w8a8_gated(xq, wq, sx, ox, sw, ow, gate, want_extrema=True)
            lo, hi = ops.minmax_by_tile(product, product.shape)
            assert torch.equal(pair[0:1].isnan(), lo.isnan()) and torch.equal(pair[1:2].isnan(), hi.isnan())
            if not bool(lo.isnan()):
                assert torch.equal(pair.view(torch.int16), torch.cat([lo, hi]).view(torch.int16)), (pair, lo, hi)
    for words in ops._EXTREMA_WORDS.values():
        assert words.tolist() == [-1, 0, 0, 0]


def test_gated_epilogue_with_extrema_inside_a_hipgraph():
    """A captured launch gets accumulator words of its own (zeroed and set by fill nodes of the capture); replays reproduce the
    eager result and leave the words in their initial state."""
    g = torch.Generator(device=DEV).manual_seed(4)
    m, n, k = 2048, 2048, 256
    xq = torch.randint(-128, 128, (m, k), device=DEV, dtype=torch.int8, generator=g)
    wq = torch.randint(-128, 128, (n, k), device=DEV, dtype=torch.int8, generator=g)
    sx, sw = torch.tensor([0.02], device=DEV), torch.rand(n, device=DEV, generator=g) * 1e-3 + 1e-4
    gate = torch.randn(m, n, device=DEV, generator=g).to(torch.bfloat16)
    want, want_pair = ops.linear_w8a8_gated(xq, wq, sx, None, sw, None, gate, want_extrema=True)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        ops.linear_w8a8_gated(xq, wq, sx, None, sw, None, gate, want_extrema=True)  # code objects, allocator
        with torch.cuda.graph(graph, stream=side):
            got, pair = ops.linear_w8a8_gated(xq, wq, sx, None, sw, None, gate, want_extrema=True)
    torch.cuda.current_stream().wait_stream(side)
    for _ in range(3):
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(got, want) and torch.equal(pair, want_pair)


@pytest.mark.parametrize("route", ["equal_parameters", "different_scale", "different_offset", "a_weight_offset", "zero_weight_offsets"])
def test_gate_up_while_estimating_takes_either_route_to_the_same_product(route):
    """ops.mlp_gate_up_w8a8_estimating decides ON THE DEVICE between the one-launch gate + up + SiLU * up kernel (the two input
    quantizers hold equal parameters, no weight offset is live) and the two linears with the gated epilogue. Either way the
    product and its [min, max] are those of silu_mul_quantize(linear_w8a8(xg, Wg), linear_w8a8(xu, Wu)) — bit for bit wherever the
    chain gives a number. Ragged M; an ordinary launch on the same stream afterwards sees the accumulator words in their initial state."""
    g = torch.Generator(device=DEV).manual_seed(len(route))
    m, n, k = 2000, 2048, 512
    x = (torch.randn(m, k, device=DEV, generator=g) * 2).to(torch.bfloat16)
    sg, og = torch.tensor([0.031], device=DEV), torch.tensor([-2.6], device=DEV)
    su, ou = sg.clone(), og.clone()
    if route == "different_scale":
        su = torch.tensor([0.04], device=DEV)
    if route == "different_offset":
        ou = torch.tensor([3.2], device=DEV)
    xg = ops.quantize_by_tile(x, sg, x.shape, 8, torch.int8, og)
    xu = ops.quantize_by_tile(x, su, x.shape, 8, torch.int8, ou)
    wg = torch.randint(-128, 128, (n, k), device=DEV, dtype=torch.int8, generator=g)
    wu = torch.randint(-128, 128, (n, k), device=DEV, dtype=torch.int8, generator=g)
    swg = torch.rand(n, device=DEV, generator=g) * 1e-3 + 1e-4
    swu = torch.rand(n, device=DEV, generator=g) * 1e-3 + 1e-4
    owg = owu = None
    if route == "zero_weight_offsets":
        owg, owu = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV) + 0.25  # rounds to zero
    if route == "a_weight_offset":
        owg, owu = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
        owu[n - 1] = 2.0
    gate = ops.linear_w8a8(xg, wg, sg, og, swg, owg, None, out_dtype=torch.bfloat16)
    up = ops.linear_w8a8(xu, wu, su, ou, swu, owu, None, out_dtype=torch.bfloat16)
    want = ops.silu_mul_quantize(gate, up, (), want_product=True)[0]
    lo, hi = ops.minmax_by_tile(want, want.shape)
    for _ in range(2):
        got, pair = ops.mlp_gate_up_w8a8_estimating(xg, xu, wg, wu, (sg, og), (su, ou), (swg, owg), (swu, owu), want_extrema=True)
        assert torch.equal(got.view(torch.int16), want.view(torch.int16))
        assert torch.equal(pair.view(torch.int16), torch.cat([lo, hi]).view(torch.int16)), (pair, lo, hi)
    plain = ops.mlp_gate_up_w8a8_estimating(xg, xu, wg, wu, (sg, og), (su, ou), (swg, owg), (swu, owu))
    assert torch.equal(plain.view(torch.int16), want.view(torch.int16))
    for words in ops._EXTREMA_WORDS.values():
        assert words.tolist() == [-1, 0, 0, 0]
    assert ops.mlp_gate_up_w8a8_estimating(xg[:100], xu[:100], wg, wu, (sg, og), (su, ou), (swg, None), (swu, None)) is None  # too few tiles


@pytest.mark.parametrize("tokens,rows,k", [(16384, (4096, 1024, 1024), 4096), (4096, (8192, 1024, 1024), 8192), (4300, (512, 256, 300), 1024), (4096, (2048, 512), 512)])  # (>= 64 tiles of 256 x 256 each)
@pytest.mark.parametrize("per_token,with_sums", [(False, True), (True, False)])
def test_q_k_v_as_one_int8_launch_equal_three_launches(tokens, rows, k, per_token, with_sums):
    """ops.linear_w8a8_multi (ABI 9: q_proj / k_proj / v_proj on the code tensor their input quantizers share — reference nn/linear.py:32-39
    three times over fallback.py:77-112 — as ONE launch of the persistent int8 GEMM, the three weight code matrices side by side in one
    buffer) == three ops.linear_w8a8 launches BIT FOR BIT at the Llama-3-8B and 70B shapes, ragged M, a ragged last matrix, per-tensor and
    per-token activation parameters, row sums handed in or reduced by the library; shapes outside the persistent kernel's class return None."""
    g = torch.Generator(device=DEV).manual_seed(tokens + k)
    x = torch.randint(-128, 128, (tokens, k), device=DEV, dtype=torch.int8, generator=g)
    w = torch.randint(-128, 128, (sum(rows), k), device=DEV, dtype=torch.int8, generator=g)
    sw = torch.rand(sum(rows), device=DEV, generator=g) * 1e-3 + 1e-4
    n_x = tokens if per_token else 1
    sx, ox = torch.rand(n_x, device=DEV, generator=g) * 0.05 + 0.01, torch.round(torch.randn(n_x, device=DEV, generator=g) * 20)
    sums = w.to(torch.int32).sum(1, dtype=torch.int32) if with_sums else None
    got = ops.linear_w8a8_multi(x, w, sx, ox, sw, rows, w_rowsum=sums)
    assert got is not None and [tuple(t.shape) for t in got] == [(tokens, r) for r in rows]
    at = 0
    for out, r in zip(got, rows):
        want = ops.linear_w8a8(x, w[at:at + r], sx, ox, sw[at:at + r], None, None, out_dtype=torch.bfloat16, w_rowsum=None if sums is None else sums[at:at + r].contiguous())
        assert torch.equal(out, want), (r, at)
        at += r
    f32 = ops.linear_w8a8_multi(x, w, sx, ox, sw, rows, out_dtype=torch.float32, w_rowsum=sums)
    assert f32 is not None and all(torch.equal(a.to(torch.bfloat16), b) for a, b in zip(f32, got))
    # not this launch's: a middle matrix that is no multiple of 256 rows, too few tiles for the persistent kernel
    assert ops.linear_w8a8_multi(x, w, sx, ox, sw, (rows[0] - 8, sum(rows[1:]) + 8)) is None
    assert ops.linear_w8a8_multi(x[:64], w[:768], sx[:64] if per_token else sx, ox[:64] if per_token else ox, sw[:768], (256, 256, 256)) is None


def test_no_workspace_where_nothing_lives_in_it():
    """include/ffq.h: the workspace holds the weight row sums of a PERSISTENT launch and the activation row sums / flag of a launch with
    weight offsets. Below the persistent kernel's shape class (the tile kernel sums its own weight rows), or with the row sums handed in,
    a launch without weight offsets runs with workspace == NULL and gives the same bits; one that needs it says so before touching `out`."""
    from fastforward_amd import _native
    from fastforward_amd.ops._base import _ptr

    lib = _native.library()
    stream = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device=DEV).manual_seed(5)

    def call(m, n, k, workspace, rowsum=None):
        xq = torch.randint(-128, 128, (m, k), device=DEV, dtype=torch.int8, generator=g)
        wq = torch.randint(-128, 128, (n, k), device=DEV, dtype=torch.int8, generator=g)
        sx, ox = torch.tensor([0.02], device=DEV), torch.tensor([3.0], device=DEV)
        sw = torch.rand(n, device=DEV, generator=g) * 1e-2 + 1e-3
        sums = wq.sum(1, dtype=torch.int32) if rowsum else None
        out = torch.full((m, n), 7.0, device=DEV, dtype=torch.bfloat16)
        ws = torch.empty(lib.ffq_linear_w8a8_workspace_bytes(m, n, k), device=DEV, dtype=torch.uint8) if workspace else None
        rc = lib.ffq_linear_w8a8(_ptr(xq), _ptr(wq), _ptr(sums), _ptr(sx), _ptr(ox), 0, _ptr(sw), None, 1, None, 0, _ptr(out), ops._tag(torch.bfloat16),
                                 None, None, 8.0, 0, m, n, k, _ptr(ws), ws.numel() if workspace else 0, stream)
        torch.cuda.synchronize()
        return rc, out, ops.linear_w8a8(xq, wq, sx, ox, sw, None, out_dtype=torch.bfloat16)

    for m, n, k in ((16, 128, 256), (300, 520, 1040)):  # below the persistent class
        assert lib.ffq_linear_w8a8_takes_earlier(m, n, k) == 0
        rc, out, want = call(m, n, k, workspace=False)
        assert rc == 0 and torch.equal(out, want)
    m, n, k = 2048, 2048, 512  # 64 tiles of 256 x 256: the persistent kernel
    assert lib.ffq_linear_w8a8_takes_earlier(m, n, k) == 1
    rc, out, want = call(m, n, k, workspace=False, rowsum=True)
    assert rc == 0 and torch.equal(out, want)
    rc, out, _ = call(m, n, k, workspace=False)
    assert rc == 8 and bool((out == 7.0).all())  # FFQ_ERR_WORKSPACE, nothing written
