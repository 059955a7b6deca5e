"""Parity of the HIP kernels (through the C ABI of libffq_hip.so) on an MI355X.

Three layers:
  1. the golden fixtures produced by the reference (same checks as tests/test_oracle_golden.py);
  2. HIP vs the C oracle on seeded random inputs at sizes the oracle finishes in seconds,
     covering every kernel family (stream / columns / generic; rows / split / columns reductions);
  3. size-independent properties at BASELINE.json's full sizes (Llama-3-8B weight shapes and
     [8, 2048, 14336] activations): quantize->dequantize->quantize idempotence, code range,
     dynamic == static-with-minmax, per-channel == stack of per-tensor, pack round trip.
Integer codes must match bit-for-bit; dequantized floats bit-for-bit as well (same IEEE ops).
"""

import os

import pytest
import torch

import fastforward_amd as ff
import parity_cases

from conftest import load_oracle, use_backend
from fastforward_amd import _cabi, _native, llama, ops
from helpers import mismatch_report, same_with_nan

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(autouse=True)
def _backend(hip_backend):
    yield


# ---- 1. golden fixtures -------------------------------------------------------------------------
def test_known_answer_vectors():
    parity_cases.check_known_answers(DEV)


def test_ties_clamps_nan_inf_negative_zero():
    parity_cases.check_edges(DEV)


def test_random_sweeps_all_granularities():
    parity_cases.check_sweeps(DEV)


def test_mixed_dtype_sweep():
    parity_cases.check_dtype_sweep(DEV)


def test_parameters_for_range():
    parity_cases.check_ranges(DEV)


@pytest.mark.parametrize("sync_free", [False, True])
def test_running_minmax_trajectories(sync_free):
    parity_cases.check_running_minmax(DEV, sync_free=sync_free)


def test_int4_codes_and_q4_0_nibble_order():
    parity_cases.check_int4(DEV)


def test_quantized_linear_w8a8():
    parity_cases.check_linear(DEV)


def test_fused_producers_rmsnorm_silu_rope():
    parity_cases.check_producers(DEV)


def test_mm_matmul_bmm_through_the_dispatcher():
    parity_cases.check_matmul_family(DEV)


def test_weight_codes_with_row_sums():
    parity_cases.check_rowsum_fusion(DEV)


def test_attention_and_fused_output_quantizer():
    parity_cases.check_attention(DEV)


def test_quantize_by_tile_backward():
    parity_cases.check_backward(DEV)


def test_mse_grid_range_estimator():
    parity_cases.check_mse_grid(DEV)


def test_gguf_block_writers():
    parity_cases.check_gguf_blocks(DEV)


def test_smoothed_minmax_trajectories():
    parity_cases.check_smoothed_minmax(DEV)


def test_large_linear_fixture_takes_the_persistent_gemm():
    parity_cases.check_linear_large(DEV)


def test_freeze_parameters_and_fuse_qdq_weights_on_the_device():
    """SURVEY 8(f) row 1 on the GPU: both ways of folding weight quantizers into the weights (reference
    quantization/freeze.py:74-125, fuse.py:199-242) write exactly A2(A1(w)) into the parameters, in place; afterwards the
    weights sit on their grids (re-quantizing is the identity) and the forward no longer depends on the weight quantizers."""
    from fastforward_amd.quantization.freeze import freeze_parameters

    def build():
        torch.manual_seed(12)
        model = torch.nn.Sequential(torch.nn.Linear(256, 192), torch.nn.Linear(192, 64)).to(DEV, torch.bfloat16)
        ff.quantize_model(model)
        for layer, gran, bits in zip(model, (ff.PerChannel(0), ff.PerBlock(1, 64, 0)), (8, 4)):
            layer.weight_quantizer = ff.nn.LinearQuantizer(bits, granularity=gran, quantized_dtype=torch.int8, device=DEV)
        x = torch.randn(8, 256, device=DEV, dtype=torch.bfloat16)
        with torch.no_grad(), ff.strict_quantization(False), ff.estimate_ranges(model, ff.range_setting.running_minmax):
            model(x)
        return model, x

    for how in ("freeze", "fuse", "fuse_and_stub"):
        model, x = build()
        with torch.no_grad(), ff.strict_quantization(False):
            want_w = [layer.weight_quantizer(layer.weight).dequantize().clone() for layer in model]
            want_y = model(x)
        pointers = [layer.weight.data_ptr() for layer in model]
        if how == "freeze":
            with freeze_parameters(model), torch.no_grad():
                model(x)
        else:
            ff.quantization.fuse_qdq_weights(model, stub_quantizers=how == "fuse_and_stub")
        for layer, w, ptr in zip(model, want_w, pointers):
            assert torch.equal(layer.weight.detach(), w) and layer.weight.data_ptr() == ptr
            assert layer.weight_quantizer.is_stub() == (how != "fuse")
        with torch.no_grad(), ff.strict_quantization(False):
            assert torch.equal(model(x), want_y)  # same forward: the weights were already what the quantizers made of them
            if how == "fuse":
                for layer, w in zip(model, want_w):
                    assert torch.equal(layer.weight_quantizer(layer.weight).dequantize(), w)  # idempotent on the grid
    # a weight-sized tensor: Q4_0 bytes of group-32 codes == the nibbles of pack_int4 next to the fp16 scales
    torch.manual_seed(2)
    w = (torch.randn(4096, 4096, device=DEV) * 0.02).to(torch.bfloat16)
    lo, hi = ops.minmax_by_tile(w, (1, 32))
    scale, offset = ops.parameters_for_range(lo, hi, 4, True, False)
    codes = ops.quantize_by_tile(w, scale, (1, 32), 4, torch.int8)
    blocks = ops.pack_q4_0_blocks(codes.reshape(-1, 32), scale)
    assert torch.equal(blocks[:, 2:].reshape(-1), ops.pack_int4(codes, block=32))
    assert torch.equal(blocks[:, :2].reshape(-1), scale.to(torch.float16).view(torch.uint8))


def test_gptq_block_kernel():
    parity_cases.check_gptq(DEV, exact=False)


@pytest.fixture()
def generic_kernels(hip_lib):
    """switch(on): route A1 / A2 / A4 to the generic one-element-per-lane kernels (the library's test hook; no env vars)."""
    yield lambda on: hip_lib.ffq_force_generic_kernels(int(on))
    hip_lib.ffq_force_generic_kernels(0)


def test_golden_sweeps_with_ieee_division_kernels(generic_kernels):
    """The generic kernels always use the compiler's IEEE division sequence."""
    generic_kernels(True)
    parity_cases.check_sweeps(DEV, name_filter=lambda n: "sweep3d" in n)


def test_fast_division_is_bit_identical_to_ieee_division(generic_kernels):
    """The streaming kernels' FMA-iteration divide vs the compiler's IEEE divide (generic kernel).

    Quotients are placed in [2^18, 2^25) where one ulp is 1/32 ... 2, so that a last-bit error in
    x / s changes round(x / s) — the subsequent rounding cannot hide it. 32 M (x, s) pairs with a
    distinct scale per 8 elements, plus raw bit patterns (denormals, huge, tiny, Inf, NaN, -0.0).
    """
    g = torch.Generator(device=DEV).manual_seed(2024)
    n = 4 << 20
    regimes = []
    for lo_exp, hi_exp in ((-30, 30), (-3, 3), (-39, -30), (30, 39)):
        s = torch.exp2(torch.empty(n, device=DEV).uniform_(lo_exp, hi_exp, generator=g)) * torch.empty(n, device=DEV).uniform_(1, 2, generator=g)
        q = torch.exp2(torch.empty(n, 8, device=DEV).uniform_(18, 25, generator=g)) * torch.where(torch.rand(n, 8, device=DEV, generator=g) < 0.5, -1.0, 1.0)
        regimes.append((q * s[:, None], s))
    # mantissa-exhaustive-ish: s with all-ones / all-zero / random mantissas
    bits = torch.randint(0x3F000000, 0x40800000, (n,), device=DEV, generator=g, dtype=torch.int32)
    bits[::3] |= 0x007FFFFF
    s = bits.view(torch.float32)
    x = torch.randint(0x49000000, 0x4B800000, (n, 8), device=DEV, generator=g, dtype=torch.int32).view(torch.float32)
    regimes.append((x, s))
    # arbitrary bit patterns for x (incl. denormal / Inf / NaN / -0.0), ordinary scales
    x = torch.randint(-(2**31), 2**31 - 1, (n, 8), device=DEV, generator=g, dtype=torch.int64).to(torch.int32).view(torch.float32)
    regimes.append((x, torch.rand(n, device=DEV, generator=g) + 0.01))
    # arbitrary bit patterns for the scale as well (zero, denormal, huge, NaN scales take the IEEE path)
    sb = torch.randint(-(2**31), 2**31 - 1, (n,), device=DEV, generator=g, dtype=torch.int64).to(torch.int32).view(torch.float32)
    regimes.append((torch.randn(n, 8, device=DEV, generator=g), sb))
    for x, s in regimes:
        for offset in (None, torch.full_like(s, 3.0)):
            generic_kernels(False)
            fast = ops.quantize_by_tile(x, s, (1, 8), 26, torch.int32, offset)
            fast_f = ops.quantize_by_tile(x, s, (1, 8), 25, torch.float32, offset)
            generic_kernels(True)
            slow = ops.quantize_by_tile(x, s, (1, 8), 26, torch.int32, offset)
            slow_f = ops.quantize_by_tile(x, s, (1, 8), 25, torch.float32, offset)
            assert torch.equal(fast, slow), mismatch_report(fast.cpu(), slow.cpu())
            assert same_with_nan(fast_f.cpu(), slow_f.cpu()), mismatch_report(fast_f.cpu(), slow_f.cpu())


def test_fast_chunk_arithmetic_equals_the_reference_chain(generic_kernels):
    """Byte containers take packed arithmetic on ordinary chunks (csrc/ffq_affine.h: pairs of elements through v_pk_fma_f32, no
    window test per element, clamp before a round-and-convert magic add, a NaN self-check per chunk) — against the generic kernel's
    IEEE division + round + clamp + cast on quotients placed ON and next to the ties k + 0.5 (one ulp either side), at the clamp
    bounds, with every scale regime of the Markstein window, integer offsets up to 2^20, raw bit patterns for x (Inf, NaN,
    denormals, -0.0: the self-check sends those chunks to the reference chain) and for the scale."""
    g = torch.Generator(device=DEV).manual_seed(77)
    n = 1 << 20
    cases = []
    for lo_exp, hi_exp in ((-3, 3), (-30, 30), (-39.9, -30), (30, 39.9)):
        s = torch.exp2(torch.empty(n, device=DEV).uniform_(lo_exp, hi_exp, generator=g)) * torch.empty(n, device=DEV).uniform_(1, 2, generator=g)
        k = torch.randint(-140, 141, (n, 16), device=DEV, generator=g).float() + 0.5
        ulps = torch.randint(-2, 3, (n, 16), device=DEV, generator=g, dtype=torch.int32)
        q = (k.view(torch.int32) + ulps).view(torch.float32)          # k + 0.5 and its neighbours
        q[:, 15] = torch.empty(n, device=DEV).uniform_(-300, 300, generator=g)
        cases.append((q * s[:, None], s))
    x = torch.randint(-(2**31), 2**31 - 1, (n, 16), device=DEV, generator=g, dtype=torch.int64).to(torch.int32).view(torch.float32)
    cases.append((x, torch.rand(n, device=DEV, generator=g) + 0.01))
    sparse = torch.randn(n, 16, device=DEV, generator=g) * 40
    sparse[::7, 3] = float("inf")
    sparse[::11, 5] = float("nan")
    sparse[::13, 0] = -0.0
    sparse[::17, 9] = 3.0e38
    cases.append((sparse, torch.rand(n, device=DEV, generator=g) * 1e-3 + 0.3))
    sb = torch.randint(-(2**31), 2**31 - 1, (n,), device=DEV, generator=g, dtype=torch.int64).to(torch.int32).view(torch.float32)
    cases.append((torch.randn(n, 16, device=DEV, generator=g), sb))
    for x, s in cases:
        for offset in (None, torch.full_like(s, 3.0), torch.randint(-(1 << 20), 1 << 20, s.shape, device=DEV, generator=g).float()):
            for bits in (8, 4):
                for data in (x, x.to(torch.bfloat16)):
                    generic_kernels(False)
                    fast = ops.quantize_by_tile(data, s, (1, 16), bits, torch.int8, offset)
                    generic_kernels(True)
                    slow = ops.quantize_by_tile(data, s, (1, 16), bits, torch.int8, offset)
                    assert torch.equal(fast, slow), mismatch_report(fast.cpu(), slow.cpu())


# ---- 2. HIP vs oracle on seeded inputs ------------------------------------------------------------
def _both(fn):
    """Run fn(device) with the HIP backend on cuda and with the oracle on cpu."""
    got = fn(DEV)
    with use_backend(load_oracle()):
        want = fn("cpu")
    return got, want


SHAPES_AND_GRANULARITIES = [
    ((7, 33), ff.PerTensor()),                      # scalar layout with a tail (231 % 8 != 0)
    ((1,), ff.PerTensor()),
    ((64, 1000), ff.PerChannel(0)),                 # rows, 1000 % 8 == 0
    ((64, 1001), ff.PerChannel(0)),                 # rows that split chunks -> generic
    ((512, 4096), ff.PerChannel(0)),
    ((96, 2048), ff.PerChannel(-1)),                # columns kernel
    ((33, 72), ff.PerChannel(-1)),
    ((10, 12, 64), ff.PerChannel(1)),               # channel with inner 64
    ((10, 12, 64), ff.PerChannel((0, 2))),          # generic
    ((128, 1024), ff.PerBlock(1, 128, 0)),          # group 128
    ((128, 1024), ff.PerBlock(1, 32, 0)),
    ((6, 40, 96), ff.PerTile((3, 8, 32))),          # generic tiles
    ((4, 128, 4096), ff.PerChannel((0, 1))),        # per-token activations
    ((3, 5, 7), ff.PerTile((1, 1, 1))),             # per element
]


@pytest.mark.parametrize("shape,gran", SHAPES_AND_GRANULARITIES, ids=lambda v: str(v))
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32, torch.float16])
@pytest.mark.parametrize("num_bits,symmetric", [(8, True), (8, False), (4, False), (3, True)])
def test_static_quantize_dequantize_matches_oracle(shape, gran, dtype, num_bits, symmetric):
    g = torch.Generator().manual_seed(hash((shape, num_bits, symmetric)) & 0xFFFF)
    x = (torch.randn(*shape, generator=g) * 2).to(dtype)
    x.view(-1)[0] = float("nan")
    if x.numel() > 3:
        x.view(-1)[1] = float("inf")
        x.view(-1)[2] = -0.0
    n = gran.parameter_dimensionality(x.shape)
    scale = torch.rand(n, generator=g) * 0.05 + 0.01
    offset = None if symmetric else torch.randn(n, generator=g) * 20

    def run(device):
        outs = []
        for qdt in (None, torch.int8, torch.int32):
            q = ff.quantization.affine.quantize_per_granularity(
                x.to(device), scale.to(device), None if offset is None else offset.to(device), gran, num_bits, qdt
            )
            outs += [q.raw_data.cpu(), q.dequantize().cpu()]
        return outs

    got, want = _both(run)
    for a, b in zip(got, want):
        assert a.dtype == b.dtype
        assert same_with_nan(a, b), mismatch_report(a, b)


@pytest.mark.parametrize("shape,gran", SHAPES_AND_GRANULARITIES, ids=lambda v: str(v))
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_minmax_and_dynamic_quantize_match_oracle(shape, gran, dtype):
    g = torch.Generator().manual_seed(len(shape) * 131 + shape[0])
    x = (torch.randn(*shape, generator=g) * 3).to(dtype)

    def run(device):
        xd = x.to(device)
        tile = gran.tile_size(xd.shape)
        tile = xd.shape if isinstance(tile, str) else tile
        lo, hi = ops.minmax_by_tile(xd, tile)
        outs = [lo.cpu(), hi.cpu()]
        lo2, hi2 = (lo * 0.5).contiguous(), (hi * 0.5).contiguous()
        flags = torch.zeros(1, dtype=torch.int32, device=device)
        ops.minmax_by_tile(xd, tile, running_min=lo2, running_max=hi2, status_flags=flags)
        outs += [lo2.cpu(), hi2.cpu(), flags.cpu()]
        for symmetric in (True, False):
            q = ff.quantization.affine.dynamic.quantize_per_granularity(xd, gran, 8, symmetric=symmetric, output_dtype=torch.int8)
            p = q.quantization_context.quantization_params
            outs += [q.raw_data.cpu(), p.scale.cpu(), p.offset.cpu(), q.dequantize().cpu()]
        return outs

    got, want = _both(run)
    for a, b in zip(got, want):
        assert same_with_nan(a, b), mismatch_report(a, b)


DYNAMIC_ONE_LAUNCH_CASES = [
    # shape, granularity                      what it exercises
    ((4, 128, 4096), ff.PerChannel((0, 1))),  # per-token activations: one 256-lane block per row
    ((96, 14336), ff.PerChannel(0)),          # 896 chunks of 16: four chunks per lane, idle lanes in the last round
    ((40, 16384), ff.PerChannel(0)),          # the longest run the one-launch form takes
    ((24, 32768), ff.PerChannel(0)),          # beyond it: composed A4 -> A5 -> A1
    ((64, 1024), ff.PerBlock(1, 128, 0)),     # group 128: 8 lanes per tile, 32 tiles per block
    ((64, 1024), ff.PerBlock(1, 32, 0)),      # 2 lanes per tile
    ((33, 16), ff.PerChannel(0)),             # one chunk per tile, a partly filled last block
    ((5, 2000), ff.PerChannel(0)),            # 125 chunks: 64 lanes x 2
    ((6, 40), ff.PerChannel(0)),              # 40 % 16 != 0 for byte containers (composed), % 8 == 0 for wide ones (one launch)
]


@pytest.mark.parametrize("shape,gran", DYNAMIC_ONE_LAUNCH_CASES, ids=lambda v: str(v))
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32, torch.float16])
def test_one_launch_dynamic_quantize_matches_oracle_and_the_composed_form(shape, gran, dtype):
    """A3 in one launch (contiguous-run tiles, no global one-sided decision) == the oracle's restatement of
    quantize_dynamic_by_tile_impl (_quantizer_impl.py:243-285) == the composed A4 -> A5 -> A1 form, bit for bit: codes, scales, offsets.
    Rows holding NaN, +-Inf, a constant (eps-clamped scale) and only zeros (0 / 0 in the symmetric branch) included."""
    g = torch.Generator().manual_seed(sum(shape) * 7 + len(shape))
    x = (torch.randn(*shape, generator=g) * 3).to(dtype)
    rows = x.view(-1, shape[-1]) if isinstance(gran, ff.PerChannel) else x.view(-1, gran.tile_size(x.shape)[-1])
    if rows.shape[0] > 4:
        rows[1, 0] = float("nan")
        rows[2, -1] = float("inf")
        rows[3, :] = 1.5
        rows[4, :] = 0.0
    rows[0, :] = rows[0, :].abs()  # a non-negative tile: one-sided only if EVERY tile were

    def run(device):
        xd = x.to(device)
        outs = []
        for symmetric, one_sided in ((False, True), (True, False), (True, True)):
            for qdt in (torch.int8, None, torch.bfloat16 if dtype == torch.bfloat16 else torch.float32):
                q = ff.quantization.affine.dynamic.quantize_per_granularity(xd, gran, 8, symmetric=symmetric, allow_one_sided=one_sided, output_dtype=qdt)
                p = q.quantization_context.quantization_params
                outs += [q.raw_data.cpu(), p.scale.cpu(), p.offset.cpu()]
        q4 = ff.quantization.affine.dynamic.quantize_per_granularity(xd, gran, 4, symmetric=False, output_dtype=torch.int8)
        outs += [q4.raw_data.cpu(), q4.quantization_context.quantization_params.scale.cpu()]
        return outs

    got, want = _both(run)
    lib = _native.library()
    previous = lib.ffq_force_generic_kernels(1)  # the composed form on the element-wise kernel family
    try:
        composed = run(DEV)
    finally:
        lib.ffq_force_generic_kernels(previous)
    for a, b, c in zip(got, want, composed):
        assert a.dtype == b.dtype
        assert same_with_nan(a, b), mismatch_report(a, b)
        assert same_with_nan(a, c), mismatch_report(a, c)


@pytest.mark.parametrize("shape,gran", [((4, 96, 4096), ff.PerChannel((0, 1))), ((300, 1024), ff.PerBlock(1, 128, 0)), ((40, 16384), ff.PerChannel(0)), ((3000, 16), ff.PerChannel(0))], ids=str)
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_symmetric_one_sided_dynamic_quantize_in_two_launches(shape, gran, dtype):
    """symmetric AND allow_one_sided (range.py:100: one-sided iff the smallest minimum of ALL tiles is >= 0): the first launch finishes
    every tile whose own minimum is negative, the second settles the rest with the verdict. Against the oracle and the composed form,
    bit for bit, for: every tile non-negative (one-sided, every tile settled late), one negative element in the LAST tile, in the
    first tile, a NaN in one tile of non-negative data (NaN >= 0 is False: two-sided), minima of -0.0 and +0.0, mixed signs everywhere.
    The two ticket words are zero after every call and the result survives 20 repeats."""
    g = torch.Generator().manual_seed(sum(shape))
    base = (torch.randn(*shape, generator=g) * 3).to(dtype)
    flat_len = base.numel()

    def variant(kind):
        x = base.abs().clone() if kind != "mixed" else base.clone()
        f = x.view(-1)
        if kind == "late_negative":
            f[flat_len - 1] = -0.125
        elif kind == "early_negative":
            f[0] = -7.0
        elif kind == "nan":
            f[flat_len // 2] = float("nan")
        elif kind == "zeros":
            f[: shape[-1]] = 0.0
            f[shape[-1]] = -0.0
        return x

    kinds = ("non_negative", "late_negative", "early_negative", "nan", "zeros", "mixed")

    def run(device):
        outs = []
        for kind in kinds:
            xd = variant(kind).to(device)
            for qdt in (torch.int8, None):
                q = ff.quantization.affine.dynamic.quantize_per_granularity(xd, gran, 8, symmetric=True, allow_one_sided=True, output_dtype=qdt)
                p = q.quantization_context.quantization_params
                outs += [q.raw_data.cpu(), p.scale.cpu(), p.offset.cpu()]
        return outs

    got, want = _both(run)
    lib = _native.library()
    previous = lib.ffq_force_generic_kernels(1)  # the composed A4 -> A5 -> A1 form
    try:
        composed = run(DEV)
    finally:
        lib.ffq_force_generic_kernels(previous)
    for i, (a, b, c) in enumerate(zip(got, want, composed)):
        assert same_with_nan(a, b), (kinds[i // 6], mismatch_report(a, b))
        assert same_with_nan(a, c), (kinds[i // 6], mismatch_report(a, c))
    for buf in ops._TICKETS.values():
        assert int(buf.abs().sum()) == 0
    xd = variant("late_negative").to(DEV)
    first = ops.quantize_dynamic_by_tile(xd, gran.tile_size(xd.shape), 8, True, True, torch.int8)
    for _ in range(20):
        again = ops.quantize_dynamic_by_tile(xd, gran.tile_size(xd.shape), 8, True, True, torch.int8)
        assert all(same_with_nan(u.cpu(), v.cpu()) for u, v in zip(first, again))


RUNNING_QUANTIZE_CASES = [
    ((512, 4096), ff.PerChannel(0)),         # a weight: one 256-lane block per two rows
    ((96, 14336), ff.PerChannel(0)),         # down_proj-shaped rows
    ((64, 1024), ff.PerBlock(1, 128, 0)),    # group 128
    ((4, 64, 2048), ff.PerChannel((0, 1))),  # per-token activations
    ((33, 16), ff.PerChannel(0)),            # one chunk per tile
    ((64, 1001), ff.PerChannel(0)),          # rows that split chunks: declined, the two steps run
    ((7, 33), ff.PerTensor()),               # one tile: declined
]


@pytest.mark.parametrize("shape,gran", RUNNING_QUANTIZE_CASES, ids=str)
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("symmetric,one_sided", [(True, True), (True, False), (False, True)])
def test_estimator_step_and_quantize_in_one_pass(shape, gran, dtype, symmetric, one_sided):
    """``estimate_ranges(model, running_minmax, sync_free=True)`` on a LinearQuantizer: three batches through the quantizer's forward
    (RunningMinMax step + the quantizer's own forward, reference common.py:218-238). HIP with the one-pass kernel
    (ops.running_minmax_quantize: per-row tilings) == HIP with a second override in the chain (the two steps) == the oracle: codes
    of every batch, running min / max, scale, offset, status flags. Batches: mixed signs; non-negative in every tile (the one-sided
    verdict flips when a later batch brings a negative value); an Inf and a NaN in the last one."""
    g = torch.Generator().manual_seed(sum(shape) + int(symmetric) * 2 + int(one_sided))
    batches = [(torch.randn(*shape, generator=g) * 2).abs().to(dtype), (torch.randn(*shape, generator=g) * 3).abs().to(dtype), (torch.randn(*shape, generator=g)).to(dtype)]
    batches[1].view(-1)[-1] = -0.5
    batches[2].view(-1)[3] = float("inf")
    batches[2].view(-1)[-2] = float("nan")

    def run(device, two_steps=False):
        quantizer = ff.nn.LinearQuantizer(8, symmetric=symmetric, allow_one_sided=one_sided, granularity=gran, quantized_dtype=torch.int8, device=device)
        outs = []
        extra = quantizer.register_override(lambda _ctx, fn, args, kwargs: fn(*args, **kwargs)) if two_steps else None
        estimators = []
        try:
            with ff.estimate_ranges(quantizer, ff.range_setting.running_minmax, sync_free=True):
                estimators = [fn for fn in quantizer.overrides if isinstance(fn, ff.range_setting.minmax.RunningMinMaxEstimator)]
                for x in batches:
                    q = quantizer(x.to(device))
                    p = q.quantization_context.quantization_params
                    assert p.dequantize_dtype == dtype
                    outs += [q.raw_data.cpu(), quantizer.scale.detach().cpu().clone(), (quantizer.offset.detach().cpu().clone() if quantizer.offset is not None else torch.zeros(1))]
                    outs += [estimators[0].min.float().cpu().clone(), estimators[0].max.float().cpu().clone(), estimators[0].status.cpu().clone()]
        except NotImplementedError:  # the deferred "Infinite" check at the end of the context (minmax.py:233-234): raised on every route alike
            outs.append(torch.tensor([1.0]))
        finally:
            if extra is not None:
                extra.remove()
        return outs

    got, want = _both(run)
    two = run(DEV, two_steps=True)
    assert len(got) == len(want) == len(two)
    for a, b, c in zip(got, want, two):
        assert same_with_nan(a, b), mismatch_report(a, b)
        assert same_with_nan(a, c), mismatch_report(a, c)
    for buf in ops._TICKETS.values():
        assert int(buf.abs().sum()) == 0


def test_an_estimator_subclass_that_overrides_a_step_sees_every_batch():
    """the one-pass route of RunningMinMaxEstimator.forward (estimator step + A5 + A1 in one launch) is this class's own
    two steps fused — a subclass that overrides estimate_step must be called for every batch, also in sync_free mode, and the
    quantizer's parameters and codes must still be those of the two-step form."""
    from fastforward_amd.range_setting.minmax import RunningMinMaxEstimator

    seen = []

    class Logging(RunningMinMaxEstimator):
        def estimate_step(self, quantizer, data):
            seen.append(tuple(data.shape))
            super().estimate_step(quantizer, data)

    torch.manual_seed(0)
    batches = [torch.randn(64, 1024, device=DEV).to(torch.bfloat16) * (i + 1) for i in range(3)]

    def run(estimator_cls):
        quantizer = ff.nn.LinearQuantizer(8, granularity=ff.PerChannel(0), quantized_dtype=torch.int8, device=DEV)
        handle = quantizer.register_override(estimator_cls(quantizer, sync_free=True))
        try:
            codes = [quantizer(x).raw_data.clone() for x in batches]
        finally:
            handle.remove()
        return codes, quantizer.scale.detach().clone()

    plain_codes, plain_scale = run(RunningMinMaxEstimator)
    assert not seen
    logged_codes, logged_scale = run(Logging)
    assert seen == [(64, 1024)] * 3
    assert torch.equal(plain_scale, logged_scale) and all(torch.equal(a, b) for a, b in zip(plain_codes, logged_codes))


@pytest.mark.parametrize("ntiles", [8193, 20000, 458752])
@pytest.mark.parametrize("range_dtype", [torch.float32, torch.bfloat16])
def test_parameters_for_range_grid_form_matches_oracle(ntiles, range_dtype):
    """A5 above 8192 tiles runs as a grid (two launches when the one-sided test of range.py:100 is global): same values as the
    oracle and as the one-block kernel, for two-sided, one-sided, NaN-holding and asymmetric ranges."""
    g = torch.Generator().manual_seed(ntiles)
    lo = (torch.randn(ntiles, generator=g) * 2 - 1).to(range_dtype)
    hi = (lo.float() + torch.rand(ntiles, generator=g) * 4).to(range_dtype)
    hi[5] = lo[5]  # empty interval: eps clamp
    cases = {"two_sided": (lo, hi), "one_sided": (lo.abs(), lo.abs() + (hi - lo).abs())}
    with_nan = lo.abs().clone()
    with_nan[ntiles - 3] = float("nan")
    cases["nan"] = (with_nan, cases["one_sided"][1])
    late_negative = lo.abs().clone()
    late_negative[ntiles - 1] = -0.25  # the only negative minimum sits in the last block's share
    cases["late_negative"] = (late_negative, cases["one_sided"][1])

    def run(device):
        outs = []
        for mn, mx in cases.values():
            for symmetric, one_sided in ((True, True), (True, False), (False, True)):
                for bits in (8, 4):
                    s, o = ops.parameters_for_range(mn.to(device), mx.to(device), bits, symmetric, one_sided)
                    outs += [s.cpu(), o.cpu()]
        return outs

    got, want = _both(run)
    lib = _native.library()
    previous = lib.ffq_force_generic_kernels(1)  # the one-block kernel
    try:
        one_block = run(DEV)
    finally:
        lib.ffq_force_generic_kernels(previous)
    for a, b, c in zip(got, want, one_block):
        assert same_with_nan(a, b), mismatch_report(a, b)
        assert same_with_nan(a, c), mismatch_report(a, c)


def test_minmax_flags_inf_and_nan():
    x = torch.randn(64, 256, device=DEV, dtype=torch.bfloat16)
    flags = torch.zeros(1, dtype=torch.int32, device=DEV)
    ops.minmax_by_tile(x, (1, 256), status_flags=flags)
    assert flags.item() == 0
    x[3, 5] = float("inf")
    ops.minmax_by_tile(x, (1, 256), status_flags=flags)
    assert flags.item() == ops.FLAG_INF
    x[7, 9] = float("nan")
    lo, hi = ops.minmax_by_tile(x, (1, 256), status_flags=flags)
    assert flags.item() == ops.FLAG_INF | ops.FLAG_NAN
    assert lo[7].isnan() and hi[7].isnan() and hi[3].isinf() and not lo[0].isnan()


def test_running_minmax_raises_on_infinite_like_reference():
    quantizer = ff.nn.LinearQuantizer(8, device=DEV)
    model = torch.nn.ModuleList([quantizer])
    x = torch.randn(4, 32, device=DEV)
    x[0, 0] = float("inf")
    with pytest.raises(NotImplementedError, match="Infinite"):
        with ff.estimate_ranges(model, ff.range_setting.running_minmax):
            quantizer(x)
    quantizer2 = ff.nn.LinearQuantizer(8, device=DEV)
    with pytest.raises(NotImplementedError, match="Infinite"):
        with ff.estimate_ranges(torch.nn.ModuleList([quantizer2]), ff.range_setting.running_minmax, sync_free=True):
            quantizer2(x)  # deferred: raised when the context exits


# the last three shapes have >= 64 tiles of 256 x 256 and K % 64 == 0: they take the direct-to-LDS kernel
@pytest.mark.parametrize("m,n,k", [(1, 16, 16), (96, 200, 144), (256, 384, 512), (300, 130, 4096), (17, 1000, 256), (2048, 2048, 256), (2050, 2300, 192), (4100, 1030, 1024)])
@pytest.mark.parametrize("x_per_row,w_per_row,x_off,w_off", [(0, 1, True, False), (0, 0, False, False), (1, 1, True, True), (0, 1, False, True)])
def test_w8a8_linear_exact_integer_math(m, n, k, x_per_row, w_per_row, x_off, w_off):
    """The contraction is exact: compare with an int64 matmul of the same codes, fp32 epilogue."""
    g = torch.Generator().manual_seed(m * 7 + n)
    xq = torch.randint(-128, 128, (m, k), generator=g, dtype=torch.int8)
    wq = torch.randint(-128, 128, (n, k), generator=g, dtype=torch.int8)
    sx = torch.rand(m if x_per_row else 1, generator=g) * 0.02 + 0.001
    sw = torch.rand(n if w_per_row else 1, generator=g) * 0.02 + 0.001
    ox = torch.randn(m if x_per_row else 1, generator=g) * 30 if x_off else None
    ow = torch.randn(n if w_per_row else 1, generator=g) * 3 if w_off else None
    bias = torch.randn(n, generator=g)
    y = ops.linear_w8a8(xq.to(DEV), wq.to(DEV), sx.to(DEV), None if ox is None else ox.to(DEV), sw.to(DEV),
                        None if ow is None else ow.to(DEV), bias=bias.to(DEV), out_dtype=torch.float32).cpu()
    xi = xq.double() + (torch.round(ox).double()[:, None] if ox is not None and x_per_row else (torch.round(ox).double() if ox is not None else 0))
    wi = wq.double() + (torch.round(ow).double()[:, None] if ow is not None and w_per_row else (torch.round(ow).double() if ow is not None else 0))
    ref = (xi @ wi.T) * (sx.double()[:, None] if x_per_row else sx.double()) * (sw.double()[None, :] if w_per_row else sw.double()) + bias.double()
    torch.testing.assert_close(y.double(), ref, rtol=2e-6, atol=2e-6 * float(ref.abs().max()))


def test_w8a8_linear_matches_oracle_bf16():
    g = torch.Generator().manual_seed(5)
    xq = torch.randint(-128, 128, (40, 256), generator=g, dtype=torch.int8)
    wq = torch.randint(-128, 128, (72, 256), generator=g, dtype=torch.int8)
    sx, ox = torch.tensor([0.013]), torch.tensor([-11.4])
    sw = torch.rand(72, generator=g) * 0.01 + 0.001

    def run(device):
        return [ops.linear_w8a8(xq.to(device), wq.to(device), sx.to(device), ox.to(device), sw.to(device), None, out_dtype=torch.bfloat16).cpu()]

    got, want = _both(run)
    # oracle = the reference's dequantize-then-float-GEMM semantics evaluated exactly
    torch.testing.assert_close(got[0].float(), want[0].float(), atol=1e-1, rtol=1.3e-2)


def test_fused_linear_is_dispatched_for_both_call_paths():
    lin = torch.nn.Linear(128, 64, bias=False).to(DEV, torch.bfloat16)
    x = torch.randn(8, 128, device=DEV, dtype=torch.bfloat16)
    wq = ff.quantization.affine.quantize_per_channel(lin.weight, torch.full((64,), 0.01, device=DEV), None, 0, 8, torch.int8)
    xq = ff.quantization.affine.quantize_per_tensor(x, torch.tensor([0.03], device=DEV), torch.tensor([3.0], device=DEV), 8, torch.int8)
    assert ff.dispatcher.dispatch("linear", xq, wq) is ff.fused_linear.fused_linear
    with ff.strict_quantization(False):
        a = torch.nn.functional.linear(xq, wq)           # __torch_function__ path (positional)
        b = ff.nn.functional.linear(xq, wq)              # functional path (keywords)
        ref = torch.nn.functional.linear(xq.dequantize(), wq.dequantize())
    assert torch.equal(a, b)
    torch.testing.assert_close(a.float(), ref.float(), atol=1e-1, rtol=1.3e-2)


# ---- 3. properties at full BASELINE sizes -----------------------------------------------------------
LLAMA8B_WEIGHTS = [(4096, 4096), (1024, 4096), (14336, 4096), (4096, 14336)]


@pytest.mark.parametrize("shape", LLAMA8B_WEIGHTS, ids=str)
def test_full_size_weight_properties(shape):
    torch.manual_seed(1234 + shape[0])
    w = (torch.randn(shape, device=DEV) * 0.02).to(torch.bfloat16)
    quantizer = ff.nn.LinearQuantizer(8, granularity=ff.PerChannel(0), quantized_dtype=torch.int8, device=DEV)
    with ff.estimate_ranges(torch.nn.ModuleList([quantizer]), ff.range_setting.running_minmax, sync_free=True):
        q = quantizer(w)
    codes = q.raw_data
    # symmetric per-channel 8-bit from the row min/max: every row reaches +-127/-128 exactly once at least
    assert codes.dtype == torch.int8 and int(codes.max()) == 127 or int(codes.min()) == -128
    amax = w.float().abs().amax(1)
    assert torch.equal(codes.float().abs().amax(1) >= 127, torch.ones_like(amax, dtype=torch.bool))
    # scale reproduces parameters_for_range on the row extrema
    # (computed on the CPU: torch's GPU kernel for tensor / python_scalar multiplies by the
    #  reciprocal, the reference's CPU path and our kernel divide)
    lo, hi = w.float().amin(1).cpu(), w.float().amax(1).cpu()
    assert torch.equal(quantizer.scale.detach().cpu(), torch.maximum(lo.abs() / 128, hi.abs() / 127))
    # quantize(dequantize(q)) == q  (idempotence) and |x - x^| <= scale/2 (+ bf16 rounding)
    deq = q.dequantize()
    again = quantizer(deq).raw_data
    assert torch.equal(again, codes)
    err = (w.float() - deq.float()).abs()
    bound = quantizer.scale.detach()[:, None] * 0.5 + deq.float().abs() * 2.0**-8
    assert bool((err <= bound).all())
    # bf16 container holds the same integers
    quantizer.quantized_dtype = None
    assert torch.equal(quantizer(w).raw_data.to(torch.int8), codes)
    # per-channel == stack of per-tensor on a few rows (tests/nn/test_linear_quantizer.py:246-276)
    for r in (0, shape[0] // 2, shape[0] - 1):
        row = ff.quantization.affine.quantize_per_tensor(w[r], quantizer.scale.detach()[r : r + 1], None, 8, torch.int8)
        assert torch.equal(row.raw_data, codes[r])
    # int4 pack round trip on the real shape (group 128)
    q4 = ff.quantization.affine.dynamic.quantize_per_granularity(w, ff.PerBlock(1, 128, 0), 4, symmetric=True, output_dtype=torch.int8)
    packed = ops.pack_int4(q4.raw_data, block=128)
    assert packed.numel() == w.numel() // 2
    assert torch.equal(ops.unpack_int4(packed, w.shape, torch.int8, block=128), q4.raw_data)


@pytest.mark.parametrize("hidden", [4096, 14336])
def test_full_size_activation_properties(hidden):
    torch.manual_seed(99)
    x = torch.randn(8, 2048, hidden, device=DEV, dtype=torch.bfloat16)
    quantizer = ff.nn.LinearQuantizer(8, symmetric=False, quantized_dtype=torch.int8, device=DEV)
    with ff.estimate_ranges(torch.nn.ModuleList([quantizer]), ff.range_setting.running_minmax, sync_free=True):
        q = quantizer(x)
    lo, hi = x.float().min().cpu(), x.float().max().cpu()  # CPU: true division, see above
    scale = ((hi - lo) / 255).clamp(torch.finfo(torch.float32).eps)
    assert torch.equal(quantizer.scale.detach().cpu(), scale.reshape(1))
    assert torch.equal(quantizer.offset.detach().cpu(), (lo / scale + 128).reshape(1))
    assert int(q.raw_data.min()) == -128 and int(q.raw_data.max()) == 127
    # dynamic quantization == static quantization with the min/max range (tests/quantization/test_dynamic.py:13-30)
    dyn = ff.quantization.affine.dynamic.quantize_per_tensor(x, 8, symmetric=False, output_dtype=torch.int8)
    assert torch.equal(dyn.raw_data, q.raw_data)
    # checksum of codes against a chunked recomputation with plain torch ops in fp32
    off = torch.round(quantizer.offset.detach())
    total = 0
    for chunk in x.chunk(8):  # tensor / tensor is a true division on the GPU as well
        total += int(torch.clamp(torch.round(chunk.float() / quantizer.scale.detach() - off), -128, 127).sum(dtype=torch.int64))
    assert int(q.raw_data.sum(dtype=torch.int64)) == total


# ---- producer-fused A1 (RMSNorm / SiLU*up / rotary) ------------------------------------------------
def _one_ulp(a, b, frac, max_ulps=1):
    ulps = parity_cases._ulps_bf16(a.cpu(), b.cpu())
    return int(ulps.max()) <= max_ulps and float((ulps > 0).float().mean()) < frac


@pytest.mark.parametrize("rows,cols", [(7, 16), (64, 256), (33, 1024), (19, 2064), (9, 4096), (5, 8192)])
@pytest.mark.parametrize("with_delta", [True, False])
def test_add_rmsnorm_quantize_matches_oracle(rows, cols, with_delta):
    gen = torch.Generator().manual_seed(rows * 131 + cols)
    x = (torch.randn(rows, cols, generator=gen) * 3).to(torch.bfloat16)
    delta = torch.randn(rows, cols, generator=gen).to(torch.bfloat16) if with_delta else None
    w = (1 + 0.3 * torch.randn(cols, generator=gen)).to(torch.bfloat16)
    qs = [(torch.tensor([0.021]), torch.tensor([3.5])), (torch.tensor([0.05]), None), (torch.tensor([0.021]), torch.tensor([3.5]))]
    with use_backend(load_oracle()):
        want = ops.add_rmsnorm_quantize(x, delta, w, 1e-5, qs, want_norm=True)
    dev = lambda t: None if t is None else t.to(DEV)
    got = ops.add_rmsnorm_quantize(dev(x), dev(delta), dev(w), 1e-5, [(dev(s), dev(o)) for s, o in qs], want_norm=True)
    assert torch.equal(got[0].cpu(), want[0])
    assert _one_ulp(got[1], want[1], 0.01, max_ulps=2)  # two bf16 roundings in sequence: see the full-size test
    for (s, o), codes in zip(qs, got[2]):
        assert torch.equal(codes, ops.quantize_by_tile(got[1], dev(s), got[1].shape, 8, torch.int8, dev(o)))
    assert torch.equal(got[2][0], got[2][2])


def test_silu_mul_and_rope_match_oracle():
    gen = torch.Generator().manual_seed(99)
    gate = (torch.randn(37, 896, generator=gen) * 4).to(torch.bfloat16)
    up = torch.randn(37, 896, generator=gen).to(torch.bfloat16)
    q = [(torch.tensor([0.07]), torch.tensor([-11.0]))]
    with use_backend(load_oracle()):
        want = ops.silu_mul_quantize(gate, up, q, want_product=True)
    got = ops.silu_mul_quantize(gate.to(DEV), up.to(DEV), [(q[0][0].to(DEV), q[0][1].to(DEV))], want_product=True)
    assert _one_ulp(got[0], want[0], 0.01)
    same = parity_cases._ulps_bf16(got[0].cpu(), want[0]) == 0
    assert torch.equal(got[1][0].cpu()[same], want[1][0][same])
    # against ATen's own silu and multiply on the device: identical
    ref = torch.nn.functional.silu(gate.to(DEV)) * up.to(DEV)
    assert torch.equal(got[0], ref)

    for b, s, hq, hk, d in [(2, 24, 8, 2, 128), (1, 5, 3, 1, 64), (3, 16, 4, 4, 16)]:
        qq = torch.randn(b, s, hq * d, generator=gen).to(torch.bfloat16)
        kk = torch.randn(b, s, hk * d, generator=gen).to(torch.bfloat16)
        cos, sin = llama.rotary_tables(s, d, 500000.0, "cpu", torch.bfloat16)
        with use_backend(load_oracle()):
            wq, wk = qq.clone(), kk.clone()
            ops.rope_(wq, wk, cos, sin, d)
        gq, gk = qq.to(DEV), kk.to(DEV)
        ops.rope_(gq, gk, cos.to(DEV), sin.to(DEV), d)
        assert torch.equal(gq.cpu(), wq) and torch.equal(gk.cpu(), wk)


def test_attention_rescale_and_mask_paths_against_float64():
    """The online softmax rescales its accumulators only when a row maximum grows by more than 2^8 (wave-uniform, rare):
    keys that dominate at chosen positions force that branch in the middle of a sequence, for some rows of a wave and not
    for others; a dominant FIRST key makes every later tile take the no-rescale path; large scores exercise the masked
    (-inf) entries of the diagonal tiles. Checked against float64 on the same bf16 inputs, and against the oracle's eager chain."""
    torch.manual_seed(21)
    d, b, s_, heads, kv_heads = 128, 1, 512, 4, 2
    for spikes, gain in (((70, 200, 450), 6.0), ((0,), 8.0), ((63, 64, 255, 256, 511), 5.0), ((), 3.0)):
        q = torch.randn(b, s_, heads * d).to(torch.bfloat16)
        k = torch.randn(b, s_, kv_heads * d).to(torch.bfloat16)
        v = torch.randn(b, s_, kv_heads * d).to(torch.bfloat16)
        if not spikes:
            q, k = (q * gain).to(torch.bfloat16), (k * gain).to(torch.bfloat16)
        for pos in spikes:  # a key with `gain` times the norm: about half of the later rows see their maximum jump here
            k[0, pos] = (k[0, pos].float() * gain).to(torch.bfloat16)
        want = parity_cases.attention_reference64(q, k, v, d, True)
        got, _ = ops.attention(q.to(DEV), k.to(DEV), v.to(DEV), d, causal=True)
        err = float((got.cpu().double() - want).abs().max())
        assert err <= parity_cases.ATTENTION_ATOL, (spikes, err)
        with use_backend(load_oracle()):
            chain, _ = ops.attention(q, k, v, d, causal=True)
        err_chain = float((chain.double() - want).abs().max())
        assert err <= err_chain + 2.0**-9, (spikes, err, err_chain)


def test_full_size_attention_properties():
    """Llama-3-8B attention shape (B=8, S=2048, 32 query heads, 8 kv heads, D=128): codes == A1 of the context this call
    produced; the context within one bf16 ulp-of-the-largest-output of torch's SDPA on the device; causality — changing
    keys / values at positions > t leaves rows <= t bit-identical; batch / head independence — a sub-batch gives the same rows."""
    torch.manual_seed(9)
    b, s_, heads, kv_heads, d = 8, 2048, 32, 8, 128
    q = torch.randn(b, s_, heads * d, device=DEV, dtype=torch.bfloat16)
    k = torch.randn(b, s_, kv_heads * d, device=DEV, dtype=torch.bfloat16)
    v = torch.randn(b, s_, kv_heads * d, device=DEV, dtype=torch.bfloat16)
    sc, of = torch.tensor([0.03], device=DEV), torch.tensor([-3.0], device=DEV)
    ctx, codes = ops.attention(q, k, v, d, causal=True, quantizer=(sc, of))
    assert torch.equal(codes, ops.quantize_by_tile(ctx, sc, ctx.shape, 8, torch.int8, of))
    sdpa = torch.nn.functional.scaled_dot_product_attention(
        q.view(b, s_, heads, d).transpose(1, 2), k.view(b, s_, kv_heads, d).transpose(1, 2), v.view(b, s_, kv_heads, d).transpose(1, 2),
        is_causal=True, enable_gqa=True).transpose(1, 2).reshape(b, s_, -1)
    assert float((ctx.float() - sdpa.float()).abs().max()) <= parity_cases.ATTENTION_ATOL
    t = 1000
    k2, v2 = k.clone(), v.clone()
    k2[:, t + 1:] = torch.randn_like(k2[:, t + 1:]) * 3
    v2[:, t + 1:] = torch.randn_like(v2[:, t + 1:])
    ctx2, _ = ops.attention(q, k2, v2, d, causal=True)
    assert torch.equal(ctx2[:, : t + 1], ctx[:, : t + 1]) and not torch.equal(ctx2[:, t + 1:], ctx[:, t + 1:])
    sub, _ = ops.attention(q[2:4].contiguous(), k[2:4].contiguous(), v[2:4].contiguous(), d, causal=True)
    assert torch.equal(sub, ctx[2:4])


def test_full_size_producers_properties():
    """Llama-3-8B activation sizes ([8, 2048, 4096] through RMSNorm, [8, 2048, 14336] through SiLU*up):
    codes == A1 of the produced value, the produced value within one bf16 ulp of the ATen chain on the
    device, rotary embedding identical to the ATen chain."""
    torch.manual_seed(5)
    x = torch.randn(8, 2048, 4096, device=DEV, dtype=torch.bfloat16)
    delta = torch.randn_like(x) * 0.3
    w = (1 + 0.1 * torch.randn(4096, device=DEV)).to(torch.bfloat16)
    s, o = torch.tensor([0.04], device=DEV), torch.tensor([2.0], device=DEV)
    total, norm, codes = ops.add_rmsnorm_quantize(x, delta, w, 1e-5, [(s, o)], want_norm=True)
    assert torch.equal(total, x + delta)
    ref = llama.LlamaRMSNorm.forward(type("N", (), {"weight": w, "variance_epsilon": 1e-5})(), total)
    # z = bf16(w * bf16(h * r)): a last-bit difference of r (summation order of the mean) moves the inner
    # rounding by one ulp on a few elements per million, and the outer rounding can double it. Measured
    # on the MI355X against a float64 evaluation: this kernel 4.9e-6 of the elements differ (max 2 ulp),
    # ATen's own chain 4.2e-6 (max 2 ulp).
    assert _one_ulp(norm, ref, 1e-4, max_ulps=2)
    assert torch.equal(codes[0], ops.quantize_by_tile(norm, s, norm.shape, 8, torch.int8, o))
    del x, delta, total, norm, ref, codes
    gate = torch.randn(8, 2048, 14336, device=DEV, dtype=torch.bfloat16) * 2
    up = torch.randn_like(gate)
    product, codes = ops.silu_mul_quantize(gate, up, [(s, o)], want_product=True)
    assert torch.equal(product, torch.nn.functional.silu(gate) * up)
    assert torch.equal(codes[0], ops.quantize_by_tile(product, s, product.shape, 8, torch.int8, o))
    del gate, up, product, codes
    q = torch.randn(8, 2048, 4096, device=DEV, dtype=torch.bfloat16)
    k = torch.randn(8, 2048, 1024, device=DEV, dtype=torch.bfloat16)
    cos, sin = llama.rotary_tables(2048, 128, 500000.0, DEV, torch.bfloat16)
    qh, kh = q.view(8, 2048, 32, 128).transpose(1, 2), k.view(8, 2048, 8, 128).transpose(1, 2)
    want_q = (qh * cos + llama._rotate_half(qh) * sin).transpose(1, 2).reshape(8, 2048, -1)
    want_k = (kh * cos + llama._rotate_half(kh) * sin).transpose(1, 2).reshape(8, 2048, -1)
    ops.rope_(q, k, cos, sin, 128)
    assert torch.equal(q, want_q) and torch.equal(k, want_k)


# ---- A8: backward ----------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape,tile", [((4096, 4096), (1, 4096)), ((8, 2048, 4096), (8, 2048, 4096)), ((4096, 4096), (1, 128)), ((1000, 1000), (1000, 1000)), ((24, 1024), (1, 1024))])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_backward_kernel_matches_composite_formulas(shape, tile, dtype):
    """HIP kernel vs the op-for-op composite on the device at Llama sizes: dinput identical, per-tile sums within
    the fp32 summation tolerance; the straight-through property dinput == grad where nothing clips."""
    torch.manual_seed(3)
    x = (torch.randn(shape, device=DEV) * 2).to(dtype)
    g = torch.randn(shape, device=DEV).to(dtype)
    ntiles = 1
    for s_, t_ in zip(shape, tile):
        ntiles *= s_ // t_
    scale = torch.rand(ntiles, device=DEV) * 0.05 + 0.02
    offset = torch.rand(ntiles, device=DEV) * 6 - 3
    got = ops.quantize_by_tile_backward(x, g, scale, tile, 8.0, offset)
    want = ops._quantize_by_tile_backward_composite(x, g, scale, tile, 8.0, offset)
    assert torch.equal(got[0], want[0])
    from fastforward_amd.quantization.tiled_tensor import tiles_to_rows
    mag = tiles_to_rows(g.float().abs(), torch.Size(tile)).sum(1)  # every term is bounded by (|bound| + |offset|) |g|
    assert bool(((got[1] - want[1]).abs() <= 4e-6 * 132 * mag + 1e-6).all())
    assert bool(((got[2] - want[2]).abs() <= 4e-6 * 0.07 * mag + 1e-6).all())
    wide = ops.quantize_by_tile_backward(x, g, scale * 1e3, tile, 8.0, offset)  # grid wide enough: nothing clips
    assert torch.equal(wide[0], g) and float(wide[2].abs().max()) == 0.0


def test_full_size_w4_fused_pack_paths():
    """[14336, 4096] bf16, PerBlock(128): fused quantize+pack == A1 then A7, fused unpack+dequantize == A7 then A2,
    and the round trip is idempotent (re-quantizing the dequantized weights gives the same nibbles)."""
    torch.manual_seed(9)
    w = (torch.randn(14336, 4096, device=DEV) * 0.02).to(torch.bfloat16)
    tile = (1, 128)
    lo, hi = ops.minmax_by_tile(w, tile)
    scale, offset = ops.parameters_for_range(lo, hi, 4, True, True)
    packed = ops.quantize_pack_int4(w, scale, tile, offset, block=128)
    assert torch.equal(packed, ops.pack_int4(ops.quantize_by_tile(w, scale, tile, 4, torch.int8, offset), block=128))
    deq = ops.unpack_dequantize_int4(packed, scale, w.shape, tile, offset, block=128)
    assert torch.equal(deq, ops.dequantize_by_tile(ops.unpack_int4(packed, w.shape, torch.int8, block=128), scale, tile, offset, torch.bfloat16))
    assert torch.equal(ops.quantize_pack_int4(deq, scale, tile, offset, block=128), packed)


# ---- min-error grid: fused kernel vs the candidate-by-candidate loop -------------------------------------
@pytest.mark.parametrize("shape,gran,symmetric", [((4096, 4096), ff.PerChannel(0), True), ((4, 512, 4096), ff.PerTensor(), False), ((1024, 1024), ff.PerBlock(1, 128, 0), True)], ids=str)
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_grid_sqerror_kernel_matches_candidate_loop(shape, gran, symmetric, dtype):
    from fastforward_amd.range_setting.min_error import _MinAvgErrorGridEstimator, mse_error

    torch.manual_seed(11)
    x = torch.randn(shape, device=DEV).to(dtype)
    results = []
    for fused in (True, False):
        q = ff.nn.LinearQuantizer(4, granularity=gran, symmetric=symmetric, device=DEV)
        error_fn = mse_error if fused else (lambda a, b: mse_error(a, b))  # a different callable forces the reference loop
        with ff.estimate_ranges(q, ff.range_setting.mse_grid, num_candidates=12, error_fn=error_fn):
            est = next(o for o in q.overrides if isinstance(o, _MinAvgErrorGridEstimator))
            q(x)
            assert est.used_fused_kernel == fused
            results.append(est.cumulative_error.float().clone())
    rtol = 2e-2 if dtype == torch.bfloat16 else 1e-5
    torch.testing.assert_close(results[0], results[1], rtol=rtol, atol=1e-12)


# ---- gate_proj + up_proj + SiLU*up + quantize in one launch ------------------------------------------------------
@pytest.mark.parametrize("m,n,k", [(256, 128, 256), (300, 256, 512), (1000, 896, 256), (2048, 1024, 4096), (77, 384, 1024)])
@pytest.mark.parametrize("x_off", [True, False])
def test_fused_gate_up_equals_three_launch_path(m, n, k, x_off):
    gen = torch.Generator().manual_seed(m * 7 + n + k)
    xq = torch.randint(-128, 128, (m, k), generator=gen, dtype=torch.int8).to(DEV)
    gq = torch.randint(-128, 128, (n, k), generator=gen, dtype=torch.int8).to(DEV)
    uq = torch.randint(-128, 128, (n, k), generator=gen, dtype=torch.int8).to(DEV)
    sx = torch.tensor([0.011], device=DEV)
    ox = torch.tensor([-3.0], device=DEV) if x_off else None
    sg = (torch.rand(n, generator=gen) * 2e-4 + 1e-4).to(DEV)
    su = (torch.rand(n, generator=gen) * 2e-4 + 1e-4).to(DEV)
    so, oo = torch.tensor([0.02], device=DEV), torch.tensor([5.0], device=DEV)
    fused = ops.mlp_gate_up_w8a8(xq, gq, uq, sx, ox, sg, su, so, oo, 8)
    gate = ops.linear_w8a8(xq, gq, sx, ox, sg, None, out_dtype=torch.bfloat16)
    up = ops.linear_w8a8(xq, uq, sx, ox, su, None, out_dtype=torch.bfloat16)
    _, (want,) = ops.silu_mul_quantize(gate, up, [(so, oo)], 8)
    assert fused is not None and torch.equal(fused, want), f"{int((fused != want).sum())} of {want.numel()} codes differ"
    assert int(want.float().std()) > 0  # the comparison is not about a saturated tensor


def test_fused_gate_up_at_llama_size_and_in_the_forward():
    torch.manual_seed(21)
    t, n, k = 8 * 2048, 14336, 4096
    xq = torch.randint(-128, 128, (t, k), device=DEV, dtype=torch.int8)
    gq = torch.randint(-128, 128, (n, k), device=DEV, dtype=torch.int8)
    uq = torch.randint(-128, 128, (n, k), device=DEV, dtype=torch.int8)
    sx, ox = torch.tensor([0.02], device=DEV), torch.tensor([4.0], device=DEV)
    sg = torch.rand(n, device=DEV) * 1e-5 + 2e-5
    su = torch.rand(n, device=DEV) * 1e-5 + 2e-5
    so, oo = torch.tensor([0.004], device=DEV), torch.tensor([-9.0], device=DEV)
    fused = ops.mlp_gate_up_w8a8(xq, gq, uq, sx, ox, sg, su, so, oo, 8)
    gate = ops.linear_w8a8(xq, gq, sx, ox, sg, None, out_dtype=torch.bfloat16)
    up = ops.linear_w8a8(xq, uq, sx, ox, su, None, out_dtype=torch.bfloat16)
    _, (want,) = ops.silu_mul_quantize(gate, up, [(so, oo)], 8)
    assert torch.equal(fused, want)
    assert float(want.float().std()) > 1  # not a saturated tensor
    assert ops.mlp_gate_up_w8a8(xq[:, :192], gq[:, :192], uq[:, :192], sx, ox, sg, su, so, oo, 8) is None  # K < 256: not covered


def test_silu_table_equals_aten_on_every_bf16_pattern():
    """csrc/ffq_silu.h: silu of a bf16 tensor is a 65536-valued function; the large-tensor kernel reads it from an LDS table it
    fills itself (2^-24 <= |x| < 2^8), x / 2 below the window and the exact expression above it. All 65536 patterns (incl.
    denormals, +-0, +-Inf, NaNs) against ATen's silu on the device, through the kernel that takes the table path."""
    patterns = torch.arange(-32768, 32768, dtype=torch.int16).view(torch.bfloat16)
    gate = patterns.repeat(512).to(DEV)  # 33.5 M elements: above the table kernel's threshold
    s, o = torch.tensor([0.05], device=DEV), torch.tensor([-7.0], device=DEV)
    for up in (torch.ones_like(gate), torch.randn(gate.shape, device=DEV).to(torch.bfloat16)):
        product, codes = ops.silu_mul_quantize(gate, up, [(s, o)], want_product=True)
        ref = torch.nn.functional.silu(gate) * up
        nan = ref.isnan()
        assert torch.equal(product.isnan(), nan)
        assert torch.equal(product[~nan].view(torch.int16), ref[~nan].view(torch.int16))
        assert torch.equal(codes[0], ops.quantize_by_tile(product, s, product.shape, 8, torch.int8, o))
    # the small-tensor kernel (exact expression per element) on the same patterns
    once = gate[:65536].contiguous()
    small, _ = ops.silu_mul_quantize(once, torch.ones_like(once), [(s, o)], want_product=True)
    ref = torch.nn.functional.silu(once)
    assert torch.equal(small.isnan(), ref.isnan())
    assert torch.equal(small[~ref.isnan()].view(torch.int16), ref[~ref.isnan()].view(torch.int16))


def test_fused_gate_up_silu_table_window_and_fallbacks():
    """MLP-mode GEMM: per-channel gate scales spread over 2^-44 .. 2^11 push gate_proj's bf16 output below, inside and above
    the window of the silu table of the persistent kernel; the three-launch path evaluates silu exactly per element."""
    gen = torch.Generator().manual_seed(77)
    m, n, k = 2048, 1024, 256
    xq = torch.randint(-128, 128, (m, k), generator=gen, dtype=torch.int8).to(DEV)
    gq = torch.randint(-128, 128, (n, k), generator=gen, dtype=torch.int8).to(DEV)
    uq = torch.randint(-128, 128, (n, k), generator=gen, dtype=torch.int8).to(DEV)
    sx, ox = torch.tensor([1.0], device=DEV), torch.tensor([2.0], device=DEV)
    sg = (torch.exp2(torch.arange(n) % 56 - 44.0) * (1 + torch.rand(n, generator=gen))).to(DEV)
    su = (torch.rand(n, generator=gen) * 2e-5 + 1e-5).to(DEV)
    so, oo = torch.tensor([0.03], device=DEV), torch.tensor([1.0], device=DEV)
    fused = ops.mlp_gate_up_w8a8(xq, gq, uq, sx, ox, sg, su, so, oo, 8)
    gate = ops.linear_w8a8(xq, gq, sx, ox, sg, None, out_dtype=torch.bfloat16)
    up = ops.linear_w8a8(xq, uq, sx, ox, su, None, out_dtype=torch.bfloat16)
    mag = gate.float().abs()
    assert bool((mag < 2.0**-24).any()) and bool((mag >= 256).any()) and bool(((mag > 1e-3) & (mag < 10)).any())
    product, (want,) = ops.silu_mul_quantize(gate, up, [(so, oo)], 8, want_product=True)
    assert torch.equal(product, torch.nn.functional.silu(gate) * up)
    assert fused is not None and torch.equal(fused, want), f"{int((fused != want).sum())} of {want.numel()} codes differ"
    assert float(want.float().std()) > 1


# ---- weight-only linear (row *J: quantized weight x plain bf16 input, fallback.py:86-112) ---------------------------
def test_weight_only_linear_fixture_from_the_reference():
    parity_cases.check_weight_only_linear(DEV)


def _wq_case(n, k, group, bits, offset, seed):
    g = torch.Generator().manual_seed(seed)
    lo, hi = -(2 ** (bits - 1)), 2 ** (bits - 1)
    codes = torch.randint(lo, hi, (n, k), generator=g, dtype=torch.int8)
    groups = k // group
    scale = torch.rand(n * groups, generator=g) * 0.02 + 0.002
    off = torch.round(torch.randn(n * groups, generator=g) * 3) + 0.25 if offset else None  # rounds half-even inside the kernel
    return codes.to(DEV), scale.to(DEV), None if off is None else off.to(DEV)


def _wq_forms(codes, group, bits, m):
    """The ways one weight reaches the kernel: int8 container, packed nibbles (4-bit codes; the GGUF block sizes and the group
    size as packing block), each converted inside the GEMM (one pass) or by A2 as its own pass (two passes, offered from 4096
    tokens on) -> (label, kwargs of ops.linear_wq, weight tensor)."""
    forms = [("int8", dict(two_pass=False), codes)]
    if m >= 4096:
        forms.append(("int8, two passes", dict(two_pass=True), codes))
    if bits == 4:
        k = codes.shape[1]
        for block in sorted({32, 64, 128, min(group, 256)}):
            if k % block == 0:
                forms.append((f"packed block {block}", dict(pack_block=block, two_pass=False), ops.pack_int4(codes, block=block)))
        if m >= 4096:
            forms.append(("packed block 128, two passes", dict(pack_block=128, two_pass=True), ops.pack_int4(codes, block=128)))
    return forms


@pytest.mark.parametrize("k,group,bits,offset", [(512, 512, 8, False), (512, 128, 4, False), (384, 64, 4, True), (1024, 1024, 8, True), (256, 128, 8, True), (1024, 256, 4, True)])
def test_weight_only_linear_operand_is_exactly_the_dequantized_weight(k, group, bits, offset):
    """x = identity: every output is ONE product 1.0 * w^, so y[m, n] == dequantize_by_tile(codes)[n, m] bit for bit —
    the conversion of the GEMM's B operand (int8 codes or packed nibbles of every packing block) is A2, checked against the
    A2 kernel itself. Scales down to 2^-126 included: the nibble path scales by s / 16 only where that is exact."""
    n = 320
    codes, scale, off = _wq_case(n, k, group, bits, offset, seed=k + group)
    scale[::7] *= 2.0**-115  # tiny (some denormal after / 16) scales on a subset of the tiles
    scale[3] = 0.0
    x = torch.eye(k, device=DEV, dtype=torch.bfloat16)
    want = ops.dequantize_by_tile(codes, scale, (1, group), off, torch.bfloat16)
    for label, kwargs, weight in _wq_forms(codes, group, bits, k):
        y = ops.linear_wq(x, weight, scale, off, group=group, **kwargs)
        assert y is not None and y.dtype == torch.bfloat16, label
        assert torch.equal(y, want.t()), label + ": " + mismatch_report(y.cpu(), want.t().cpu())


@pytest.mark.parametrize("m,n,k,group", [(1, 256, 128, 128), (300, 130, 512, 512), (257, 515, 1024, 128), (2050, 2300, 192, 64), (4100, 1030, 1024, 1024), (77, 64, 4096, 128), (4200, 520, 640, 128)])
@pytest.mark.parametrize("offset,bias,out_dtype", [(False, False, torch.bfloat16), (True, True, torch.bfloat16), (False, True, torch.float32)])
def test_weight_only_linear_matches_float64_of_the_same_operands(m, n, k, group, offset, bias, out_dtype):
    """Ragged M / N, every K-loop length, grouped and per-channel parameters, offsets, bias, both output dtypes, every storage
    form of the weight: within one output rounding (+ fp32 accumulation) of the float64 product of x and the A2-dequantized
    weight — and all forms agree bit for bit (same operands, same tile walk, same summation order)."""
    gen = torch.Generator().manual_seed(m * 3 + n + k)
    bits = 8 if group == k else 4
    codes, scale, off = _wq_case(n, k, group, bits, offset, seed=m + n + k)
    x = torch.randn(m, k, generator=gen).to(torch.bfloat16).to(DEV)
    b = torch.randn(n, generator=gen).to(torch.bfloat16).to(DEV) if bias else None
    w_hat = ops.dequantize_by_tile(codes, scale, (1, group), off, torch.bfloat16)
    ref = x.double() @ w_hat.double().t() + (0 if b is None else b.double())
    rtol = 2.0**-8 if out_dtype == torch.bfloat16 else 1e-5
    # Up to 16 rows the int8 container and nibbles packed with block 128 take the skinny kernel (csrc/ffq_wskinny.hip) where K % 128 == 0,
    # everything else up to 512 rows the 128-column tiles (csrc/ffq_wmid.hip), beyond that the 256-row tiles: forms agree bit for bit
    # within a kernel (same operands, same walk, same summation order); across kernels only the fp32 summation order differs (include/ffq.h)
    first: dict[bool, torch.Tensor] = {}
    for label, kwargs, weight in _wq_forms(codes, group, bits, m):
        y = ops.linear_wq(x, weight, scale, off, group=group, bias=b, out_dtype=out_dtype, **kwargs)
        assert y is not None and y.dtype == out_dtype and y.shape == (m, n), label
        torch.testing.assert_close(y.double(), ref, rtol=rtol, atol=1e-5 * float(ref.abs().max()) + 1e-6 * k, msg=lambda msg: f"{label}: {msg}")
        skinny = m <= 16 and k % 128 == 0 and kwargs.get("pack_block", 0) in (0, 128)
        assert torch.equal(y, first.setdefault(skinny, y)), f"{label} differs from the first form of its kernel"


@pytest.mark.parametrize("m,n,k,group,bits,offset", [(1, 128, 128, 128, 8, False), (300, 256, 512, 512, 8, False), (257, 640, 1024, 128, 4, True),
                                                     (4100, 1152, 1024, 1024, 8, True), (4200, 384, 640, 128, 4, False), (2050, 2304, 192, 64, 4, True)])
def test_weight_only_gate_up_launch_is_the_two_linears_and_silu_mul(m, n, k, group, bits, offset):
    """ops.mlp_gate_up_wq == silu_mul_quantize(linear_wq(x, gate), linear_wq(x, up)) bit for bit (reference mlp.py:30-40 over
    fallback.py:86-112), for every storage form of the weights, ragged M, one- and two-pass: the projections are the same
    accumulators in the same order, the epilogue the same SiLU (table == ATen on all bf16 patterns) and the same roundings."""
    gc, gs, go = _wq_case(n, k, group, bits, offset, seed=m + n)
    uc, us, uo = _wq_case(n, k, group, bits, offset, seed=m + n + 1)
    gen = torch.Generator().manual_seed(m + k)
    x = (torch.randn(m, k, generator=gen) * 1.5).to(torch.bfloat16).to(DEV)
    x[0, :] *= 40.0  # gate values outside the table's window on one row: the patch path
    # both sides on the K slices the one-launch form plans for itself: the split fixes the fp32 summation order (include/ffq.h)
    # ... and on the kernel family the one-launch form belongs to, the 256-row tiles (up to 512 rows a plain launch takes the skinny form or
    # the 128-column tiles otherwise, whose summation orders are their own)
    lib = _native.library()
    split = int(lib.ffq_linear_wq_split(m, n, k, 1))
    previous = lib.ffq_force_generic_kernels(1)
    try:
        gate = ops.linear_wq(x, gc, gs, go, group=group, two_pass=False, split=split)
        up = ops.linear_wq(x, uc, us, uo, group=group, two_pass=False, split=split)
    finally:
        lib.ffq_force_generic_kernels(previous)
    want, _ = ops.silu_mul_quantize(gate, up, (), want_product=True)
    assert float(want.float().abs().max()) > 0
    for (label, kwargs, gw), (_, _, uw) in zip(_wq_forms(gc, group, bits, m), _wq_forms(uc, group, bits, m)):
        got = ops.mlp_gate_up_wq(x, gw, uw, gs, go, us, uo, group=group, **kwargs)
        assert got is not None and got.dtype == torch.bfloat16 and got.shape == (m, n), label
        assert torch.equal(got, want), label + ": " + mismatch_report(got.cpu(), want.cpu())
    # default policy (library decides about the two passes) agrees as well
    assert torch.equal(ops.mlp_gate_up_wq(x, gc, uc, gs, go, us, uo, group=group), want)
    # not covered: N % 128 != 0, fp32 activations
    assert ops.mlp_gate_up_wq(x, gc[:64], uc[:64], gs[: 64 * (k // group)], None if go is None else go[: 64 * (k // group)], us[: 64 * (k // group)],
                              None if uo is None else uo[: 64 * (k // group)], group=group) is None
    assert ops.mlp_gate_up_wq(x.float(), gc, uc, gs, go, us, uo, group=group) is None


@pytest.mark.parametrize("m,n,k,group,bits", [(1, 256, 2048, 2048, 8), (40, 520, 1024, 128, 4), (300, 256, 1536, 1536, 8), (515, 770, 2048, 128, 4), (2048, 1024, 4096, 4096, 8),
                                              (4352, 4096, 1024, 1024, 8), (4200, 4224, 768, 128, 4)])
def test_weight_only_linear_split_k_slices_sum_in_a_fixed_order(m, n, k, group, bits, generic_kernels):
    """(Up to 512 rows the product takes the skinny kernel or the 128-column tiles, whose own split-K is covered by tests/test_skinny_gpu.py
    and tests/test_mid_gpu.py: those cases run the 256-row-tile kernel here through the library's test hook, so its exchange stays covered
    at every row count.)
    Split-K (the tiles of the last, partly filled round of the persistent walk — all tiles when there are fewer than CUs — have
    their K range cut into slices; the units of a tile exchange fp32 partial sums and each finishes a fixed share in a fixed
    order): every forced split — incl. uneven slices, ragged M / N, whole rounds ahead of the split tail (272 tiles: 16 of them
    split) — stays within one output rounding of the float64 product, repeats bit for bit launch after launch (no dependence on
    arrival order), leaves the ticket buffer zero, and where the sums are order-independent (small integers, power-of-two
    scales) ALL splits give the same bits."""
    gen = torch.Generator().manual_seed(m + n + k)
    codes, scale, off = _wq_case(n, k, group, bits, True, seed=m + k)
    x = torch.randn(m, k, generator=gen).to(torch.bfloat16).to(DEV)
    w_hat = ops.dequantize_by_tile(codes, scale, (1, group), off, torch.bfloat16)
    ref = x.double() @ w_hat.double().t()
    lib = _native.library()
    generic_kernels(m <= 512)
    plan = int(lib.ffq_linear_wq_split(m, n, k, 0))
    # all units of a tile wait for each other: a split is admitted while tiles * split <= CUs (and slices keep >= 2 super-steps)
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    tail = (-(-m // 256) * -(-n // 256)) % cus  # tiles of the last, partly filled round of the persistent walk
    assert tail > 0 and int(lib.ffq_linear_wq_tickets(m, n, k, 0)) >= 2 * tail  # (the ticket figure is an upper bound over the forms)
    most = min(cus // tail, (k // 64) // 2, 32)
    assert 1 <= plan <= most
    packed = ops.pack_int4(codes, block=128) if bits == 4 else None
    for split in sorted({1, 2, 3, 5, 8, 16, plan}):
        if split > most:
            continue
        y = ops.linear_wq(x, codes, scale, off, group=group, split=split)
        torch.testing.assert_close(y.double(), ref, rtol=2.0**-8, atol=1e-5 * float(ref.abs().max()) + 1e-6 * k, msg=lambda msg: f"split {split}: {msg}")
        for _ in range(3):
            assert torch.equal(ops.linear_wq(x, codes, scale, off, group=group, split=split), y), f"split {split}: two launches differ"
        if packed is not None:
            assert torch.equal(ops.linear_wq(x, packed, scale, off, group=group, pack_block=128, split=split), y), f"split {split}: packed nibbles differ"
        y32 = ops.linear_wq(x, codes, scale, off, group=group, split=split, out_dtype=torch.float32)
        assert torch.equal(y32.to(torch.bfloat16), y), f"split {split}: fp32 output rounds to something else"
        if split > 1:
            # units that give up their wait (a peer that cannot become resident; here: every odd slice, at once, through the library's
            # test hook) publish all their pieces and leave; the last arriver finishes them in the same order: the same bits
            previous = lib.ffq_force_generic_kernels(2 | int(m <= 512))
            try:
                for _ in range(3):
                    assert torch.equal(ops.linear_wq(x, codes, scale, off, group=group, split=split), y), f"split {split}: abandoned units change the result"
            finally:
                lib.ffq_force_generic_kernels(previous)
    assert all(int(t.abs().sum()) == 0 for t in ops._TICKETS.values()), "a launch left tickets behind"
    # order-independent sums: integer activations in [-4, 4], scales 2^-3, integer offsets -> every partial sum is exact in fp32
    xi = torch.randint(-4, 5, (m, k), generator=gen).to(torch.bfloat16).to(DEV)
    s2 = torch.full_like(scale, 0.125)
    o2 = None if off is None else torch.round(off)
    exact = (xi.double() @ ops.dequantize_by_tile(codes, s2, (1, group), o2, torch.float32).double().t()).to(torch.float32)
    for split in (1, 2, 3, 5, 8, 16):
        if split <= most:
            assert torch.equal(ops.linear_wq(xi, codes, s2, o2, group=group, split=split, out_dtype=torch.float32), exact), f"split {split}"
    # a forced split without scratch, or one whose units could not all be resident, is refused, never silently changed
    with pytest.raises(Exception):
        ops.linear_wq(x, codes, scale, off, group=group, split=most + 1)
    out = torch.empty(m, n, dtype=torch.bfloat16, device=DEV)
    if most >= 2:
        rc = lib.ffq_linear_wq(x.data_ptr(), int(_cabi.DType.BF16), codes.data_ptr(), int(_cabi.DType.I8), 0, scale.data_ptr(), None, scale.numel(), group, None, 0,
                               out.data_ptr(), int(_cabi.DType.BF16), m, n, k, None, 0, None, 2, None)
        assert rc != 0


@pytest.mark.parametrize("m,rows,k,group,bits,offset", [(300, (512, 256, 256), 512, 512, 8, False), (2048, (1024, 256, 256), 1024, 128, 4, True), (77, (256, 130), 256, 64, 4, True),
                                                        (4200, (768, 256, 320), 640, 640, 8, True)])
def test_weight_only_linears_on_one_input_as_one_launch(m, rows, k, group, bits, offset, oracle_lib):
    """ops.linear_wq_multi (q_proj / k_proj / v_proj: three QuantizedLinear modules on one hidden state, reference nn/linear.py:32-39
    three times) == the separate ops.linear_wq calls BIT FOR BIT under the same split (one tile walk over the concatenated column
    tiles, the same K order inside every tile), every storage form, ragged M and a ragged last matrix; the default plan within one
    output rounding of float64; the oracle's composition agrees on a small case."""
    gen = torch.Generator().manual_seed(m + k)
    x = torch.randn(m, k, generator=gen).to(torch.bfloat16).to(DEV)
    cases = [_wq_case(n, k, group, bits, offset, seed=m + n + i) for i, n in enumerate(rows)]
    codes, scales, offs = [c[0] for c in cases], [c[1] for c in cases], [c[2] for c in cases]
    want = [ops.linear_wq(x, c, s_, o, group=group, two_pass=False, split=1) for c, s_, o in cases]
    forms = [("int8", dict(two_pass=False), codes)]
    if m >= 4096:
        forms.append(("int8, two passes", dict(two_pass=True), codes))
    if bits == 4:
        forms.append(("packed", dict(pack_block=64, two_pass=False), [ops.pack_int4(c, block=64) for c in codes]))
    for label, kwargs, weights in forms:
        got = ops.linear_wq_multi(x, weights, scales, offs, group=group, split=1, **kwargs)
        assert got is not None and len(got) == len(rows), label
        # the separate launches of the SAME storage form (up to 16 rows a packing block other than 128 takes the 128-column tiles,
        # the int8 container the skinny kernel: another summation order, include/ffq.h)
        same_kernel = want if m > 16 or "pack_block" not in kwargs else [ops.linear_wq(x, w_, s_, o, group=group, split=1, **kwargs) for w_, (_, s_, o) in zip(weights, cases)]
        for g, w_, n in zip(got, same_kernel, rows):
            assert g.shape == (m, n) and torch.equal(g, w_), f"{label}: " + mismatch_report(g.cpu(), w_.cpu())
        for g, w_ in zip(got, want):
            torch.testing.assert_close(g.double(), w_.double(), rtol=2.0**-7, atol=1e-5 * float(w_.double().abs().max()) + 1e-6 * k)
    planned = ops.linear_wq_multi(x, codes, scales, offs, group=group)
    for g, (c, s_, o) in zip(planned, cases):
        ref = x.double() @ ops.dequantize_by_tile(c, s_, (1, group), o, torch.bfloat16).double().t()
        torch.testing.assert_close(g.double(), ref, rtol=2.0**-8, atol=1e-5 * float(ref.abs().max()) + 1e-6 * k)
    assert torch.equal(ops.linear_wq_multi(x, codes, scales, offs, group=group, out_dtype=torch.float32)[1].to(torch.bfloat16), planned[1])
    # not this kernel's: a middle matrix that is no multiple of 256 rows, mixed offsets
    assert ops.linear_wq_multi(x, [codes[-1], codes[0]], [scales[-1], scales[0]], [offs[-1], offs[0]], group=group) is None or rows[-1] % 256 == 0
    assert ops.linear_wq_multi(x, codes[:2], scales[:2], [offs[0], None if offs[1] is not None else scales[1]], group=group) is None
    if m <= 300:
        host = ops.linear_wq_multi  # the oracle through the same Python surface
        with use_backend(oracle_lib):
            cpu = host(x.cpu(), [c.cpu() for c in codes], [s_.cpu() for s_ in scales], [None if o is None else o.cpu() for o in offs], group=group)
        for g, c_ in zip(planned, cpu):
            torch.testing.assert_close(g.cpu().float(), c_.float(), rtol=2.0**-6, atol=2.0**-6 * float(c_.float().abs().max()))


def test_weight_only_gate_up_against_the_oracle_composition(oracle_lib):
    """The C restatement (two ffq_linear_wq in double + its silu_mul) on a small case: products within one bf16 rounding of the
    exact projections' product wherever the projections themselves agree after rounding."""
    from fastforward_amd import _cabi
    import ctypes

    m, n, k, group = 48, 128, 256, 128
    gc, gs, go = _wq_case(n, k, group, 4, True, seed=5)
    uc, us, uo = _wq_case(n, k, group, 4, True, seed=6)
    x = torch.randn(m, k, generator=torch.Generator().manual_seed(9)).to(torch.bfloat16)
    got = ops.mlp_gate_up_wq(x.to(DEV), gc, uc, gs, go, us, uo, group=group).cpu()
    lib = oracle_lib
    out = torch.empty(m, n, dtype=torch.bfloat16)
    host = [t.cpu().contiguous() for t in (gc, uc, gs, go, us, uo)]
    rc = lib.ffq_mlp_gate_up_wq(x.data_ptr(), int(_cabi.DType.BF16), host[0].data_ptr(), host[1].data_ptr(), int(_cabi.DType.I8), 0, host[2].data_ptr(), host[3].data_ptr(),
                                host[4].data_ptr(), host[5].data_ptr(), host[2].numel(), group, out.data_ptr(), m, n, k, None, 0, None, 0, None)
    assert rc == 0
    # the GEMM accumulates in fp32 in its own order: a projection may land on the neighbouring bf16 -> compare loosely, and
    # exactly where the HIP projections equal the oracle's
    torch.testing.assert_close(got.float(), out.float(), rtol=2.0**-6, atol=2.0**-6 * float(out.float().abs().max()))
    assert (got == out).float().mean() > 0.9


def test_weight_only_linear_refuses_what_it_does_not_cover():
    codes, scale, off = _wq_case(64, 192, 96, 4, False, seed=1)  # groups of 96: not a multiple of 64
    x = torch.randn(8, 192, device=DEV, dtype=torch.bfloat16)
    assert ops.linear_wq(x, codes, scale, off, group=96) is None
    assert ops.linear_wq(x.float(), codes, scale[:64], None) is None  # fp32 activations: the float fallback's job
    assert ops.linear_wq(x, ops.pack_int4(codes, block=24), scale, off, group=192, pack_block=24) is None  # packing blocks: powers of two >= 32
    # ... and the dispatcher then runs the reference's path (dequantize + F.linear) with the same result as ever
    w = torch.randn(64, 192, device=DEV, dtype=torch.bfloat16)
    q = ff.nn.LinearQuantizer(4, granularity=ff.PerBlock(1, 96, 0), quantized_dtype=torch.int8, device=DEV)
    with ff.estimate_ranges(q, ff.range_setting.running_minmax), torch.no_grad():
        wq = q(w)
    assert ff.dispatcher.dispatch("linear", input=x, weight=wq) is None
    with ff.strict_quantization(False), torch.no_grad():
        assert torch.equal(ff.nn.functional.linear(x, wq), torch.nn.functional.linear(x, wq.dequantize()))


# ---- A8 and the grid estimator on strided channels / N-d tiles: HIP by-tile kernels, no device-ATen composite ---------
_ODD_TILINGS = [((96, 160), (96, 1)), ((64, 48, 40), (16, 8, 4)), ((33, 17), (1, 1)), ((6, 64, 20), (6, 1, 20)), ((128, 96), (32, 32)), ((1000, 24), (1000, 1)), ((7, 9), (7, 9))]


@pytest.mark.parametrize("shape,tile", _ODD_TILINGS, ids=str)
@pytest.mark.parametrize("dtype,with_offset", [(torch.float32, True), (torch.bfloat16, True), (torch.float32, False)])
def test_backward_by_tile_kernel_covers_every_tiling(shape, tile, dtype, with_offset, monkeypatch):
    """PerChannel(-1), PerChannel(1) on 3-D, N-d tiles, element-wise tiles, 2-D blocks: the HIP library itself produces the
    gradients (the composite of device tensor ops is never called) and they equal the op-for-op formulas."""
    torch.manual_seed(sum(shape))
    x = (torch.randn(shape, device=DEV) * 2).to(dtype)
    g = torch.randn(shape, device=DEV).to(dtype)
    ntiles = 1
    for s_, t_ in zip(shape, tile):
        ntiles *= s_ // t_
    scale = torch.rand(ntiles, device=DEV) * 0.05 + 0.02
    offset = (torch.rand(ntiles, device=DEV) * 6 - 3) if with_offset else None
    want = ops._quantize_by_tile_backward_composite(x, g, scale, tile, 4.0, offset)

    def never(*a, **k):
        raise AssertionError("the composite was called: the HIP kernel did not cover this tiling")

    monkeypatch.setattr(ops.static, "_quantize_by_tile_backward_composite", never)  # (the caller looks the composite up in its own module)
    got = ops.quantize_by_tile_backward(x, g, scale, tile, 4.0, offset)
    assert torch.equal(got[0], want[0])
    from fastforward_amd.quantization.tiled_tensor import tiles_to_rows
    mag = tiles_to_rows(g.float().abs(), torch.Size(tile)).sum(1)
    assert bool(((got[1] - want[1]).abs() <= 4e-6 * 12 * mag + 1e-6).all())
    if with_offset:
        assert bool(((got[2] - want[2]).abs() <= 4e-6 * 0.07 * mag + 1e-6).all())
    else:
        assert got[2].numel() == 0


@pytest.mark.parametrize("shape,tile", _ODD_TILINGS[:5], ids=str)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_grid_error_by_tile_kernel_covers_every_tiling(shape, tile, dtype):
    torch.manual_seed(sum(shape) + 1)
    x = torch.randn(shape, device=DEV).to(dtype)
    ntiles = 1
    for s_, t_ in zip(shape, tile):
        ntiles *= s_ // t_
    ncand = 19
    scales = torch.rand(ncand, ntiles, device=DEV) * 0.3 + 0.05
    offsets = torch.round(torch.randn(ncand, ntiles, device=DEV) * 2)
    got = ops.grid_sqerror_by_tile(x, scales, offsets, tile, 4.0)
    assert got is not None, "the HIP kernel did not cover this tiling"
    from fastforward_amd.quantization.tiled_tensor import tiles_to_rows
    want = torch.empty_like(got)
    for c in range(ncand):  # the reference's loop: quantize, dequantize, squared error per tile (min_error.py:218-231)
        q = ops.quantize_by_tile(x, scales[c], tile, 4, dtype, offsets[c])
        d = ops.dequantize_by_tile(q, scales[c], tile, offsets[c], dtype)
        want[c] = tiles_to_rows(((d - x) ** 2), torch.Size(tile)).float().sum(1)
    rtol = 2e-2 if dtype == torch.bfloat16 else 1e-5
    torch.testing.assert_close(got, want, rtol=rtol, atol=1e-6)
    again = ops.grid_sqerror_by_tile(x, scales, offsets, tile, 4.0, out=got.clone())
    torch.testing.assert_close(again, 2 * got, rtol=1e-6, atol=0)


def test_equal_quantizers_share_one_launch_on_the_same_activation():
    """q_proj / k_proj / v_proj quantize the same hidden state with their own input quantizers (reference nn/linear.py:33);
    once their parameters have stopped changing and are known to be equal, the later ones reuse the first one's codes
    (quantization/affine/_memo.py) — same bits, fewer launches; any write to the tensor or to a parameter ends the reuse.
    Reuse exists only inside ``sibling_quantizers()`` (this package's module forwards open it around the sibling calls)."""
    from fastforward_amd.quantization.affine._memo import RECENT, sibling_quantizers

    torch.manual_seed(5)
    x = torch.randn(4, 64, 256, device=DEV, dtype=torch.bfloat16)
    qs = [ff.nn.LinearQuantizer(8, symmetric=False, quantized_dtype=torch.int8, device=DEV) for _ in range(3)]
    with torch.no_grad():
        with ff.estimate_ranges(torch.nn.ModuleList(qs), ff.range_setting.running_minmax), sibling_quantizers():
            for q in qs:
                q(x)  # calibration: parameters rewritten on every call, never "stable" -> no host reads, no reuse
        want = ops.quantize_by_tile(x, qs[0].scale, x.shape, 8, torch.int8, qs[0].offset)
        hits = RECENT.hits
        rounds = []
        for _ in range(3):  # the final versions were sighted by the last calibration call: verdicts may be read in the first round
            with sibling_quantizers():
                rounds.append([q(x).raw_data for q in qs])
        third = rounds[2]
        assert RECENT.hits >= hits + 4 and third[1].data_ptr() == third[0].data_ptr() == third[2].data_ptr()
        for codes in rounds[0] + rounds[1] + rounds[2]:
            assert torch.equal(codes, want)
        assert RECENT._data is None and not RECENT._entries, "the slot outlived its block"
        # outside a block every call launches A1 (the reference's behaviour)
        before = RECENT.hits
        loose = [q(x).raw_data for q in qs]
        assert RECENT.hits == before and len({c.data_ptr() for c in loose}) == 3 and all(torch.equal(c, want) for c in loose)
        # a new tensor, a write autograd can see, a raw-pointer write of this package: all miss
        y = x.clone()
        with sibling_quantizers():
            a = qs[0](y).raw_data
            y.mul_(0.5)
            b = qs[1](y).raw_data
        assert RECENT.hits == before and torch.equal(b, ops.quantize_by_tile(y, qs[1].scale, y.shape, 8, torch.int8, qs[1].offset)) and not torch.equal(a, b)
        # a parameter that moves: the other quantizers no longer share with it, and its own codes follow the new range
        lo, hi = qs[2].quantization_range
        qs[2].quantization_range = (lo * 0.5, hi * 0.5)
        with sibling_quantizers():
            c = [q(x).raw_data for q in qs]
        assert torch.equal(c[0], want) and torch.equal(c[1], want) and not torch.equal(c[2], want)
        assert torch.equal(c[2], ops.quantize_by_tile(x, qs[2].scale, x.shape, 8, torch.int8, qs[2].offset))


def test_sibling_estimators_share_one_reduction_and_merge_exactly():
    """RunningMinMax on q / k / v inputs: inside ``sibling_quantizers()`` the tensor is reduced ONCE and every estimator merges the
    two numbers into ITS OWN running state (reference minmax.py:227-239 per quantizer). Exact whatever the states are — here the
    three quantizers start from different earlier batches — and for Inf / NaN batches (the status word included)."""
    from fastforward_amd.quantization.affine._memo import RECENT, sibling_quantizers

    torch.manual_seed(11)
    earlier = [torch.randn(4, 64, 256, device=DEV, dtype=torch.bfloat16) * s for s in (0.5, 2.0, 5.0)]
    batches = [torch.randn(4, 64, 256, device=DEV, dtype=torch.bfloat16) * s for s in (1.0, 3.0, 0.1)]
    spiked = batches[1].clone()
    spiked[0, 0, 0] = float("inf")

    def run(shared: bool, feed):
        qs = [ff.nn.LinearQuantizer(8, symmetric=sym, quantized_dtype=torch.int8, device=DEV) for sym in (False, False, True)]
        with torch.no_grad(), ff.estimate_ranges(torch.nn.ModuleList(qs), ff.range_setting.running_minmax, sync_free=True) as handles:
            for q, x in zip(qs, earlier):  # three different running states
                q(x)
            for x in feed:
                if shared:
                    with sibling_quantizers():
                        codes = [q(x).raw_data for q in qs]
                else:
                    codes = [q(x).raw_data for q in qs]
            estimators = [next(iter(q.overrides)) for q in qs]
            state = [(e.min.clone(), e.max.clone(), e.status.clone()) for e in estimators]
            del handles
            return [(q.scale.clone(), q.offset.clone() if q.offset is not None else None) for q in qs], codes, state

    hits = RECENT.extrema_hits
    got = run(True, batches)
    assert RECENT.extrema_hits == hits + 2 * len(batches)  # the second and third sibling of every batch
    want = run(False, batches)
    for (gs, go), (ws, wo) in zip(got[0], want[0]):
        assert torch.equal(gs, ws) and (go is None) == (wo is None) and (go is None or torch.equal(go, wo))
    assert all(torch.equal(a, b) for a, b in zip(got[1], want[1]))
    assert all(torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]) for a, b in zip(got[2], want[2]))
    # an infinite batch: the same running extrema and the same status flags on every sibling, then the reference's error
    for shared in (True, False):
        with pytest.raises(NotImplementedError, match="[Ii]nfinite"):
            run(shared, [spiked])


def test_a_write_through_dot_data_between_two_quantizer_calls_gets_fresh_codes():
    """``h.data.mul_(2)`` moves neither ``h._version`` nor ``h.data_ptr()``: a cache keyed on them would hand the second of two
    equal quantizers the first one's codes where the reference re-quantizes (nn/linear.py:32-39). The memo only lives inside the
    package's own sibling blocks, so user code between two module calls always sees A1 of the CURRENT bytes — on bare
    quantizers and on the module graph (two forwards of a quantized Llama around a ``.data`` write of the embedding output)."""
    from fastforward_amd.quantization.affine._memo import RECENT

    torch.manual_seed(6)
    h = torch.randn(2, 64, 256, device=DEV, dtype=torch.bfloat16)
    qs = [ff.nn.LinearQuantizer(8, symmetric=False, quantized_dtype=torch.int8, device=DEV) for _ in range(2)]
    with torch.no_grad():
        for q in qs:
            q.quantization_range = (torch.tensor([-4.0], device=DEV), torch.tensor([4.0], device=DEV))
        for _ in range(3):  # parameters sighted, verdicts readable: everything a version-keyed cache would need to hit
            first, second = qs[0](h).raw_data, qs[1](h).raw_data
        assert torch.equal(first, second)
        version, pointer = h._version, h.data_ptr()
        first = qs[0](h).raw_data.clone()
        h.data.mul_(2)
        assert h._version == version and h.data_ptr() == pointer  # invisible to the counters
        second = qs[1](h).raw_data
        assert torch.equal(second, ops.quantize_by_tile(h, qs[1].scale, h.shape, 8, torch.int8, qs[1].offset)) and not torch.equal(second, first)
        # module graph: the same model, the same ids, the embedding table scaled through .data between the two forwards
        cfg = llama.LlamaConfig.tiny()
        model = llama.quantize_llama(llama.build_model(cfg, DEV, seed=3))
        ids = torch.randint(0, cfg.vocab_size, (2, 64), device=DEV)
        llama.calibrate(model, [ids])
        with ff.strict_quantization(False):
            for _ in range(2):
                before = model(ids)
            model.embed_tokens.weight.data.mul_(3.0)
            after = model(ids)
            with llama.eager_modules():
                want = model(ids)
        assert not torch.equal(before, after)
        torch.testing.assert_close(after.float(), want.float(), rtol=0.05, atol=0.05 * float(want.float().abs().max()))
        assert RECENT._data is None


def test_batched_weight_quantization_equals_member_by_member():
    """ffq_quantize_rows_batch: the seven weights of a Llama-3-8B decoder layer (and a ragged mix with real offsets) in one
    launch; every member's codes equal its own A1 launch bit for bit. Shapes the kernel declines return None."""
    torch.manual_seed(12)
    shapes = [(4096, 4096), (1024, 4096), (1024, 4096), (4096, 4096), (14336, 4096), (14336, 4096), (4096, 14336)]
    ws = [(torch.randn(s, device=DEV) * 0.02).to(torch.bfloat16) for s in shapes]
    scales = [w.float().abs().amax(1) / 127 for w in ws]
    offsets = [None] * 7
    got = ops.quantize_rows_batch(ws, scales, offsets, 8)
    assert got is not None
    for w, s, c in zip(ws, scales, got):
        assert torch.equal(c, ops.quantize_by_tile(w, s, (1, w.shape[1]), 8, torch.int8))
    # the same launch with the row sums of the codes (the int8 GEMM's zero-point term): += into zeroed int32 slices of one pool
    pool = torch.zeros(sum(s[0] for s in shapes), dtype=torch.int32, device=DEV)
    sums, at = [], 0
    for s in shapes:
        sums.append(pool[at:at + s[0]])
        at += s[0]
    again = ops.quantize_rows_batch(ws, scales, offsets, 8, rowsums=sums)
    assert again is not None
    for c, c2, r in zip(got, again, sums):
        assert torch.equal(c, c2) and torch.equal(r, c.to(torch.int32).sum(1, dtype=torch.int32))
    x8 = torch.randint(-128, 128, (512, 4096), device=DEV, dtype=torch.int8)
    sx, ox = torch.tensor([0.02], device=DEV), torch.tensor([3.0], device=DEV)
    assert torch.equal(ops.linear_w8a8(x8, got[0], sx, ox, scales[0], None, out_dtype=torch.bfloat16, w_rowsum=sums[0]),
                       ops.linear_w8a8(x8, got[0], sx, ox, scales[0], None, out_dtype=torch.bfloat16))
    assert ops.quantize_rows_batch([ws[0][:, :2560].contiguous()], [scales[0]], [None], 8, rowsums=[sums[0]]) is None  # 2560 columns: a wave would straddle rows
    small = [(torch.randn(256, 1024, device=DEV)).to(torch.bfloat16), (torch.randn(512, 48, device=DEV) * 3).to(torch.bfloat16), (torch.randn(16, 4096, device=DEV)).to(torch.bfloat16)]
    sc = [torch.rand(w.shape[0], device=DEV) * 0.05 + 0.01 for w in small]
    of = [torch.round(torch.randn(256, device=DEV) * 5) + 0.5, None, torch.zeros(16, device=DEV)]
    got = ops.quantize_rows_batch(small, sc, of, 4)
    assert got is not None
    for w, s, o, c in zip(small, sc, of, got):
        assert torch.equal(c, ops.quantize_by_tile(w, s, (1, w.shape[1]), 4, torch.int8, o))
    assert ops.quantize_rows_batch([small[0][:, :40].contiguous()], [sc[0]], [None]) is None  # 40 columns: not a multiple of 16
    assert ops.quantize_rows_batch([small[0].float()], [sc[0]], [None]) is None                # fp32 weights: member by member


def test_fractional_bit_widths_do_not_take_the_clamp_first_row_kernels():
    """ffq_quantize_rows_rowsum / ffq_quantize_rows_batch clamp before they round, which equals the reference's round, clamp,
    truncating cast (_quantizer_impl.py:154-169) only for integer clamp bounds. With 3.5 bits both entries decline (ops return None, the C ABI
    answers FFQ_ERR_DTYPE) and the general A1 kernel — whose codes are the oracle's (tests/test_oracle_golden.py) — is what a caller gets."""
    torch.manual_seed(41)
    w = (torch.randn(256, 1024, device=DEV) * 3).to(torch.bfloat16)
    scale = torch.rand(256, device=DEV) * 0.5 + 0.25
    assert ops.quantize_rows_rowsum(w, scale, None, 3.5) is None
    assert ops.quantize_rows_batch([w, w], [scale, scale], [None, None], 3.5) is None
    lib = ops._native.library()
    codes = torch.empty(w.shape, dtype=torch.int8, device=DEV)
    sums = torch.zeros(256, dtype=torch.int32, device=DEV)
    rc = lib.ffq_quantize_rows_rowsum(ops._ptr(w), ops._tag(w.dtype), ops._ptr(scale), None, 256, 1024, 3.5, ops._ptr(codes), ops._ptr(sums), None)
    assert rc == 6  # FFQ_ERR_DTYPE
    got = ops.quantize_by_tile(w, scale, (1, 1024), 3.5, torch.int8)
    with use_backend(load_oracle()):
        want = ops.quantize_by_tile(w.cpu(), scale.cpu(), (1, 1024), 3.5, torch.int8)
    assert torch.equal(got.cpu(), want)
    assert int(got.min()) == -5 and int(got.max()) == 4  # trunc(-2^2.5) = -5, trunc(2^2.5 - 1) = 4: not the -6 a clamp-first kernel gives


@pytest.mark.parametrize("causal", [True, False])
def test_attention_rotates_q_on_the_way_in(causal):
    """ops.attention(q_rope=(cos, sin)) on an UN-rotated q == ops.rope_ on q followed by ops.attention: context and o_proj codes bit
    for bit (the rotation is the rotary kernel's arithmetic applied to the fragments a lane already holds); k alone through
    rope_(None, k, ...) == the k half of the joint call. GQA shape of the 8B model at a short sequence; the oracle restates the
    same composition on the CPU."""
    g = torch.Generator().manual_seed(9)
    b, s, h, hkv, d = 2, 256, 8, 2, 128
    q = (torch.randn(b, s, h * d, generator=g) * 1.5).to(torch.bfloat16)
    k = (torch.randn(b, s, hkv * d, generator=g) * 1.5).to(torch.bfloat16)
    v = torch.randn(b, s, hkv * d, generator=g).to(torch.bfloat16)
    cos, sin = (t.to(torch.bfloat16) for t in (torch.cos(torch.randn(s, d, generator=g) * 3), torch.sin(torch.randn(s, d, generator=g) * 3)))
    sc, of = torch.tensor([0.021]), torch.tensor([1.7])

    def run(device):
        qd, kd, vd, cd, sd = (t.to(device) for t in (q, k, v, cos, sin))
        q1, k1 = qd.clone(), kd.clone()
        ops.rope_(q1, k1, cd, sd, d)
        want_ctx, want_codes = ops.attention(q1, k1, vd, d, causal=causal, quantizer=(sc.to(device), of.to(device)))
        k2 = kd.clone()
        ops.rope_(None, k2, cd, sd, d)
        assert torch.equal(k2, k1)
        got_ctx, got_codes = ops.attention(qd, k2, vd, d, causal=causal, quantizer=(sc.to(device), of.to(device)), q_rope=(cd, sd))
        assert torch.equal(got_ctx, want_ctx) and torch.equal(got_codes, want_codes)
        return [got_ctx.float().cpu(), got_codes.cpu()]

    _both(run)  # the equalities hold on the HIP path and in the oracle's restatement (HIP against oracle: test_attention_and_fused_output_quantizer)
