"""Full-size parity of the W8A8 GEMM and of A1 / A2 / A4 at EVERY BASELINE shape (Llama-3-8B and Llama-3-70B).

The contraction of integer codes is exact, so there is one right answer. It is computed here INDEPENDENTLY of the HIP
GEMM family: row-chunked float64 matmuls of the codes on the device (rocBLAS dgemm; exact while |sum| < 2^53), never by
another launch of the kernel under test. With unit scales the kernel's fp32 output IS the int32 accumulator (|acc| < 2^24
for random int8 codes at K <= 28672), so the comparison is `torch.equal`; with real scales / offsets the epilogue's fp32
operations are restated with elementwise torch ops (one IEEE operation each, same order as csrc/ffq_linear.hip).

Reference path: src/fastforward/_gen/fallback.py:77-112 (dequantize, dequantize, F.linear) — the kernel contracts the
codes instead; tolerance against the reference's bf16 chain is covered by G6 / check_linear.
"""

import pytest
import torch

import fastforward_amd as ff

from fastforward_amd import ops

pytestmark = pytest.mark.gpu
DEV = "cuda"
T = 8 * 2048  # tokens of one BASELINE step

# (N, K) of every quantized linear: Llama-3-8B then Llama-3-70B (q/o, k/v, gate/up, down)
LLAMA8B = [(4096, 4096), (1024, 4096), (14336, 4096), (4096, 14336)]
LLAMA70B = [(8192, 8192), (1024, 8192), (28672, 8192), (8192, 28672)]


@pytest.fixture(autouse=True)
def _backend(hip_backend):
    yield


def exact_accumulators(xq: torch.Tensor, wq64: torch.Tensor, rows: slice) -> torch.Tensor:
    """sum_k xq[m, k] * wq[n, k] for the rows `rows` in float64 (exact: every partial sum is an integer below 2^53),
    as int64. `wq64` is the weight codes already converted to float64."""
    return (xq[rows].double() @ wq64.T).round().to(torch.int64)


def _codes(n: int, k: int, seed: int):
    g = torch.Generator(device=DEV).manual_seed(seed)
    xq = torch.randint(-128, 128, (T, k), device=DEV, dtype=torch.int8, generator=g)
    wq = torch.randint(-128, 128, (n, k), device=DEV, dtype=torch.int8, generator=g)
    return xq, wq, g


@pytest.mark.parametrize("n,k", LLAMA8B + LLAMA70B, ids=lambda v: str(v))
def test_plain_gemm_equals_integer_ground_truth_at_full_size(n, k):
    """T = 16384 tokens: (a) unit scales, no offsets: fp32 output == the exact accumulator; (b) the forward's real
    configuration (per-tensor asymmetric activation, per-channel symmetric weight, bf16 output): == the fp32 epilogue
    restated with torch ops on the exact accumulator and the exact row sums."""
    xq, wq, g = _codes(n, k, 7 * n + k)
    one = torch.ones(1, device=DEV)
    ones_n = torch.ones(n, device=DEV)
    sx = torch.tensor([0.0173], device=DEV)
    ox = torch.tensor([11.3], device=DEV)  # rounds to 11 inside the kernel, as A2 does
    sw = torch.rand(n, device=DEV, generator=g) * 1e-3 + 2e-4
    got_unit = ops.linear_w8a8(xq, wq, one, None, ones_n, None, out_dtype=torch.float32)
    got_real = ops.linear_w8a8(xq, wq, sx, ox, sw, None, out_dtype=torch.bfloat16)
    rsw = wq.sum(dim=1, dtype=torch.int64)
    assert int(rsw.abs().max()) < 2**24
    chunk = 2048 if n * k <= 14336 * 4096 else 1024
    wq64 = wq.double()
    for r0 in range(0, T, chunk):
        rows = slice(r0, r0 + chunk)
        acc = exact_accumulators(xq, wq64, rows)
        assert int(acc.abs().max()) < 2**24  # representable in fp32: the unit-scale output must be the integer itself
        assert torch.equal(got_unit[rows], acc.float()), f"rows {r0}..: {int((got_unit[rows] != acc.float()).sum())} accumulators differ"
        # epilogue of csrc/ffq_linear.hip: v = float(acc) + ox * rowsum_w[n];  y = (sx * sw[n]) * v;  bf16(y)
        v = acc.float() + torch.round(ox) * rsw.float()[None, :]
        y = ((sx * sw)[None, :] * v).to(torch.bfloat16)
        assert torch.equal(got_real[rows], y), f"rows {r0}..: {int((got_real[rows] != y).sum())} outputs differ"
        del acc, v, y


@pytest.mark.parametrize("m,n,k", [(4099, 2048, 512), (2050, 2304, 256), (4100, 1032, 1024), (16383, 1096, 384), (2304, 4104, 640)], ids=lambda v: str(v))
def test_persistent_gemm_ragged_edges_bf16(m, n, k):
    """The persistent kernel's half-precision epilogue on shapes whose last row tile / last column tile / last wave are cut
    (whole-line non-temporal stores through the LDS slab where a wave's 64 columns are inside, scalar stores where not, row
    predicates): every output equals the fp32 epilogue restated on the exact accumulator."""
    g = torch.Generator(device=DEV).manual_seed(m + 3 * n + k)
    xq = torch.randint(-128, 128, (m, k), device=DEV, dtype=torch.int8, generator=g)
    wq = torch.randint(-128, 128, (n, k), device=DEV, dtype=torch.int8, generator=g)
    sx, ox = torch.tensor([0.021], device=DEV), torch.tensor([-6.6], device=DEV)
    sw = torch.rand(n, device=DEV, generator=g) * 1e-3 + 2e-4
    got = ops.linear_w8a8(xq, wq, sx, ox, sw, None, out_dtype=torch.bfloat16)
    acc = exact_accumulators(xq, wq.double(), slice(0, m))
    rsw = wq.sum(dim=1, dtype=torch.int64)
    want = ((sx * sw)[None, :] * (acc.float() + torch.round(ox) * rsw.float()[None, :])).to(torch.bfloat16)
    assert got.shape == (m, n) and torch.equal(got, want), f"{int((got != want).sum())} of {want.numel()} outputs differ"
    # same launch geometry, fp32 and fp16 outputs
    assert torch.equal(ops.linear_w8a8(xq, wq, sx, ox, sw, None, out_dtype=torch.float32), (sx * sw)[None, :] * (acc.float() + torch.round(ox) * rsw.float()[None, :]))
    assert torch.equal(ops.linear_w8a8(xq, wq, sx, ox, sw, None, out_dtype=torch.float16),
                       ((sx * sw)[None, :] * (acc.float() + torch.round(ox) * rsw.float()[None, :])).to(torch.float16))


@pytest.mark.parametrize("n,k", [(14336, 4096), (28672, 8192)], ids=lambda v: str(v))
def test_mlp_mode_equals_integer_ground_truth_at_full_size(n, k):
    """gate_proj + up_proj + SiLU*up + quantize in one launch at T = 16384 (8B and 70B): the int8 codes equal
    A1( bf16(silu(bf16(gate))) * bf16(up) ) with gate / up formed from the EXACT accumulators by torch ops — ATen's silu
    and multiply on the device (the chain the reference's MLP runs, quantized_llama/mlp.py:30-40) and the A1 kernel,
    none of which is the GEMM under test."""
    g = torch.Generator(device=DEV).manual_seed(n + k)
    xq = torch.randint(-128, 128, (T, k), device=DEV, dtype=torch.int8, generator=g)
    gq = torch.randint(-128, 128, (n, k), device=DEV, dtype=torch.int8, generator=g)
    uq = torch.randint(-128, 128, (n, k), device=DEV, dtype=torch.int8, generator=g)
    sx, ox = torch.tensor([0.02], device=DEV), torch.tensor([4.0], device=DEV)
    sg = torch.rand(n, device=DEV, generator=g) * 1e-5 + 2e-5
    su = torch.rand(n, device=DEV, generator=g) * 1e-5 + 2e-5
    so, oo = torch.tensor([0.004], device=DEV), torch.tensor([-9.0], device=DEV)
    fused = ops.mlp_gate_up_w8a8(xq, gq, uq, sx, ox, sg, su, so, oo, 8)
    assert fused is not None
    rsg, rsu = gq.sum(dim=1, dtype=torch.int64).float(), uq.sum(dim=1, dtype=torch.int64).float()
    differing = 0
    gq64, uq64 = gq.double(), uq.double()
    for r0 in range(0, T, 1024):
        rows = slice(r0, r0 + 1024)
        gate = ((sx * sg)[None, :] * (exact_accumulators(xq, gq64, rows).float() + ox * rsg[None, :])).to(torch.bfloat16)
        up = ((sx * su)[None, :] * (exact_accumulators(xq, uq64, rows).float() + ox * rsu[None, :])).to(torch.bfloat16)
        z = torch.nn.functional.silu(gate) * up
        want = ops.quantize_by_tile(z, so, z.shape, 8, torch.int8, oo)
        differing += int((fused[rows] != want).sum())
        if r0 == 0:
            assert float(want.float().std()) > 1  # not a saturated tensor
        del gate, up, z, want
    assert differing == 0, f"{differing} of {fused.numel()} codes differ from the independent chain"


@pytest.mark.parametrize("n,k", [(4096, 4096), (8192, 28672)], ids=lambda v: str(v))
def test_weight_offset_and_per_token_paths_at_full_size(n, k):
    """Real weight offsets (the ow * sum_k xq and K * ox * ow terms from the side reduction over the activation codes) and
    per-token activation parameters in the persistent kernel at T = 16384: every output equals the fp32 epilogue restated
    on the exact integer contraction and the exact row sums."""
    xq, wq, g = _codes(n, k, n - k)
    sx = torch.rand(T, device=DEV, generator=g) * 0.01 + 0.01
    ox = torch.round(torch.randn(T, device=DEV, generator=g) * 20)
    sw = torch.rand(n, device=DEV, generator=g) * 1e-3 + 2e-4
    ow = torch.round(torch.randn(n, device=DEV, generator=g) * 3)
    got = ops.linear_w8a8(xq, wq, sx, ox, sw, ow, out_dtype=torch.float32)
    got16 = ops.linear_w8a8(xq, wq, sx, ox, sw, ow, out_dtype=torch.bfloat16)
    rsw = wq.sum(dim=1, dtype=torch.int64).float()
    rsx = xq.sum(dim=1, dtype=torch.int64).float()
    wq64 = wq.double()
    step = 2048 if n * k <= 4096 * 4096 else 1024
    for r0 in range(0, T, step):
        rows = slice(r0, r0 + step)
        acc = exact_accumulators(xq, wq64, rows)
        assert int(acc.abs().max()) < 2**24
        # csrc/ffq_linear.hip: v = float(acc) + ox * rsw;  v += ow * rsx;  v += (K * ox) * ow;  y = (sx * sw) * v
        v = acc.float() + ox[rows, None] * rsw[None, :]
        v = v + ow[None, :] * rsx[rows, None]
        v = v + (float(k) * ox[rows, None]) * ow[None, :]
        y = (sx[rows, None] * sw[None, :]) * v
        assert torch.equal(got[rows], y), f"rows {r0}..: {int((got[rows] != y).sum())} outputs differ"
        assert torch.equal(got16[rows], y.to(torch.bfloat16))
        del acc, v, y


@pytest.mark.parametrize("n,k", LLAMA8B + [(28672, 8192)], ids=lambda v: str(v))
def test_requantizing_epilogue_at_full_size(n, k):
    """The output quantizer inside the GEMM's epilogue (fallback.py:110-111) at T = 16384: int8 codes == A1 (the kernel pinned
    by G1-G3) of the bf16 tensor the plain launch's epilogue forms from the exact accumulators."""
    xq, wq, g = _codes(n, k, 5 * n + k)
    sx, ox = torch.tensor([0.0173], device=DEV), torch.tensor([11.3], device=DEV)
    sw = torch.rand(n, device=DEV, generator=g) * 1e-3 + 2e-4
    rsw = wq.sum(dim=1, dtype=torch.int64).float()
    wq64 = wq.double()
    chunk = 2048 if n * k <= 14336 * 4096 else 1024

    def plain_output(rows):
        return ((sx * sw)[None, :] * (exact_accumulators(xq, wq64, rows).float() + torch.round(ox) * rsw[None, :])).to(torch.bfloat16)

    # a grid that resolves the output (about 50 steps per standard deviation... of which 256 exist: the tails clip)
    so, oo = (plain_output(slice(0, chunk)).float().std() / 50).reshape(1), torch.tensor([-7.4], device=DEV)
    got = ops.linear_w8a8(xq, wq, sx, ox, sw, None, out_dtype=torch.int8, out_scale=so, out_offset=oo, out_num_bits=8, requant_from=torch.bfloat16)
    differing = 0
    for r0 in range(0, T, chunk):
        rows = slice(r0, r0 + chunk)
        y = plain_output(rows)
        want = ops.quantize_by_tile(y, so, y.shape, 8, torch.int8, oo)
        differing += int((got[rows] != want).sum())
        if r0 == 0:
            assert float(want.float().std()) > 30 and int(want.max()) == 127 and int(want.min()) == -128  # a real grid that also clips
        del y, want
    assert differing == 0, f"{differing} of {got.numel()} codes differ"


# ---- A1 / A2 / A4 on the 70B tensors (the 8B ones: tests/test_parity_gpu.py) ------------------------------------------
@pytest.mark.parametrize("shape", LLAMA70B, ids=str)
def test_70b_weight_properties(shape):
    torch.manual_seed(4321 + shape[0])
    w = (torch.randn(shape, device=DEV) * 0.02).to(torch.bfloat16)
    quantizer = ff.nn.LinearQuantizer(8, granularity=ff.PerChannel(0), quantized_dtype=torch.int8, device=DEV)
    with ff.estimate_ranges(torch.nn.ModuleList([quantizer]), ff.range_setting.running_minmax, sync_free=True):
        q = quantizer(w)
    codes = q.raw_data
    assert codes.dtype == torch.int8
    assert bool((codes.float().abs().amax(1) >= 127).all())  # every row reaches the end of its symmetric grid
    lo, hi = w.float().amin(1).cpu(), w.float().amax(1).cpu()  # CPU: true division (torch's GPU tensor / scalar multiplies by 1/x)
    assert torch.equal(quantizer.scale.detach().cpu(), torch.maximum(lo.abs() / 128, hi.abs() / 127))
    deq = q.dequantize()
    assert torch.equal(quantizer(deq).raw_data, codes)  # idempotence
    err = (w.float() - deq.float()).abs()
    assert bool((err <= quantizer.scale.detach()[:, None] * 0.5 + deq.float().abs() * 2.0**-8).all())
    # checksum of the codes against plain torch ops in fp32, in row chunks (tensor / tensor is a true division on the GPU)
    total = 0
    for wc, sc in zip(w.chunk(8), quantizer.scale.detach().chunk(8)):
        total += int(torch.clamp(torch.round(wc.float() / sc[:, None]), -128, 127).sum(dtype=torch.int64))
    assert int(codes.sum(dtype=torch.int64)) == total
    # W4 group-128: fused quantize+pack == A1 then A7; unpack round trip
    tile = (1, 128)
    glo, ghi = ops.minmax_by_tile(w, tile)
    gs, go = ops.parameters_for_range(glo, ghi, 4, True, True)
    packed = ops.quantize_pack_int4(w, gs, tile, go, block=128)
    q4 = ops.quantize_by_tile(w, gs, tile, 4, torch.int8, go)
    assert torch.equal(packed, ops.pack_int4(q4, block=128))
    assert torch.equal(ops.unpack_int4(packed, w.shape, torch.int8, block=128), q4)


@pytest.mark.parametrize("hidden", [8192, 28672])
def test_70b_activation_properties(hidden):
    torch.manual_seed(77)
    x = torch.randn(8, 2048, hidden, device=DEV, dtype=torch.bfloat16)
    quantizer = ff.nn.LinearQuantizer(8, symmetric=False, quantized_dtype=torch.int8, device=DEV)
    with ff.estimate_ranges(torch.nn.ModuleList([quantizer]), ff.range_setting.running_minmax, sync_free=True):
        q = quantizer(x)
    lo, hi = x.float().min().cpu(), x.float().max().cpu()
    scale = ((hi - lo) / 255).clamp(torch.finfo(torch.float32).eps)
    assert torch.equal(quantizer.scale.detach().cpu(), scale.reshape(1))
    assert torch.equal(quantizer.offset.detach().cpu(), (lo / scale + 128).reshape(1))
    assert int(q.raw_data.min()) == -128 and int(q.raw_data.max()) == 127
    dyn = ff.quantization.affine.dynamic.quantize_per_tensor(x, 8, symmetric=False, output_dtype=torch.int8)
    assert torch.equal(dyn.raw_data, q.raw_data)
    off = torch.round(quantizer.offset.detach())
    total = 0
    for chunk in x.chunk(16):
        total += int(torch.clamp(torch.round(chunk.float() / quantizer.scale.detach() - off), -128, 127).sum(dtype=torch.int64))
    assert int(q.raw_data.sum(dtype=torch.int64)) == total
    # A2 at full size: every dequantized value is (code + round(offset)) * scale rounded once to bf16
    deq = q.dequantize()
    for dc, cc in zip(deq.chunk(16), q.raw_data.chunk(16)):
        assert torch.equal(dc, ((cc.float() + off) * quantizer.scale.detach()).to(torch.bfloat16))


# ---- weight-only linear at full size: an input family on which EVERY fp32 summation order gives the same sum ---------
@pytest.mark.parametrize("n,k", LLAMA8B + LLAMA70B, ids=lambda v: str(v))
@pytest.mark.parametrize("group", [None, 128], ids=["per_channel", "group128"])
def test_weight_only_linear_is_exact_where_the_sum_is_order_independent(n, k, group):
    """T = 16384. Activations are small integers (exact in bf16), scales are powers of two, so every product is an integer
    multiple of 2^-4 and every partial sum stays below 2^24 of those units: fp32 accumulation is exact in ANY order and the
    bf16 output is the correctly rounded exact value — compared bit for bit with a float64 product of the same operands
    (rocBLAS dgemm, chunked), for W8 per output channel (BASELINE config 2) and W4 group-128 (config 4)."""
    g = torch.Generator(device=DEV).manual_seed(n * 3 + k + (group or 0))
    bits = 8 if group is None else 4
    amp = 4 if k > 16384 else 8
    x = torch.randint(-amp, amp + 1, (T, k), device=DEV, generator=g).to(torch.bfloat16)
    codes = torch.randint(-(2 ** (bits - 1)), 2 ** (bits - 1), (n, k), device=DEV, dtype=torch.int8, generator=g)
    grp = k if group is None else group
    exps = torch.randint(3, 5, (n * (k // grp),), device=DEV, generator=g)
    scale = torch.pow(2.0, -exps.float())
    w64 = (codes.double().view(n, k // grp, grp) * scale.double().view(n, k // grp, 1)).view(n, k)
    chunk = 2048 if n * k <= 14336 * 4096 else 1024
    # the codes as int8 (converted inside the GEMM / by A2 as its own pass) and, for W4, as packed nibbles (config 4's storage)
    forms = [("int8, one pass", codes, dict(two_pass=False)), ("int8, two passes", codes, dict(two_pass=True))]
    if bits == 4:
        packed = ops.pack_int4(codes, block=128)
        forms += [("packed nibbles, one pass", packed, dict(pack_block=128, two_pass=False)), ("packed nibbles, two passes", packed, dict(pack_block=128, two_pass=True))]
    for label, weight, kwargs in forms:
        y = ops.linear_wq(x, weight, scale, None, group=grp, **kwargs)
        assert y is not None, label
        for r0 in range(0, T, chunk):
            ref = x[r0:r0 + chunk].double() @ w64.t()
            assert float(ref.abs().max()) * 16 < 2**24
            assert torch.equal(y[r0:r0 + chunk], ref.to(torch.bfloat16)), f"{label}, rows {r0}..: {int((y[r0:r0 + chunk] != ref.to(torch.bfloat16)).sum())} outputs differ"
            del ref
        del y
