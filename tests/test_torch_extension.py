"""csrc/ffq_torch.cpp -> libffq_torch.so: the four operators of the reference's registry and the hot entry points behind its
dispatcher and range estimator (linear_w8a8, bmm_w8a8, linear_wq, running_minmax_step) registered for the HIP dispatch key in
C++ on top of the C ABI ("torch extension -> C ABI -> HIP").

CPU: the library is built, loads beside PyTorch, registers device kernels for all of them and leaves the host
route alone (host tensors still reach the Python implementation, which refuses them: there is no CPU path).
GPU: `torch.ops.fastforward_amd.*` (C++) and `ops.*` (Python -> ctypes) call the same kernels — identical bits, identical
error types — and the C++ route launches on torch's CURRENT stream (a hipGraph capture of it replays).
"""

import pytest
import torch

from fastforward_amd import ops
from fastforward_amd.exceptions import BackendError

OPS = torch.ops.fastforward_amd


def _dump(name: str) -> str:
    return torch._C._dispatch_dump(f"fastforward_amd::{name}")


def test_the_extension_is_built_and_registers_every_operator():
    assert ops.TORCH_EXTENSION_PATH.exists(), "build it with: make -C fastforward_amd/csrc"
    assert ops.NATIVE_DISPATCH
    for name in ("quantize_by_tile", "dequantize_by_tile", "quantize_dynamic_by_tile", "quantize_by_tile_backward", "running_minmax_step",
                 "linear_w8a8", "bmm_w8a8", "linear_wq"):
        assert "CUDA: registered at ffq_torch.cpp" in _dump(name), _dump(name)
        assert "CompositeExplicitAutograd" in _dump(name)  # the Python body stays registered (serves without the extension)


def test_the_extension_links_the_product_library_and_nothing_of_the_oracle():
    import subprocess

    needed = subprocess.run(["readelf", "-d", str(ops.TORCH_EXTENSION_PATH)], capture_output=True, text=True, check=True).stdout
    assert "libffq_hip.so" in needed and "$ORIGIN" in needed
    assert "oracle" not in needed


def test_host_tensors_do_not_reach_the_device_kernels():
    """The C++ kernels sit under the device key only: a host tensor reaches the operator's Python body, which runs the reference's
    ATen chain for the four registry operators (fastforward_amd/_host.py) and refuses everything that exists as a kernel only."""
    x, s = torch.randn(4, 8), torch.tensor([0.1])
    q = OPS.quantize_by_tile(x, s, [4, 8], 8.0, torch.int8, None)
    assert torch.equal(q, torch.clamp(torch.round(x / 0.1), -128, 127).to(torch.int8))
    assert torch.equal(OPS.dequantize_by_tile(q, s, [4, 8], None, None), q * s)
    with pytest.raises(BackendError, match="no CPU"):
        OPS.linear_w8a8(q, q, s, None, s, None, None, torch.bfloat16, None, None, 8.0, None, None)


CASES = [
    # (shape, tile, data dtype, scale dtype, offset dtype or None, num_bits, output dtype or None)
    ((64, 256), (64, 256), torch.bfloat16, torch.float32, torch.float32, 8, torch.int8),
    ((64, 256), (1, 256), torch.bfloat16, torch.float32, None, 8, torch.int8),
    ((64, 256), (1, 256), torch.float32, torch.float32, torch.float32, 4, None),
    ((64, 256), (64, 1), torch.float16, torch.float16, None, 8, None),
    ((8, 4, 128), (1, 4, 32), torch.bfloat16, torch.bfloat16, torch.bfloat16, 3, torch.float32),
    ((2048,), (128,), torch.float32, torch.float64, None, 8, torch.int16),
    ((), (), torch.float32, torch.float32, torch.float32, 8, None),
]


def _params(shape, tile, scale_dtype, offset_dtype, device):
    ntiles = 1
    for s, t in zip(shape, tile):
        ntiles *= s // t
    g = torch.Generator().manual_seed(ntiles)
    scale = (torch.rand(ntiles, generator=g) * 0.05 + 0.01).to(scale_dtype).to(device)
    offset = None if offset_dtype is None else (torch.randn(ntiles, generator=g) * 3).to(offset_dtype).to(device)
    return scale, offset


@pytest.mark.gpu
@pytest.mark.parametrize("shape,tile,dtype,scale_dtype,offset_dtype,bits,out_dtype", CASES)
def test_both_routes_give_the_same_bits(shape, tile, dtype, scale_dtype, offset_dtype, bits, out_dtype):
    assert ops.NATIVE_DISPATCH
    torch.manual_seed(0)
    x = torch.randn(shape).to(dtype).cuda()
    scale, offset = _params(shape, tile, scale_dtype, offset_dtype, "cuda")
    q_cpp = OPS.quantize_by_tile(x, scale, list(tile), float(bits), out_dtype, offset)
    q_py = ops.quantize_by_tile(x, scale, tile, bits, out_dtype, offset)
    assert q_cpp.dtype == q_py.dtype and q_cpp.shape == q_py.shape and torch.equal(q_cpp, q_py)
    for back in (None, dtype):
        d_cpp = OPS.dequantize_by_tile(q_cpp, scale, list(tile), offset, back)
        d_py = ops.dequantize_by_tile(q_py, scale, tile, offset, back)
        assert d_cpp.dtype == d_py.dtype and torch.equal(d_cpp, d_py)


@pytest.mark.gpu
def test_strided_inputs_and_shaped_parameters():
    x = torch.randn(256, 64, device="cuda", dtype=torch.bfloat16).t()  # [64, 256], not contiguous
    scale = torch.rand(64, 1, device="cuda") * 0.05 + 0.01  # the parameter shape a PerChannel quantizer keeps
    q_cpp = OPS.quantize_by_tile(x, scale, [1, 256], 8.0, torch.int8, None)
    assert q_cpp.is_contiguous() and torch.equal(q_cpp, ops.quantize_by_tile(x, scale, (1, 256), 8, torch.int8))
    assert torch.equal(OPS.dequantize_by_tile(q_cpp.t().contiguous().t(), scale, [1, 256], None, torch.bfloat16),
                       ops.dequantize_by_tile(q_cpp, scale, (1, 256), None, torch.bfloat16))


@pytest.mark.gpu
def test_the_reference_error_types_come_out_of_the_cpp_route():
    x = torch.randn(10, 4, device="cuda")
    s = torch.tensor([0.1], device="cuda")
    with pytest.raises(ValueError, match="dimensionality"):  # tiled_tensor.py:24-29
        OPS.quantize_by_tile(x, s, [4], 8.0, None, None)
    with pytest.raises(ValueError, match="divi"):  # tiled_tensor.py:31-42
        OPS.quantize_by_tile(x, s, [3, 4], 8.0, None, None)
    with pytest.raises(RuntimeError):  # scale[:, None] does not broadcast over the tiles
        OPS.quantize_by_tile(x, torch.ones(3, device="cuda"), [5, 4], 8.0, None, None)
    with pytest.raises(RuntimeError, match="same device"):
        OPS.quantize_by_tile(x, s.cpu(), [10, 4], 8.0, None, None)
    with pytest.raises(NotImplementedError, match="not supported"):
        OPS.dequantize_by_tile(x.to(torch.complex64), s, [10, 4], None, None)
    for route in (lambda: OPS.quantize_by_tile(x, s, [10, 4], 40.0, torch.int8, None), lambda: ops.quantize_by_tile(x, s, (10, 4), 40, torch.int8)):
        with pytest.raises(RuntimeError):  # _quantizer_impl.py:165-167: the container cannot hold the bit width
            route()


@pytest.mark.gpu
def test_the_cpp_route_launches_on_the_current_stream():
    x = torch.randn(512, 256, device="cuda", dtype=torch.bfloat16)
    scale = torch.rand(512, device="cuda") * 0.05 + 0.01
    want = ops.quantize_by_tile(x, scale, (1, 256), 8, torch.int8)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        OPS.quantize_by_tile(x, scale, [1, 256], 8.0, torch.int8, None)  # allocator warm-up on the side stream
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            got = OPS.quantize_by_tile(x, scale, [1, 256], 8.0, torch.int8, None)
            back = OPS.dequantize_by_tile(got, scale, [1, 256], None, torch.bfloat16)
    torch.cuda.synchronize()
    x.copy_(x.flip(0))  # new contents, same addresses: only a replay of the captured launches can produce the new codes
    want = ops.quantize_by_tile(x, scale, (1, 256), 8, torch.int8)
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    assert torch.equal(back, ops.dequantize_by_tile(want, scale, (1, 256), None, torch.bfloat16))


@pytest.mark.gpu
def test_quantizer_modules_run_through_the_cpp_route():
    """LinearQuantizer -> affine/_autograd.py -> torch.ops.fastforward_amd.quantize_by_tile: the registry route is the C++ one."""
    import fastforward_amd as ff

    q = ff.nn.LinearQuantizer(8, granularity=ff.PerChannel(0), quantized_dtype=torch.int8, device="cuda")
    w = (torch.randn(128, 256, device="cuda") * 0.02).to(torch.bfloat16)
    with ff.estimate_ranges(q, ff.range_setting.running_minmax):
        q(w)
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU]) as prof:
        codes = q(w)
    names = {e.name for e in prof.events()}
    assert "fastforward_amd::quantize_by_tile" in names
    assert torch.equal(codes.raw_data, ops.quantize_by_tile(w, q.scale, (1, 256), 8, torch.int8, q.offset))


@pytest.mark.gpu
def test_the_dynamic_operator_and_the_backward_run_in_cpp_with_the_same_results_and_errors():
    """quantize_dynamic_by_tile / quantize_by_tile_backward: dispatcher -> C++ == the Python bodies bit for bit, the package's own
    QuantizationError for an empty input (reference _quantizer_impl.py:259-264) raised from C++ through the CPython API, the
    composite backward (strided channels, half-precision parameters) as ATen calls."""
    from fastforward_amd.exceptions import QuantizationError

    g = torch.Generator().manual_seed(4)
    x = torch.randn(64, 512, generator=g).to(torch.bfloat16).cuda()
    for tile, sym, one in (((1, 512), False, True), ((1, 512), True, True), ((64, 512), False, True), ((1, 128), True, False), ((64, 1), False, True)):
        for qdt in (torch.int8, None):
            got, want = OPS.quantize_dynamic_by_tile(x, list(tile), 8.0, sym, one, qdt), ops.quantize_dynamic_by_tile(x, tile, 8, sym, one, qdt)
            for a, b in zip(got, want):
                assert a.dtype == b.dtype and torch.equal(a, b)
    with pytest.raises(QuantizationError, match="empty"):
        OPS.quantize_dynamic_by_tile(x[:0], [1, 512], 8.0, False, True, torch.int8)
    with pytest.raises(ValueError, match="dimensionality"):
        OPS.quantize_dynamic_by_tile(x, [512], 8.0, False, True, torch.int8)
    grad = torch.randn(64, 512, generator=g).to(torch.bfloat16).cuda()
    for tile, scale_dtype, with_offset in (((1, 512), torch.float32, True), ((64, 512), torch.float32, False), ((64, 1), torch.float32, True),
                                           ((1, 512), torch.bfloat16, True), ((8, 64), torch.float32, True)):
        n = (64 // tile[0]) * (512 // tile[1])
        scale = (torch.rand(n, generator=g) * 0.05 + 0.01).to(scale_dtype).cuda()
        offset = (torch.randn(n, generator=g) * 3).to(scale_dtype).cuda() if with_offset else None
        got = OPS.quantize_by_tile_backward(x, grad, scale, list(tile), 8.0, offset)
        want = ops.quantize_by_tile_backward(x, grad, scale, tile, 8.0, offset)
        assert len(got) == 3
        for a, b in zip(got, want):
            assert a.shape == b.shape and a.dtype == b.dtype and torch.equal(a, b), (tile, scale_dtype)


@pytest.mark.gpu
def test_estimator_step_and_gemm_entry_points_run_in_cpp_with_the_same_results():
    """running_minmax_step, linear_w8a8 (plain, bias, fused output quantizer, given row sums), bmm_w8a8, linear_wq (skinny rows,
    split-K tiles, packed nibbles, the two-pass form): torch.ops.fastforward_amd.* (C++) == the Python bodies (ctypes), and
    ops.* takes the C++ route by itself."""
    g = torch.Generator().manual_seed(9)
    x = torch.randn(4, 96, 256, generator=g).to(torch.bfloat16).cuda()
    for tile in ((4, 96, 256), (1, 1, 256)):
        n = 1 if tile[0] == 4 else 4 * 96
        states = []
        for route in (OPS.running_minmax_step, ops._running_minmax_step):
            lo, hi = torch.full((n,), float("inf"), dtype=torch.bfloat16, device="cuda"), torch.full((n,), float("-inf"), dtype=torch.bfloat16, device="cuda")
            scale, offset, flags = torch.empty(n, device="cuda"), torch.empty(n, device="cuda"), torch.zeros(1, dtype=torch.int32, device="cuda")
            for step in range(3):
                route(x * (step + 1), list(tile), lo, hi, flags, 8.0, False, True, scale, offset)
            states.append((lo, hi, scale, offset, flags))
        for a, b in zip(*states):
            assert torch.equal(a, b)
    xq = torch.randint(-128, 128, (300, 512), generator=g, dtype=torch.int8).cuda()
    wq = torch.randint(-128, 128, (384, 512), generator=g, dtype=torch.int8).cuda()
    sx, ox = torch.tensor([0.02]).cuda(), torch.tensor([3.0]).cuda()
    sw, ow = (torch.rand(384, generator=g) * 1e-2 + 1e-3).cuda(), torch.round(torch.randn(384, generator=g) * 2).cuda()
    bias = torch.randn(384, generator=g).to(torch.bfloat16).cuda()
    so, oo = torch.tensor([0.05]).cuda(), torch.tensor([-2.0]).cuda()
    rowsum = wq.int().sum(1).to(torch.int32)
    for args in ((xq, wq, sx, ox, sw, None, None, torch.bfloat16, None, None, 8.0, None, None), (xq, wq, sx, ox, sw, ow, bias, torch.float32, None, None, 8.0, None, None),
                 (xq, wq, sx, ox, sw, None, None, torch.int8, so, oo, 8.0, rowsum, torch.bfloat16), (xq, wq, sx, None, sw[:1], None, None, torch.bfloat16, so, None, 4.0, None, torch.float32)):
        got, want = OPS.linear_w8a8(*args), ops._linear_w8a8(*args)
        assert got.dtype == want.dtype and torch.equal(got, want)
    assert torch.equal(ops.linear_w8a8(xq, wq, sx, ox, sw, None), ops._linear_w8a8(xq, wq, sx, ox, sw, None, None, torch.bfloat16, None, None, 8.0, None, None))
    with pytest.raises(TypeError, match="int8"):
        OPS.linear_w8a8(xq.float(), wq, sx, ox, sw, None, None, torch.bfloat16, None, None, 8.0, None, None)
    xb, wb = xq[:256].reshape(4, 64, 512), wq[:256].reshape(4, 64, 512)
    args = (xb, wb, sx, ox, sw[:1], ow[:1], torch.bfloat16, None, None, 8.0, None)
    assert torch.equal(OPS.bmm_w8a8(*args), ops._bmm_w8a8(*args))
    w4 = torch.randint(-8, 8, (384, 512), generator=g, dtype=torch.int8).cuda()
    s4 = (torch.rand(384 * 4, generator=g) * 1e-1 + 1e-2).cuda()
    packed = ops.pack_int4(w4, block=128)
    for m in (1, 40, 300):
        xs = torch.randn(m, 512, generator=g).to(torch.bfloat16).cuda()
        for args in ((xs, wq, sw, None, 512, None, torch.bfloat16, 0, -1, 0), (xs, wq, sw, ow, 512, bias, torch.float32, 0, 0, 2), (xs, w4, s4, None, 128, None, torch.bfloat16, 0, 1, 0),
                     (xs, packed, s4, None, 128, None, torch.bfloat16, 128, -1, 0)):
            got, want = OPS.linear_wq(*args), ops._linear_wq(*args)
            assert got.dtype == want.dtype and torch.equal(got, want), (m, args[4:])


@pytest.mark.gpu
def test_split_k_launches_captured_into_a_graph_own_their_ticket_words():
    """A launch captured into a hipGraph takes ticket words of its own (zeroed by a node of the capture), on both routes: the replay
    reproduces the eager result and the eager buffer of the stream stays untouched (shared counters between a graph and
    eager launches)."""
    g = torch.Generator().manual_seed(2)
    x = torch.randn(64, 4096, generator=g).to(torch.bfloat16).cuda()
    w = torch.randint(-128, 128, (1024, 4096), generator=g, dtype=torch.int8).cuda()
    s = (torch.rand(1024, generator=g) * 1e-2 + 1e-3).cuda()
    want = ops.linear_wq(x, w, s, None)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ops.linear_wq(x, w, s, None)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            got_cpp = ops.linear_wq(x, w, s, None)
            got_py = ops._linear_wq(x, w, s, None, 4096, None, torch.bfloat16, 0, -1, 0)
    torch.cuda.current_stream().wait_stream(side)
    for _ in range(5):
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(got_cpp, want) and torch.equal(got_py, want)
