"""csrc/ffq_torch.cpp -> libffq_torch.so: the two static operators registered for the HIP dispatch key in C++ on top of the
C ABI ("torch extension -> C ABI -> HIP").

CPU: the library is built, loads beside PyTorch, registers device kernels for exactly the two operators and leaves the host
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


def test_the_extension_is_built_and_registers_the_two_static_operators():
    assert ops.TORCH_EXTENSION_PATH.exists(), "build it with: make -C fastforward_amd/csrc"
    assert ops.NATIVE_DISPATCH
    for name in ("quantize_by_tile", "dequantize_by_tile"):
        assert "CUDA: registered at ffq_torch.cpp" in _dump(name), _dump(name)
    for name in ("quantize_dynamic_by_tile", "quantize_by_tile_backward"):  # Python: QuantizationError / the composite backward
        assert "CUDA:" not in _dump(name)


def test_the_extension_links_the_product_library_and_nothing_of_the_oracle():
    import subprocess

    needed = subprocess.run(["readelf", "-d", str(ops.TORCH_EXTENSION_PATH)], capture_output=True, text=True, check=True).stdout
    assert "libffq_hip.so" in needed and "$ORIGIN" in needed
    assert "oracle" not in needed


def test_host_tensors_do_not_reach_the_device_kernels():
    x, s = torch.randn(4, 8), torch.tensor([0.1])
    with pytest.raises(BackendError, match="no CPU"):
        OPS.quantize_by_tile(x, s, [4, 8], 8.0, torch.int8, None)
    with pytest.raises(BackendError, match="no CPU"):
        OPS.dequantize_by_tile(x.to(torch.int8), s, [4, 8], None, None)


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
