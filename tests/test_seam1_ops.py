"""Seam 1: the four torch custom ops ``torch.ops.fastforward_amd.*`` (the reference's ``fastforward::*`` schemas,
src/fastforward/quantization/_quantizer_impl.py:144-285, fake impls :288-339).

CPU part: schemas equal the reference's, every op has a Meta kernel whose outputs match the real ones in shape / dtype
(FakeTensorMode), and the autograd wrappers (affine/_autograd.py) reach the kernels THROUGH the operator registry.
GPU part (-m gpu): the same ops on HIP tensors equal the direct C-ABI wrappers, and ``torch.library.opcheck`` passes.
"""

from unittest import mock

import pytest
import torch

import fastforward_amd as ff

from fastforward_amd import ops

# printed from the live reference ops; `fastforward::` -> `fastforward_amd::`
REFERENCE_SCHEMAS = {
    "quantize_by_tile": "fastforward_amd::quantize_by_tile(Tensor data, Tensor scale, SymInt[] tile_size, float num_bits, ScalarType? output_dtype, Tensor? offset=None) -> Tensor",
    "dequantize_by_tile": "fastforward_amd::dequantize_by_tile(Tensor data, Tensor scale, SymInt[] tile_size, Tensor? offset=None, ScalarType? output_dtype=None) -> Tensor",
    "quantize_dynamic_by_tile": "fastforward_amd::quantize_dynamic_by_tile(Tensor data, SymInt[] tile_size, float num_bits, bool symmetric, bool allow_one_sided, ScalarType? output_dtype) -> (Tensor, Tensor, Tensor)",
    "quantize_by_tile_backward": "fastforward_amd::quantize_by_tile_backward(Tensor data, Tensor output_grad, Tensor scale, SymInt[] tile_size, float num_bits, Tensor? offset=None) -> Tensor[]",
}


def _samples(device, dtype=torch.float32):
    g = torch.Generator().manual_seed(5)
    x = torch.randn(16, 64, generator=g).to(dtype).to(device)
    grad = torch.randn(16, 64, generator=g).to(dtype).to(device)
    scale = (torch.rand(16, generator=g) * 0.05 + 0.01).to(device)
    offset = (torch.randn(16, generator=g) * 3).to(device)
    return x, grad, scale, offset


def _call_all(x, grad, scale, offset):
    o = torch.ops.fastforward_amd
    q = o.quantize_by_tile(x, scale, [1, 64], 8.0, torch.int8, offset)
    return {
        "quantize": q,
        "quantize_default_dtype": o.quantize_by_tile(x, scale, [1, 64], 4.0, None, None),
        "dequantize": o.dequantize_by_tile(q, scale, [1, 64], offset, x.dtype),
        "dynamic": o.quantize_dynamic_by_tile(x, [1, 64], 8.0, False, True, torch.int8),
        "backward": o.quantize_by_tile_backward(x, grad, scale, [1, 64], 8.0, offset),
        "backward_no_offset": o.quantize_by_tile_backward(x, grad, scale, [1, 64], 8.0, None),
    }


def _flat(result):
    out = []
    for key, value in result.items():
        for i, t in enumerate(value if isinstance(value, (list, tuple)) else [value]):
            out.append((f"{key}[{i}]", t))
    return out


def test_schemas_are_the_reference_schemas():
    for name, want in REFERENCE_SCHEMAS.items():
        assert str(getattr(torch.ops.fastforward_amd, name).default._schema) == want


def test_every_op_has_a_meta_kernel_that_matches_the_real_outputs(oracle_backend):
    from torch._subclasses.fake_tensor import FakeTensorMode

    real = _flat(_call_all(*_samples("cpu")))
    with FakeTensorMode() as mode:
        fake_inputs = [mode.from_tensor(t) for t in _samples("cpu")]
        fake = _flat(_call_all(*fake_inputs))
    assert [k for k, _ in real] == [k for k, _ in fake]
    for (name, r), (_, f) in zip(real, fake):
        assert tuple(r.shape) == tuple(f.shape) and r.dtype == f.dtype, (name, r.shape, f.shape, r.dtype, f.dtype)


def test_autograd_wrappers_go_through_the_operator_registry(oracle_backend):
    """LinearQuantizer -> AffineQuantizationFunction -> _autograd -> torch.ops.fastforward_amd.* -> ops.* (reference call
    sites affine/_autograd.py:86,99,121,148)."""
    x, grad, scale, offset = _samples("cpu")
    x.requires_grad_(True)
    quantizer = ff.nn.LinearQuantizer(8, symmetric=False, granularity=ff.PerChannel(0))
    quantizer.quantization_range = (x.detach().min(1).values, x.detach().max(1).values)
    seen = []
    original = torch.ops.fastforward_amd.quantize_by_tile.default.__call__

    class Spy(torch.utils._python_dispatch.TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            if func.namespace == "fastforward_amd":
                seen.append(func._schema.name)
            return func(*args, **(kwargs or {}))

    with Spy():
        q = quantizer(x)
        deq = q.dequantize()
        deq.backward(grad)
        ff.quantization.affine.dynamic.quantize_per_tensor(x.detach(), 8)
    assert seen == ["fastforward_amd::quantize_by_tile", "fastforward_amd::dequantize_by_tile",
                    "fastforward_amd::quantize_by_tile_backward", "fastforward_amd::quantize_dynamic_by_tile"], seen
    assert x.grad is not None and quantizer.scale.grad is not None and quantizer.offset.grad is not None
    del original


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_registered_ops_on_the_gpu_equal_the_c_abi_wrappers(hip_backend, dtype):
    x, grad, scale, offset = _samples("cuda", dtype)
    got = _call_all(x, grad, scale, offset)
    q = ops.quantize_by_tile(x, scale, (1, 64), 8, torch.int8, offset)
    assert torch.equal(got["quantize"], q)
    assert torch.equal(got["quantize_default_dtype"], ops.quantize_by_tile(x, scale, (1, 64), 4, None, None))
    assert torch.equal(got["dequantize"], ops.dequantize_by_tile(q, scale, (1, 64), offset, dtype))
    for a, b in zip(got["dynamic"], ops.quantize_dynamic_by_tile(x, (1, 64), 8, False, True, torch.int8)):
        assert torch.equal(a, b)
    for a, b in zip(got["backward"], ops.quantize_by_tile_backward(x, grad, scale, (1, 64), 8.0, offset)):
        assert torch.equal(a, b)
    # the oracle says the same (seeded input, codes bit-exact)
    from conftest import load_oracle, use_backend

    with use_backend(load_oracle()):
        want = ops.quantize_by_tile(x.cpu(), scale.cpu(), (1, 64), 8, torch.int8, offset.cpu())
    assert torch.equal(got["quantize"].cpu(), want)


@pytest.mark.gpu
def test_opcheck_of_the_four_ops(hip_backend):
    x, grad, scale, offset = _samples("cuda")
    q = ops.quantize_by_tile(x, scale, (1, 64), 8, torch.int8, offset)
    o = torch.ops.fastforward_amd
    tests = ("test_schema", "test_faketensor", "test_autograd_registration")
    torch.library.opcheck(o.quantize_by_tile.default, (x, scale, [1, 64], 8.0, torch.int8, offset), test_utils=tests)
    torch.library.opcheck(o.quantize_by_tile.default, (x, scale, [1, 64], 8.0, None), test_utils=tests)
    torch.library.opcheck(o.dequantize_by_tile.default, (q, scale, [1, 64], offset, torch.float32), test_utils=tests)
    torch.library.opcheck(o.quantize_dynamic_by_tile.default, (x, [16, 64], 8.0, True, True, torch.int8), test_utils=tests)
    torch.library.opcheck(o.quantize_by_tile_backward.default, (x, grad, scale, [1, 64], 8.0, offset), test_utils=tests)
