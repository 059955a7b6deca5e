"""Host-side surface on CPU: the oracle stands in for the HIP library (same C ABI, host pointers).

Re-expresses, against fastforward_amd's imports, the assertions of the reference tests that pin this
path: tests/test_dispatcher.py, tests/quantization/test_tiled_tensor.py,
tests/nn/test_linear_quantizer.py (initialisation / error behaviour / per-channel == per-tensor),
tests/quantization/test_dynamic.py, tests/range_setting/test_minmax.py, plus conversion
(`quantize_model`), the override stack and QuantizedTensor's torch-function routing.
"""

import copy
import math
import pickle

from unittest import mock

import pytest
import torch

import fastforward_amd as ff

from conftest import golden
from fastforward_amd.dispatcher import DispatcherPriority, Predicate, dispatch, register
from fastforward_amd.exceptions import BackendError, QuantizationError
from fastforward_amd.quantization import tiled_tensor
from fastforward_amd.quantization.affine import AffineQuantizationFunction, StaticAffineQuantParams


@pytest.fixture(autouse=True)
def _backend(oracle_backend):
    yield


def spy(fn):
    return mock.Mock(wraps=fn)


# ---- dispatcher (reference tests/test_dispatcher.py) -----------------------------------------------
def test_predicate_algebra_short_circuits():
    m1, m2 = spy(lambda x, y: x == "test"), spy(lambda x, y: y == "not")
    p = Predicate(m1) | Predicate(m2)
    assert p("test", "not")
    m1.assert_called_once()
    m2.assert_not_called()
    m1.reset_mock(), m2.reset_mock()
    assert not p("not", "test")
    m1.assert_called_once()
    m2.assert_called_once()
    q = Predicate(m1) & Predicate(m2)
    m1.reset_mock(), m2.reset_mock()
    assert not q("nope", "not")
    m2.assert_not_called()
    assert (~Predicate(m1))("nope", "x")


def test_dispatch_order_is_newest_first_within_a_priority():
    op = "softmax_order_test"
    p1, k1 = spy(lambda input, dim: input.dtype == torch.int8), spy(lambda input, dim: "k1")
    p2, k2 = spy(lambda input, dim: dim == 1), spy(lambda input, dim: "k2")
    p3, k3 = spy(lambda input, dim: isinstance(input, str)), spy(lambda input, dim: "k3")
    hooks = [register(op, Predicate(p), k) for p, k in ((p1, k1), (p2, k2), (p3, k3))]
    try:
        x = torch.zeros(2, 3, dtype=torch.int8)
        kernel = dispatch(op, x, 1)
        assert kernel(x, 1) == "k2"
        p3.assert_called_once(), p2.assert_called_once(), p1.assert_not_called()
        k3.assert_not_called(), k1.assert_not_called()
        # fallbacks are consulted after every DEFAULT kernel
        fb = register(op, None, lambda input, dim: "fallback", DispatcherPriority.FALLBACK)
        hooks.append(fb)
        assert dispatch(op, x, 1)(x, 1) == "k2"
        assert dispatch(op, torch.zeros(1), 0)(x, 0) == "fallback"
    finally:
        for h in hooks:
            h.remove()
    assert dispatch(op, x, 1) is None


def test_registration_hook_and_decorator_forms():
    op = "hook_test"
    with register(op, Predicate(lambda a: a > 0), lambda a: a * 2):
        assert dispatch(op, 3)(3) == 6
        assert dispatch(op, -3) is None
    assert dispatch(op, 3) is None

    @register(op, Predicate(lambda a: True))
    def kernel(a):
        return a + 1

    try:
        assert kernel(1) == 2 and dispatch(op, 5) is kernel
    finally:
        ff.dispatcher._DISPATCHER[op].clear()


def test_linear_call_signatures_match_reference_capture():
    """What predicate/kernel see on both call paths equals what was captured from the reference (G9)."""
    seen = []

    def predicate(*args, **kwargs):
        seen.append((len(args), sorted(kwargs)))
        return False

    x = ff.quantization.affine.quantize_per_tensor(torch.randn(2, 16), 0.1, None, 8)
    w = ff.quantization.affine.quantize_per_tensor(torch.randn(4, 16), 0.1, None, 8)
    with ff.strict_quantization(False), register("linear", Predicate(predicate), lambda *a, **k: None):
        ff.nn.functional.linear(x, w)
        torch.nn.functional.linear(x, w)
    expected = [tuple(e) if isinstance(e, (list, tuple)) else e for e in golden("g9_dispatcher.pt")["functional_then_torch"]]
    assert [(n, list(k)) for n, k in seen] == [(n, list(k)) for n, k in expected]


def test_same_kernel_for_functional_torch_function_and_operator():
    a = ff.quantization.affine.quantize_per_tensor(torch.randn(5, 1, 3), 0.1, None, 8)
    b = ff.quantization.affine.quantize_per_tensor(torch.randn(1, 2, 1), 0.2, 3.0, 8)
    with ff.strict_quantization(False):
        expected = a.dequantize() + b.dequantize()
        torch.testing.assert_close(torch.add(a, b), expected)
        torch.testing.assert_close(a + b, expected)
        kernel = spy(lambda input, other, **kw: input.dequantize() - other.dequantize())
        with register("add", Predicate(lambda *a_, **k_: True), kernel):
            out1 = torch.add(a, b)
            out2 = a + b
        assert kernel.call_count == 2
        torch.testing.assert_close(out1, a.dequantize() - b.dequantize())
        torch.testing.assert_close(out2, out1)


# ---- tile layout (reference tests/quantization/test_tiled_tensor.py:10-43) ---------------------------
def test_tiles_to_rows_known_layout():
    blocks = [torch.ones(2, 2) * v for v in range(1, 9)]
    p1 = torch.stack([torch.hstack(blocks[0:2]), torch.hstack(blocks[2:4])]).reshape(4, 4)
    p2 = torch.stack([torch.hstack(blocks[4:6]), torch.hstack(blocks[6:8])]).reshape(4, 4)
    data = torch.stack([p1, p2])
    rows = tiled_tensor.tiles_to_rows(data, (1, 2, 2))
    torch.testing.assert_close(rows, torch.ones(8, 4) * torch.arange(1, 9)[:, None])
    torch.testing.assert_close(tiled_tensor.rows_to_tiles(rows, data.shape, (1, 2, 2)), data)
    for bad in ((1, 2, 3, 4), (3, 2, 2)):
        with pytest.raises(ValueError):
            tiled_tensor.tiles_to_rows(data, bad)
    with pytest.raises(ValueError):
        tiled_tensor.rows_to_tiles(rows[:, :-1], data.shape, (1, 2, 2))
    with pytest.raises(ValueError):
        tiled_tensor.rows_to_tiles(rows[:, :-1], data.shape, (2, 2, 2))


def test_kernel_parameter_order_is_tiles_to_rows_order():
    """The C ABI's tile index == row index of tiles_to_rows, for every layout class."""
    g = torch.Generator().manual_seed(0)
    for shape, tile in [((6, 8), (1, 8)), ((6, 8), (6, 1)), ((6, 8), (2, 4)), ((4, 6, 8), (4, 1, 8)), ((4, 6, 8), (1, 6, 1)), ((4, 6, 8), (2, 3, 4)), ((4, 6, 8), (1, 1, 4))]:
        x = torch.randn(*shape, generator=g)
        rows = tiled_tensor.tiles_to_rows(x, tile)
        lo, hi = ff.ops.minmax_by_tile(x, tile)
        assert torch.equal(lo, rows.min(-1).values) and torch.equal(hi, rows.max(-1).values), (shape, tile)
        scale = torch.rand(rows.shape[0], generator=g) + 0.1
        q = ff.ops.quantize_by_tile(x, scale, tile, 4, None)
        ref = tiled_tensor.rows_to_tiles(torch.clamp(torch.round(rows / scale[:, None]), -8, 7), shape, tile)
        assert torch.equal(q, ref), (shape, tile)


# ---- granularities -----------------------------------------------------------------------------------
def test_granularity_tile_sizes():
    shape = torch.Size((4, 6, 8))
    assert ff.PerTensor().tile_size(shape) == "data_shape"
    assert ff.PerChannel().tile_size(shape) == (1, 6, 8)
    assert ff.PerChannel(-1).tile_size(shape) == (4, 6, 1)
    assert ff.PerChannel((0, 2)).tile_size(shape) == (1, 6, 1)
    assert ff.PerBlock(2, 4, 0).tile_size(shape) == (1, 6, 4)
    assert ff.PerBlock(block_dims=1, block_sizes=128, per_channel_dims=0).tile_size(torch.Size((4096, 4096))) == (1, 128)
    assert ff.PerBlock(1, 128, 0).parameter_dimensionality(torch.Size((4096, 4096))) == 131072
    assert ff.PerTile((2, 3, 4)).tile_size(shape) == (2, 3, 4)
    with pytest.raises(ValueError):
        ff.PerTile((3, 3, 4)).tile_size(shape)
    with pytest.raises(ValueError):
        ff.PerBlock(2, 3, 0).tile_size(shape)
    with pytest.raises(ValueError):
        ff.PerBlock(2, 16, 0).tile_size(shape)
    assert ff.PerChannel(0) == ff.PerChannel((0,)) and ff.PerChannel(0) != ff.PerChannel(1) and ff.PerTensor() == ff.PerTensor()
    from fastforward_amd.quantization.granularity import granularity_from_sizes

    assert granularity_from_sizes(shape, shape) == ff.PerTensor()
    assert granularity_from_sizes(shape, torch.Size((1, 6, 8))) == ff.PerChannel(0)
    assert granularity_from_sizes(shape, torch.Size((1, 6, 4))).tile_size(shape) == (1, 6, 4)


# ---- LinearQuantizer (reference tests/nn/test_linear_quantizer.py) ---------------------------------
@pytest.mark.parametrize("quantized_dtype", [torch.float16, torch.float32, torch.int16])
@pytest.mark.parametrize("param_dtype", [torch.float16, torch.float32])
def test_linear_quantizer_initialisation_behaviour(quantized_dtype, param_dtype):
    data = torch.rand(32, 14, 17)
    quantizer = ff.nn.LinearQuantizer(2, symmetric=False, quantized_dtype=quantized_dtype, param_dtype=param_dtype)
    with pytest.raises(ValueError, match="uninitialized quantizer"):
        quantizer(data)
    assert quantizer.has_uninitialized_params and quantizer.quantization_range == (None, None)
    quantizer.quantization_range = (data.min(), data.max())
    q = quantizer(data)
    assert isinstance(q, ff.QuantizedTensor) and q.raw_data.dtype == quantized_dtype
    assert all(p.dtype == param_dtype for p in quantizer.parameters())


def test_linear_quantizer_range_setter_errors_and_defaults():
    quantizer = ff.nn.LinearQuantizer(2, symmetric=False)
    lo, hi = torch.tensor(0.1), torch.tensor(0.9)
    with pytest.raises(ValueError):
        quantizer.quantization_range = (lo, hi, hi)
    with pytest.raises(ValueError):
        quantizer.quantization_range = lo
    quantizer.quantization_range = (lo, hi)
    assert not quantizer.has_uninitialized_params
    # defaults of the reference: symmetric, one-sided allowed, offset is a BUFFER; None when not allowed
    default = ff.nn.LinearQuantizer(8)
    assert default.symmetric and "offset" in default._buffers and isinstance(default.granularity, ff.PerTensor)
    assert ff.nn.LinearQuantizer(8, allow_one_sided=False).offset is None
    assert isinstance(ff.nn.LinearQuantizer(8, symmetric=False).offset, torch.nn.Parameter)
    default.quantization_range = (-1.0, 2.0)
    assert torch.equal(default.offset, torch.zeros(1)) and default.offset.dtype == torch.float32
    lo_, hi_ = default.quantization_range
    torch.testing.assert_close(hi_, torch.tensor([2.0]))


@pytest.mark.parametrize("channel_dim", [(0,), (1,), (2,), (0, 1), (0, 2), (1, 2), (0, 1, 2)])
@pytest.mark.parametrize("data_shape", [(3, 4, 9), (32, 14, 17)])
def test_parameter_count_must_match_tile_count(data_shape, channel_dim):
    data = torch.rand(data_shape)
    n = math.prod(data.shape[d] for d in channel_dim)
    good = ff.nn.LinearQuantizer(2, symmetric=False, granularity=ff.PerChannel(channel_dim))
    good.quantization_range = (torch.zeros(n), torch.ones(n))
    assert good(data) is not None
    bad = ff.nn.LinearQuantizer(2, symmetric=False, granularity=ff.PerChannel(channel_dim))
    bad.quantization_range = (torch.zeros(2), torch.ones(2))
    with pytest.raises(RuntimeError):
        bad(data)


def test_per_channel_equals_stack_of_per_tensor():
    g = torch.Generator().manual_seed(3)
    data = torch.randn(6, 10, generator=g)
    scale, offset = torch.rand(6, generator=g) + 0.1, torch.randn(6, generator=g)
    whole = ff.quantization.affine.quantize_per_channel(data, scale, offset, 0, 3)
    for r in range(6):
        row = ff.quantization.affine.quantize_per_tensor(data[r], scale[r : r + 1], offset[r : r + 1], 3)
        assert torch.equal(whole.raw_data[r], row.raw_data)
        assert torch.equal(whole.dequantize()[r], row.dequantize())


def test_precision_guard_and_type_errors():
    with pytest.raises(RuntimeError, match="not enough"):
        ff.quantization.affine.quantize_by_tile(torch.tensor([257.0]), torch.tensor([1.0]), torch.tensor([0.0]), torch.Size((1,)), 16, output_dtype=torch.bfloat16)
    with pytest.raises(ValueError):  # rank mismatch between tile and data
        ff.quantization.affine.quantize_by_tile(torch.zeros(4, 4), torch.ones(1), None, torch.Size((4,)), 8)
    dyn = ff.quantization.affine.dynamic.quantization_context(ff.PerTensor(), 8)
    with pytest.raises(TypeError, match="dynamic"):
        AffineQuantizationFunction.dequantize(torch.zeros(3), dyn.quantization_params)
    with pytest.raises(TypeError):
        AffineQuantizationFunction.quantize(torch.zeros(3), object())
    with pytest.raises(QuantizationError):
        ff.quantization.affine.dynamic.quantize_per_tensor(torch.zeros(0), 8)


def test_dynamic_equals_static_with_minmax_range():
    g = torch.Generator().manual_seed(11)
    data = torch.randn(8, 20, generator=g)
    for symmetric in (False, True):
        dyn = ff.quantization.affine.dynamic.quantize_per_channel(data, 0, 4, symmetric=symmetric)
        lo, hi = data.min(1).values, data.max(1).values
        scale, offset = ff.quantization.affine.parameters_for_range(lo, hi, 4, symmetric=symmetric, allow_one_sided=True)
        stat = ff.quantization.affine.quantize_per_channel(data, scale, offset, 0, 4)
        assert torch.equal(dyn.raw_data, stat.raw_data)
        assert torch.equal(dyn.dequantize(), stat.dequantize())


def test_gradients_flow_through_the_backward_op():
    data = torch.linspace(-8, 8, 17, requires_grad=True)
    scale, offset = torch.tensor([2.0], requires_grad=True), torch.tensor([2.3], requires_grad=True)
    params = StaticAffineQuantParams(scale=scale, offset=offset, num_bits=2, granularity=ff.PerTensor())
    q = AffineQuantizationFunction.quantize(data, params)
    q.dequantize().sum().backward()
    for t in (data, scale, offset):
        assert t.grad is not None and torch.count_nonzero(t.grad)
    # straight-through inside the grid, zero on clipped elements
    inside = (torch.round(data.detach() / 2.0 - 2.0) >= -2) & (torch.round(data.detach() / 2.0 - 2.0) <= 1)
    assert torch.equal(data.grad != 0, inside)


# ---- QuantizedTensor -----------------------------------------------------------------------------------
def test_quantized_tensor_routing_and_strictness():
    q = ff.quantization.affine.quantize_per_tensor(torch.randn(4, 6), torch.tensor([0.05]), None, 8)
    assert isinstance(q, ff.QuantizedTensor) and q.shape == (4, 6) and q.dtype == torch.float32
    assert not isinstance(q.raw_data, ff.QuantizedTensor) and q.int_repr() is not q
    with pytest.raises(QuantizationError, match="strict_quantization"):
        torch.relu(q)
    with ff.strict_quantization(False):
        torch.testing.assert_close(torch.relu(q), torch.relu(q.dequantize()))
        assert q.float().dtype == torch.float32 and q.double().dtype == torch.float64
        assert q.to(torch.float16).dtype == torch.float16
    with pytest.raises(NotImplementedError):
        q[0]
    with pytest.raises(NotImplementedError):
        q.add_(1)
    with pytest.warns(UserWarning):
        assert q.is_quantized is False
    assert isinstance(q.view(4, 6), ff.QuantizedTensor) and isinstance(q.reshape(6, 4), ff.QuantizedTensor)
    assert isinstance(q.contiguous(), ff.QuantizedTensor)
    assert isinstance(q.transpose(0, 1).contiguous(), ff.QuantizedTensor)
    c = q.clone()
    assert torch.equal(c.raw_data, q.raw_data) and c.quant_args().scale is not q.quant_args().scale
    d = copy.deepcopy(q)
    assert torch.equal(d.dequantize(), q.dequantize())
    r = pickle.loads(pickle.dumps(q))
    assert isinstance(r, ff.QuantizedTensor) and torch.equal(r.dequantize(), q.dequantize())
    assert "quant_func=AffineQuantizationFunction" in repr(q)
    with pytest.raises(ValueError):
        q.to(torch.zeros(1))


def test_per_channel_view_requires_same_shape():
    q = ff.quantization.affine.quantize_per_channel(torch.randn(4, 6), torch.ones(4), None, 0, 8)
    assert isinstance(q.view(4, 6), ff.QuantizedTensor)  # same shape: always allowed
    with pytest.raises(QuantizationError):
        q.view(6, 4)  # no kernel for per-channel reshape -> dequantization fallback -> strict error


# ---- modules: conversion, overrides, estimators ------------------------------------------------------------
def make_model():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.Linear(32, 8, bias=False))


def test_quantize_model_swaps_classes_and_creates_stubs():
    model = ff.quantize_model(make_model())
    assert type(model).__name__ == "QuantizedSequential" and isinstance(model[0], ff.nn.QuantizedLinear)
    lin = model[0]
    for name, tag in (("input_quantizer", "activation/input"), ("weight_quantizer", "parameter/weight"), ("bias_quantizer", "parameter/bias"), ("output_quantizer", "activation/output")):
        stub = getattr(lin, name)
        assert isinstance(stub, ff.nn.QuantizerStub) and tag in stub.quant_metadata and tag.split("/")[0] in stub.quant_metadata
    assert lin.weight_quantizer.quant_metadata.shape == lin.weight.shape
    assert model[1].bias_quantizer is None
    assert list(ff.nn.named_quantizers(model)) == [] and len(list(ff.nn.named_quantizers(model, skip_stubs=False))) == 7
    x = torch.randn(3, 16)
    with ff.strict_quantization(False):
        torch.testing.assert_close(model(x), make_model()(x))  # stubs are identities
    with pytest.raises(QuantizationError, match="no quantized version"):
        ff.quantize_model(torch.nn.Sequential(torch.nn.Conv2d(1, 1, 1)))
    surrogate = ff.nn.surrogate_quantized_modules(torch.nn.Sequential(torch.nn.Tanh()))
    assert torch.nn.Tanh in surrogate
    ff.quantize_model(torch.nn.Sequential(torch.nn.Tanh()), extra_conversion=surrogate)


def test_strict_quantization_of_quantized_linear():
    model = ff.quantize_model(make_model())
    with pytest.raises(QuantizationError):
        model(torch.randn(3, 16))  # strict by default: stubs do not produce QuantizedTensors


def test_override_stack_order_and_removal():
    quantizer = ff.nn.QuantizerStub()
    calls = []

    def make(tag):
        def fn(ctx, nxt, args, kwargs):
            calls.append(tag)
            return nxt(*args, **kwargs) + 1

        return fn

    h1, h2 = quantizer.register_override(make("first")), quantizer.register_override(make("second"))
    assert quantizer(torch.zeros(1)).item() == 2 and calls == ["second", "first"]
    h2.remove()
    calls.clear()
    assert quantizer(torch.zeros(1)).item() == 1 and calls == ["first"]
    with h1:
        pass
    assert quantizer(torch.zeros(1)).item() == 0 and list(quantizer.overrides) == []


def test_disable_and_enable_quantization():
    model = ff.quantize_model(make_model())
    model[0].input_quantizer = ff.nn.LinearQuantizer(4)
    model[0].input_quantizer.quantization_range = (-1.0, 1.0)
    x = torch.randn(3, 16)
    with ff.disable_quantization(model):
        assert not ff.get_strict_quantization()
        assert not isinstance(model[0].input_quantizer(x), ff.QuantizedTensor)
        with ff.enable_quantization(model):
            assert isinstance(model[0].input_quantizer(x), ff.QuantizedTensor)
    assert ff.get_strict_quantization() and isinstance(model[0].input_quantizer(x), ff.QuantizedTensor)


def test_running_minmax_matches_reference_test_values():
    """tests/range_setting/test_minmax.py:41-87: five scaled batches, per tensor and per channel."""
    g = torch.Generator().manual_seed(1)
    base = torch.randn(10, 12, generator=g)
    batches = [base * (i + 1) for i in range(5)]
    for gran, reduce_dims in ((ff.PerTensor(), None), (ff.PerChannel(0), 1)):
        quantizer = ff.nn.LinearQuantizer(8, symmetric=False, granularity=gran)
        with ff.estimate_ranges(torch.nn.ModuleList([quantizer]), ff.range_setting.running_minmax):
            for b in batches:
                out = quantizer(b)
                assert isinstance(out, ff.QuantizedTensor)
        stacked = torch.stack(batches)
        lo = stacked.min() if reduce_dims is None else stacked.amin((0, 2))
        hi = stacked.max() if reduce_dims is None else stacked.amax((0, 2))
        got_lo, got_hi = quantizer.quantization_range
        torch.testing.assert_close(got_lo.reshape(lo.shape), lo, rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(got_hi.reshape(hi.shape), hi, rtol=1e-5, atol=1e-5)
        assert list(quantizer.overrides) == []


def test_disable_quantization_estimation_passes_data_through():
    quantizer = ff.nn.LinearQuantizer(8)
    with ff.estimate_ranges(torch.nn.ModuleList([quantizer]), ff.range_setting.running_minmax, disable_quantization=True):
        out = quantizer(torch.randn(4, 4))
    assert not isinstance(out, ff.QuantizedTensor) and not quantizer.has_uninitialized_params


def test_estimator_rejects_instances_with_args_and_unsupported_quantizers():
    with pytest.raises(ValueError):
        with ff.estimate_ranges(torch.nn.ModuleList(), ff.range_setting.running_minmax(), True):
            pass
    with pytest.raises(TypeError):
        with ff.estimate_ranges(torch.nn.ModuleList([ff.nn.DynamicLinearQuantizer(8)]), ff.range_setting.running_minmax):
            pass


def test_smoothed_minmax():
    quantizer = ff.nn.LinearQuantizer(8, symmetric=False)
    a, b = torch.tensor([-1.0, 2.0]), torch.tensor([-3.0, 1.0])
    with ff.estimate_ranges(torch.nn.ModuleList([quantizer]), ff.range_setting.smoothed_minmax, gamma=0.5):
        quantizer(a), quantizer(b)
    lo, hi = quantizer.quantization_range
    torch.testing.assert_close(lo, torch.tensor([-2.0])), torch.testing.assert_close(hi, torch.tensor([1.5]))


def test_flags_setters_return_restoring_contexts():
    assert ff.get_strict_quantization()
    with ff.strict_quantization(False):
        assert not ff.get_strict_quantization()
    assert ff.get_strict_quantization()
    ff.set_strict_quantization(False)
    assert not ff.get_strict_quantization()

    @ff.flags.context(ff.strict_quantization, True)
    def inner():
        return ff.get_strict_quantization()

    assert inner() and not ff.get_strict_quantization()


def test_no_cpu_fallback_in_the_product(hip_lib):
    """With the real library active the KERNEL entry points refuse host memory loudly: nothing of the device path ever computes on
    the CPU. (Host tensors handed to the four operators / the estimator take the reference's own device-agnostic ATen chain,
    fastforward_amd/_host.py — BASELINE configs[0]; tests/test_host_route.py.) A HIP tensor without the library raises BackendError."""
    from conftest import use_backend

    with use_backend(hip_lib):
        with pytest.raises(BackendError, match="no CPU"):
            ff.ops.linear_w8a8(torch.zeros(4, 64, dtype=torch.int8), torch.zeros(8, 64, dtype=torch.int8), torch.ones(1), None, torch.ones(8), None)
        with pytest.raises(BackendError):
            ff.ops.pack_int4(torch.zeros(64, dtype=torch.int8), block=32)
        with pytest.raises(BackendError):
            ff.ops.add_rmsnorm_quantize(torch.zeros(2, 64, dtype=torch.bfloat16), None, torch.ones(64, dtype=torch.bfloat16), 1e-5, [])


def test_fuse_qdq_weights_snaps_weights_and_is_idempotent(oracle_backend):
    """Reference quantization/fuse.py:199-242 and its tests (tests/quantization/test_fuse.py): weights become their QDQ
    values, re-quantizing them is a no-op, stubbing removes the weight quantizers, tied weights must agree on the grid."""
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(32, 16), torch.nn.Linear(16, 8))
    ff.quantize_model(model)
    for layer in (model[0], model[1]):
        layer.weight_quantizer = ff.nn.LinearQuantizer(4, granularity=ff.PerChannel(0))
    with ff.estimate_ranges(model, ff.range_setting.running_minmax), ff.strict_quantization(False):
        model(torch.randn(4, 32))
    with ff.strict_quantization(False):
        want = [layer.weight_quantizer(layer.weight).dequantize().clone() for layer in (model[0], model[1])]
        before = model(torch.ones(2, 32))
    targets = ff.quantization.find_weight_quantizers(model)
    assert [t[1] for t in targets] == ["weight", "weight"] and len(targets) == 2
    ff.quantization.fuse_qdq_weights(model)
    for layer, w in zip((model[0], model[1]), want):
        assert torch.equal(layer.weight, w)
        with ff.strict_quantization(False):
            assert torch.equal(layer.weight_quantizer(layer.weight).dequantize(), w)  # idempotent
    with ff.strict_quantization(False):
        assert torch.equal(model(torch.ones(2, 32)), before)
    ff.quantization.fuse_qdq_weights(model, stub_quantizers=True)
    assert model[0].weight_quantizer.is_stub() and model[1].weight_quantizer.is_stub()
    assert ff.quantization.find_weight_quantizers(model) == []
    # tied weight, two different grids
    a, b = torch.nn.Linear(8, 8, bias=False), torch.nn.Linear(8, 8, bias=False)
    b.weight = a.weight
    tied = torch.nn.Sequential(a, b)
    ff.quantize_model(tied)
    tied[0].weight_quantizer = ff.nn.LinearQuantizer(8)
    tied[1].weight_quantizer = ff.nn.LinearQuantizer(2)
    for q in (tied[0].weight_quantizer, tied[1].weight_quantizer):
        q.quantization_range = (torch.tensor(-1.0), torch.tensor(1.0))
    with pytest.raises(ff.exceptions.QuantizationError, match="tied"):
        ff.quantization.fuse_qdq_weights(tied)


@pytest.mark.parametrize("symmetric", [True, False])
@pytest.mark.parametrize("negative_data", [True, False])
def test_uniform_search_grid(symmetric, negative_data):
    """Reference tests/range_setting/test_minerror.py:59-82."""
    from fastforward_amd.range_setting.min_error import _UniformSearchGrid

    data = torch.rand((5, 7))
    if negative_data:
        data = data - 0.5
    lo, hi = _UniformSearchGrid()(data, symmetric=symmetric, parameter_dimensionality=5, num_candidates=3)
    assert lo.shape == (3, 5) and hi.shape == (3, 5) and bool((lo < hi).all())


@pytest.mark.parametrize("symmetric", [True, False])
@pytest.mark.parametrize("negative_data", [True, False])
@pytest.mark.parametrize("gran", [ff.PerChannel(0), ff.PerChannel(-1), ff.PerTensor(), ff.PerTile((12, 16))], ids=str)
@pytest.mark.parametrize("custom_error", [False, True])
def test_mse_grid_error_decreases_with_a_finer_grid(oracle_backend, symmetric, negative_data, gran, custom_error):
    """Reference tests/range_setting/test_minerror.py:14-56, for the one-pass kernel path and for the
    candidate-by-candidate loop a custom error function takes."""
    from fastforward_amd.range_setting.min_error import mse_error

    torch.manual_seed(3)
    data = torch.randn(24, 16)
    if not negative_data:
        data = data.abs()
    quantizer = ff.nn.LinearQuantizer(8, granularity=gran, symmetric=symmetric)
    error_fn = (lambda a, b: mse_error(a, b)) if custom_error else mse_error
    errors = []
    for n in (9, 81):
        with ff.estimate_ranges(quantizer, ff.range_setting.mse_grid, num_candidates=n, error_fn=error_fn):
            quantizer(data)
        errors.append(float(torch.sum((quantizer(data).dequantize() - data) ** 2)))
    assert errors[0] >= errors[1]


# ---- GPTQ helpers (reference tests/quantization/test_gptq.py:25-140, assertions re-expressed) -----------------------------
def _smoothed_4bit(granularity, weights, symmetric=False):
    quantizer = ff.nn.LinearQuantizer(4, granularity=granularity, symmetric=symmetric)
    with ff.strict_quantization(False), ff.estimate_ranges(quantizer, ff.range_setting.smoothed_minmax):
        quantizer(weights)
    return quantizer


@pytest.mark.parametrize("granularity", [
    ff.PerTensor(), ff.PerChannel(0), ff.PerChannel(1), ff.PerChannel((0, 1)), ff.PerBlock(1, 16, 0), ff.PerBlock((0, 1), (16, 16)), ff.PerTile((16, 16)),
], ids=["tensor", "rows", "columns", "elements", "row_groups_of_16", "blocks_16x16", "tiles_16x16"])
def test_gptq_column_operator_equals_the_whole_matrix_quantizer(granularity):
    """One rule for every granularity (the [rows / tr, columns / tc] parameter grid): quantize-dequantizing column by
    column gives the matrix the quantizer itself produces."""
    from fastforward_amd.quantization.gptq import column_quantizer

    torch.manual_seed(3)
    w = torch.randn(64, 128)
    quantizer = _smoothed_4bit(granularity, w)
    with ff.strict_quantization(False):
        whole = quantizer(w).dequantize()
    by_column = torch.stack([column_quantizer(quantizer, w.shape, c)(w[:, c]) for c in range(w.shape[1])], dim=1)
    assert torch.equal(by_column, whole)


def test_gptq_partial_range_update_writes_one_group_and_clears_a_stale_offset():
    from fastforward_amd.quantization.gptq import update_partial_range

    torch.manual_seed(0)
    gran = ff.PerBlock(1, 8, 0)
    w = torch.randn(16, 32)
    quantizer = _smoothed_4bit(gran, w)
    before = quantizer.scale.detach().clone().view(16, 4)
    piece = w[:, 16:24]
    versions = (quantizer.scale._version, quantizer.offset._version)
    update_partial_range(quantizer, piece.min(-1).values, piece.max(-1).values, param_view_shape=(16, 4), param_view_index=(slice(None), 2))
    # the writes move the parameters' version counters: caches keyed on them (llama.FusedForward's weight codes) see the change
    assert quantizer.scale._version > versions[0] and quantizer.offset._version > versions[1]
    s, o = ff.quantization.affine.parameters_for_range(piece.min(-1).values, piece.max(-1).values, num_bits=4, symmetric=False, allow_one_sided=True)
    assert torch.equal(quantizer.scale.detach().view(16, 4)[:, 2], s) and torch.equal(quantizer.offset.detach().view(16, 4)[:, 2], o)
    keep = [0, 1, 3]
    assert torch.equal(quantizer.scale.detach().view(16, 4)[:, keep], before[:, keep])
    # symmetric quantizer calibrated on positive weights carries the one-sided offset; a group whose new range spans zero
    # gets offset 0 (parameters_for_range returns no offset there) and a range that covers the negative side
    pos = torch.rand(16, 32) + 0.1
    quantizer = _smoothed_4bit(gran, pos, symmetric=True)
    assert bool((quantizer.offset != 0).any())
    others = quantizer.offset.detach().clone().view(16, 4)[:, [0, 2, 3]]
    piece = pos[:, 8:16].clone()
    piece[:, 0] = -0.9
    lo, hi = piece.min(-1).values, piece.max(-1).values
    update_partial_range(quantizer, lo, hi, param_view_shape=(16, 4), param_view_index=(slice(None), 1))
    assert not bool(quantizer.offset.detach().view(16, 4)[:, 1].any())
    rmin, rmax = ff.quantization.affine.quantization_range(quantizer.scale.detach().view(16, 4)[:, 1], quantizer.offset.detach().view(16, 4)[:, 1], 4)
    assert bool((rmin <= lo).all()) and bool((rmax >= hi - 1e-6).all())
    assert torch.equal(quantizer.offset.detach().view(16, 4)[:, [0, 2, 3]], others)


# ---- freeze_parameters (reference quantization/freeze.py:74-125; assertions of tests/quantization/test_freeze.py re-expressed) ----
def _two_quantized_linears():
    torch.manual_seed(4)
    model = torch.nn.Sequential(torch.nn.Linear(16, 16), torch.nn.Linear(16, 16))
    ff.quantize_model(model)
    for layer in model:
        layer.weight_quantizer = ff.nn.LinearQuantizer(4, granularity=ff.PerChannel(0))
        layer.weight_quantizer.quantization_range = (layer.weight.detach().amin(1), layer.weight.detach().amax(1))
    return model


def test_freeze_parameters_snaps_parameters_in_place_and_retires_their_quantizers():
    from fastforward_amd.quantization.freeze import freeze_parameters

    model = _two_quantized_linears()
    with ff.strict_quantization(False):
        want = [layer.weight_quantizer(layer.weight).dequantize().detach().clone() for layer in model]
    before = [layer.weight.detach().clone() for layer in model]
    metadata = [layer.weight_quantizer.quant_metadata for layer in model]
    pointers = [layer.weight.data_ptr() for layer in model]
    with freeze_parameters(model):
        model(torch.randn(2, 16))
    for layer, w, b, meta, ptr in zip(model, want, before, metadata, pointers):
        assert torch.equal(layer.weight.detach(), w) and not torch.equal(layer.weight.detach(), b)  # quantized values, in place
        assert layer.weight.data_ptr() == ptr and isinstance(layer.weight, torch.nn.Parameter)
        assert isinstance(layer.weight_quantizer, ff.nn.QuantizerStub) and layer.weight_quantizer.quant_metadata == meta
    assert not list(ff.nn.named_quantizers(model))  # everything that was called is a stub now
    # a model without quantizers passes through untouched
    plain = torch.nn.Sequential(torch.nn.Linear(4, 4))
    with freeze_parameters(plain):
        plain(torch.randn(2, 4))


def test_freeze_parameters_can_keep_the_quantizers_and_skips_disabled_ones():
    from fastforward_amd.quantization.freeze import freeze_parameters

    model = _two_quantized_linears()
    kept = [layer.weight_quantizer for layer in model]
    with freeze_parameters(model, remove_quantizers=False):
        model(torch.randn(2, 16))
    assert [layer.weight_quantizer for layer in model] == kept
    assert all(not list(q.overrides) for q in kept)  # the freeze overrides are gone when the context exits
    with ff.strict_quantization(False):  # frozen weights sit on the grid: quantizing them again changes nothing
        for layer in model:
            assert torch.equal(layer.weight_quantizer(layer.weight).dequantize(), layer.weight)
    model = _two_quantized_linears()
    before = [layer.weight.detach().clone() for layer in model]
    kept = [layer.weight_quantizer for layer in model]
    with ff.disable_quantization(model), freeze_parameters(model):
        model(torch.randn(2, 16))
    assert all(torch.equal(layer.weight.detach(), b) for layer, b in zip(model, before))
    assert [layer.weight_quantizer for layer in model] == kept


def test_sibling_codes_the_device_decides_restated_on_the_host(oracle_backend):
    """ops.quantize_by_tile_unless_same / ops.linear_w8a8_earlier through the oracle's restatement (include/ffq.h: sibling quantizers
    while estimating): A1 runs unless scale bits and rounded offsets are the earlier quantizer's, and the linear reads the earlier
    codes exactly then. The GPU twins of these cases are tests/test_siblings_gpu.py."""
    from fastforward_amd import ops

    g = torch.Generator().manual_seed(2)
    m, n, k = 2048, 2048, 256
    x = (torch.randn(m, k, generator=g) * 2).to(torch.bfloat16)
    wq = torch.randint(-128, 128, (n, k), dtype=torch.int8, generator=g)
    sw = torch.rand(n, generator=g) * 1e-3 + 1e-4
    e_scale, e_offset = torch.tensor([0.031]), torch.tensor([2.9])
    first = ops.quantize_by_tile(x, e_scale, x.shape, 8, torch.int8, e_offset)
    for scale, offset, same in ((0.031, 3.2, True), (0.031, 2.4, False), (0.04, 2.9, False)):
        scale, offset = torch.tensor([scale]), torch.tensor([offset])
        own = ops.quantize_by_tile(x, scale, x.shape, 8, torch.int8, offset)
        maybe = ops.quantize_by_tile_unless_same(x, scale, offset, 8, e_scale, e_offset)
        if same:
            assert torch.equal(own, first)
            maybe.fill_(77)  # (unwritten)
        else:
            assert torch.equal(maybe, own)
        want = ops.linear_w8a8(own, wq, scale, offset, sw, None, None, out_dtype=torch.bfloat16)
        got = ops.linear_w8a8_earlier(maybe, (first, e_scale, e_offset), wq, scale, offset, sw, None, out_dtype=torch.bfloat16)
        assert torch.equal(got.view(torch.int16), want.view(torch.int16))
    assert ops.linear_w8a8_earlier(maybe[:64], (first[:64], e_scale, e_offset), wq, scale, offset, sw, None) is None
    assert ops.quantize_by_tile_unless_same(x.reshape(-1)[:40], scale, offset, 8, e_scale, e_offset) is None


def test_marks_of_device_decided_siblings_are_settled_or_refused():
    """_memo.RECENT's marks (``sibling_quantizers(undecided=True)``): ``settle`` selects the codes in force with the comparison the
    launch made (scale bits, rounded offsets); a mark still alive when the outermost block ends is settled there; and a reader that
    finds one of the four parameter tensors rewritten since the launch is refused — the comparison could no longer be repeated."""
    from fastforward_amd.quantization.affine._memo import RECENT

    class Holder:  # (what the memo needs of a QuantizedTensor)
        def __init__(self, raw):
            self.raw_data = raw

    first = torch.ones(32, dtype=torch.int8)
    e_scale, e_offset = torch.tensor([0.5]), torch.tensor([0.8])
    for scale, offset, same in ((0.5, 1.2, True), (0.5, None, False), (0.25, 0.8, False), (0.5, 1.5, False)):
        scale, offset = torch.tensor([scale]), None if offset is None else torch.tensor([offset])
        own = Holder(torch.full((32,), 7, dtype=torch.int8))
        with RECENT.scope(undecided=True):
            RECENT.mark_undecided(own, (first, e_scale, e_offset), scale, offset)
            assert RECENT.earlier_of(own)[0] is first
        assert RECENT.earlier_of(own) is None  # settled on the way out
        assert torch.equal(own.raw_data, first if same else torch.full((32,), 7, dtype=torch.int8))
    own = Holder(torch.zeros(32, dtype=torch.int8))
    scale = torch.tensor([0.5])
    RECENT.mark_undecided(own, (first, e_scale, e_offset), scale, None)
    scale.mul_(2.0)
    with pytest.raises(RuntimeError, match="rewritten"):
        RECENT.earlier_of(own)
    with pytest.raises(RuntimeError, match="rewritten"):
        RECENT.settle(own)
    RECENT.clear()
    # `clear` (the `finally` of `scope()`) settles EVERY marked tensor and raises the first failure behind the loop — one
    # rewritten parameter must not leave the other marked tensors with unwritten codes
    stale, fine = Holder(torch.zeros(32, dtype=torch.int8)), Holder(torch.full((32,), 7, dtype=torch.int8))
    moved, kept = torch.tensor([0.5]), torch.tensor([0.5])
    with pytest.raises(RuntimeError, match="rewritten"):
        with RECENT.scope(undecided=True):
            RECENT.mark_undecided(stale, (first, e_scale, e_offset), moved, None)
            RECENT.mark_undecided(fine, (first, e_scale, e_offset), kept, torch.tensor([1.2]))
            moved.mul_(2.0)
    assert torch.equal(fine.raw_data, first) and getattr(fine, "_ffq_earlier", None) is None  # settled although its neighbour failed
    RECENT.clear()


def test_host_route_bit_width_rule_follows_the_reference_table():
    """fastforward_amd/_host.py::can_support_bitwidth == reference _quantizer_impl.py:44-75 (mantissa + 2 for floats incl. the fp8
    variants and complex dtypes by component, integer width + 2): the widest admitted and the first refused bit width per dtype."""
    from fastforward_amd import _host

    table = {torch.bfloat16: 9, torch.float16: 12, torch.float32: 25, torch.float64: 54, torch.float8_e4m3fn: 5, torch.float8_e4m3fnuz: 5,
             torch.float8_e5m2: 4, torch.float8_e5m2fnuz: 4, torch.complex64: 25, torch.int8: 10, torch.int16: 18, torch.int32: 34, torch.uint8: 10}
    for dtype, widest in table.items():
        assert _host.can_support_bitwidth(dtype, widest) and _host.can_support_bitwidth(dtype, widest - 0.5), dtype
        assert not _host.can_support_bitwidth(dtype, widest + 0.5), dtype
