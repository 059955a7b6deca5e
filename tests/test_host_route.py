"""The host-tensor route of the operators (fastforward_amd/_host.py): the reference's device-agnostic ATen chain for tensors in
HOST memory — BASELINE configs[0] ("single nn.Linear 1024 x 1024, 8-bit per-tensor weight LinearQuantizer on CPU eager") and the
reference's default ``device="cpu"`` — against the golden fixtures the reference produced (G1-G5, G6, G11, G16), with NO oracle
injected: the package's own code on its own.

What this route is not: a fallback of the device path. A HIP tensor never takes it (``ops._host_route`` looks at the device of
the tensor), and the kernel-only entry points still refuse host memory.
"""

import pytest
import torch

import fastforward_amd as ff
import parity_cases

from fastforward_amd import _native, ops
from fastforward_amd.exceptions import BackendError, QuantizationError


@pytest.fixture(autouse=True)
def _product_library_only():
    """No injection: `ops._prepare` is the product's own, the library (if loaded at all) the HIP one."""
    assert ops._prepare is ops._PRODUCT_PREPARE
    yield
    assert ops._prepare is ops._PRODUCT_PREPARE


def test_known_answer_vectors():
    parity_cases.check_known_answers("cpu")


def test_ties_clamps_nan_inf_negative_zero():
    parity_cases.check_edges("cpu")


def test_random_sweeps_all_granularities():
    parity_cases.check_sweeps("cpu")


def test_mixed_dtype_sweep():
    parity_cases.check_dtype_sweep("cpu")


def test_parameters_for_range():
    parity_cases.check_ranges("cpu")


@pytest.mark.parametrize("sync_free", [False, True])
def test_running_minmax_trajectories(sync_free):
    parity_cases.check_running_minmax("cpu", sync_free=sync_free)


def test_quantize_by_tile_backward():
    parity_cases.check_backward("cpu")


def test_smoothed_minmax_trajectories():
    parity_cases.check_smoothed_minmax("cpu")


def test_quantized_linear_through_the_generated_fallback():
    """G6: on host tensors nobody claims the linear in the dispatcher (HIP operands only), so QuantizedLinear.forward runs
    dequantize, F.linear, output quantizer — the reference's own path (_gen/fallback.py:77-112)."""
    parity_cases.check_linear("cpu")


def test_baseline_config_1_runs_as_stated():
    """nn.Linear(1024, 1024) fp32, LinearQuantizer(8) per-tensor on the weight with the reference's defaults (symmetric,
    device="cpu"), x [4, 1024]: quantize_model -> calibrate (RunningMinMax) -> forward, all on the host."""
    torch.manual_seed(1234)
    model = torch.nn.Sequential(torch.nn.Linear(1024, 1024))
    weight = model[0].weight.detach().clone()
    ff.quantize_model(model)
    model[0].weight_quantizer = ff.nn.LinearQuantizer(8)
    assert model[0].weight_quantizer.scale.device.type == "cpu" if not model[0].weight_quantizer.has_uninitialized_params else True
    x = torch.randn(4, 1024)
    previous = ff.get_strict_quantization()
    ff.set_strict_quantization(False)
    try:
        with ff.estimate_ranges(model, ff.range_setting.running_minmax):
            model(x)
        y = model(x)
        q = model[0].weight_quantizer(model[0].weight)
    finally:
        ff.set_strict_quantization(previous)
    scale = max(float(weight.min().abs()) / 128, float(weight.max().abs()) / 127)  # range.py:107-111
    assert float(model[0].weight_quantizer.scale) == pytest.approx(scale, rel=1e-6) and float(model[0].weight_quantizer.offset) == 0.0
    want = torch.clamp(torch.round(weight / model[0].weight_quantizer.scale.detach()), -128, 127)
    assert isinstance(q, ff.QuantizedTensor) and torch.equal(q.raw_data.detach(), want)
    assert torch.equal(y.detach(), torch.nn.functional.linear(x, want * model[0].weight_quantizer.scale.detach(), model[0].bias.detach()))


def test_dynamic_quantization_and_errors_on_host_tensors():
    x = torch.randn(8, 64)
    q = ff.quantization.affine.dynamic.quantize_per_channel(x, 0, 8)
    p = q.quantization_context.quantization_params
    lo, hi = x.min(1).values, x.max(1).values
    scale = ((hi - lo) / 255).clamp(torch.finfo(torch.float32).eps)
    assert torch.equal(p.scale, scale) and torch.equal(p.offset, torch.round(lo / scale + 128))
    with pytest.raises(QuantizationError, match="empty"):
        ff.quantization.affine.dynamic.quantize_per_tensor(torch.zeros(0), 8)
    with pytest.raises(ValueError, match="dimensionality"):
        ops.quantize_by_tile(x, torch.ones(1), (64,), 8, None)
    with pytest.raises(RuntimeError, match="not enough"):
        ops.quantize_by_tile(x, torch.ones(1), (8, 64), 12, torch.int8)


def test_kernel_only_entry_points_still_refuse_host_memory_and_hip_tensors_never_take_this_route(monkeypatch):
    with pytest.raises(BackendError, match="HIP device only"):
        ops.pack_int4(torch.zeros(64, dtype=torch.int8), block=32)
    with pytest.raises(BackendError):
        ops.linear_w8a8(torch.zeros(4, 64, dtype=torch.int8), torch.zeros(8, 64, dtype=torch.int8), torch.ones(1), None, torch.ones(8), None)
    # the route is chosen by the tensor's device alone: anything that is not host memory is refused by it
    class _Device:
        type = "cuda"

    class _Fake:
        device = _Device()

    assert not ops._host_route(_Fake())  # type: ignore[arg-type]
    assert ops._host_route(torch.zeros(1))
    # ... and with the oracle injected (tests only) host tensors go to the injected library instead, as before
    import conftest

    with conftest.use_backend(conftest.load_oracle()):
        assert not ops._host_route(torch.zeros(1))
    assert _native.LIBRARY_PATH.name == "libffq_hip.so"
