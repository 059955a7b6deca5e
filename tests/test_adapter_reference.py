"""adapter.install() against the real reference (only where /root/reference is importable).

Runs on CPU: the hooks are registered for the "cpu" device type with the oracle standing in for the
HIP library, which exercises the exact registration path a maintainer would use for "cuda".
"""

import os
import sys

import pytest
import torch

REFERENCE = "/root/reference/src"
SHIM = "/tmp/ffshim"

pytestmark = pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="the reference only exists in the build container")


def test_reference_ops_run_on_this_backend(oracle_backend, tmp_path):
    shim = tmp_path / "optree"
    shim.mkdir()
    (shim / "__init__.py").write_text("from torch.utils._pytree import tree_map, tree_flatten, tree_unflatten, tree_leaves\n")
    sys.path[:0] = [REFERENCE, str(tmp_path)]
    try:
        import fastforward as ff_ref

        from fastforward_amd import adapter

        g = torch.Generator().manual_seed(0)
        x = torch.randn(16, 32, generator=g)
        quantizer = ff_ref.nn.LinearQuantizer(4, symmetric=False, granularity=ff_ref.PerChannel(0))
        quantizer.quantization_range = (x.min(1).values, x.max(1).values)
        before = quantizer(x)
        dyn_before = ff_ref.quantization.affine.dynamic.quantize_per_tensor(x, 8)
        from unittest import mock

        from fastforward_amd import ops

        with mock.patch.object(ops, "quantize_by_tile", wraps=ops.quantize_by_tile) as spy:
            attached = adapter.install(device_types=("cpu",), register_linear=False)
            assert len(attached) == 4
            after = quantizer(x)  # now served by this package's op body through the C ABI
            assert spy.call_count == 1
        dyn_after = ff_ref.quantization.affine.dynamic.quantize_per_tensor(x, 8)
        assert isinstance(after, ff_ref.QuantizedTensor)
        assert torch.equal(after.raw_data, before.raw_data) and torch.equal(after.dequantize(), before.dequantize())
        assert torch.equal(dyn_after.raw_data, dyn_before.raw_data)
    finally:
        sys.path.remove(REFERENCE)
        sys.path.remove(str(tmp_path))


def test_dispatcher_hooks_serve_the_reference_operators(oracle_backend, tmp_path):
    """install() also registers linear (W8A8 and weight-only), mm, matmul and bmm in the REFERENCE's dispatcher
    (src/fastforward/dispatcher.py:233-265; fallbacks _gen/fallback.py:77-112, 699-798): the reference's own
    ff.nn.functional.* then run this package's kernels, and agree with the reference's float fallback within its own
    half-precision tolerance (tests/quantization/test_tiled_affine.py:43-55)."""
    shim = tmp_path / "optree"
    shim.mkdir()
    (shim / "__init__.py").write_text("from torch.utils._pytree import tree_map, tree_flatten, tree_unflatten, tree_leaves\n")
    sys.path[:0] = [REFERENCE, str(tmp_path)]
    policy = None
    try:
        import fastforward as ff_ref

        from fastforward_amd import adapter, fused_linear

        g = torch.Generator().manual_seed(1)
        x = torch.randn(12, 128, generator=g)
        w = torch.randn(48, 128, generator=g) * 0.1
        qx = ff_ref.quantization.affine.quantize_per_tensor(x, 0.03, 2.0, 8)
        qw = ff_ref.quantization.affine.quantize_per_channel(w, torch.full((48,), 0.002), None, 0, 8)
        qwt = ff_ref.quantization.affine.quantize_per_tensor(w.t().contiguous(), 0.002, None, 8)
        xb = torch.randn(3, 12, 64, generator=g)
        wb = torch.randn(3, 64, 20, generator=g)
        qxb = ff_ref.quantization.affine.quantize_per_tensor(xb, 0.03, None, 8)
        qwb = ff_ref.quantization.affine.quantize_per_tensor(wb, 0.03, None, 8)
        x16 = torch.randn(5, 128, generator=g).to(torch.bfloat16)
        w16 = (torch.randn(32, 128, generator=g) * 0.1).to(torch.bfloat16)
        qw16 = ff_ref.quantization.affine.quantize_per_channel(w16, torch.full((32,), 0.003), None, 0, 8)
        F = ff_ref.nn.functional
        with ff_ref.strict_quantization(False):
            before = [F.linear(qx, qw), F.mm(qx, qwt), F.matmul(qx, qwt), F.bmm(qxb, qwb), F.linear(x16, qw16)]
            attached = adapter.install(device_types=("cpu",), register_linear=True)
            assert {"dispatcher:linear", "dispatcher:linear(weight-only)", "dispatcher:mm", "dispatcher:matmul", "dispatcher:bmm"} <= set(attached)
            assert ff_ref.dispatcher.dispatch("linear", input=qx, weight=qw) == adapter.REFERENCE_KERNELS.linear
            with fused_linear.weight_only_kernel(False):  # the A/B arm: nobody claims a weight-only linear, the reference's path runs
                assert ff_ref.dispatcher.dispatch("linear", input=x16, weight=qw16) is None
            # 5 tokens: the weight-code GEMM takes every token count since round 4 (no threshold)
            assert ff_ref.dispatcher.dispatch("linear", input=x16, weight=qw16) == adapter.REFERENCE_KERNELS.weight_only_linear
            assert ff_ref.dispatcher.dispatch("mm", input=qx, mat2=qwt) == adapter.REFERENCE_KERNELS.mm
            assert ff_ref.dispatcher.dispatch("matmul", input=qx, other=qwt) == adapter.REFERENCE_KERNELS.mm
            assert ff_ref.dispatcher.dispatch("bmm", input=qxb, mat2=qwb) == adapter.REFERENCE_KERNELS.bmm
            assert ff_ref.dispatcher.dispatch("linear", input=x, weight=w) is None  # nothing quantized: not ours
            after = [F.linear(qx, qw), F.mm(qx, qwt), F.matmul(qx, qwt), F.bmm(qxb, qwb), F.linear(x16, qw16)]
            # the reference's own LinearQuantizer as output quantizer (fallback.py:110-111): fused into the GEMM's epilogue by the
            # same code path this package's dispatcher uses; the result is the reference's QuantizedTensor with the same codes
            oq = ff_ref.nn.LinearQuantizer(8, symmetric=False, quantized_dtype=torch.int8)
            oq.quantization_range = (float(before[0].min()), float(before[0].max()))
            with torch.no_grad():
                fused = F.linear(qx, qw, output_quantizer=oq)
                two_pass = oq(after[0])
            assert isinstance(fused, ff_ref.QuantizedTensor) and fused.raw_data.dtype == torch.int8
            assert torch.equal(fused.raw_data, two_pass.raw_data) and torch.equal(fused.dequantize(), two_pass.dequantize())
        for a, b in zip(after, before):
            assert a.shape == b.shape and a.dtype == b.dtype
            torch.testing.assert_close(a.float(), b.float(), atol=1e-1, rtol=1.3e-2)
    finally:
        if policy is not None:
            policy.__exit__(None, None, None)
        sys.path.remove(REFERENCE)
        sys.path.remove(str(tmp_path))


def test_install_attaches_the_cpp_kernels_under_the_reference_operator_names(oracle_backend, tmp_path):
    """adapter.install() for the device key: fastforward::* get the C++ dispatch-key kernels of libffq_torch.so (dispatcher -> C++ ->
    C ABI), not Python functions — checked on the dispatch table (no GPU here); the host route of the reference's ops is untouched."""
    shim = tmp_path / "optree"
    shim.mkdir()
    (shim / "__init__.py").write_text("from torch.utils._pytree import tree_map, tree_flatten, tree_unflatten, tree_leaves\n")
    sys.path[:0] = [REFERENCE, str(tmp_path)]
    try:
        import fastforward as ff_ref

        from fastforward_amd import adapter, ops

        assert ops.NATIVE_DISPATCH
        x = torch.randn(8, 16)
        before = ff_ref.quantization.affine.quantize_per_tensor(x, 0.05, 1.0, 8)
        attached = adapter.install(register_linear=False)
        assert len(attached) == 4
        for name in ("quantize_by_tile", "dequantize_by_tile", "quantize_dynamic_by_tile", "quantize_by_tile_backward"):
            assert "CUDA: registered at ffq_torch.cpp" in torch._C._dispatch_dump(f"fastforward::{name}"), name
        # host tensors never reach the device kernels (the reference's eager chain — or, after the tests above, the oracle-backed host hook)
        after = ff_ref.quantization.affine.quantize_per_tensor(x, 0.05, 1.0, 8)
        assert torch.equal(after.raw_data, before.raw_data)
        adapter.install(register_linear=False)  # idempotent
    finally:
        sys.path.remove(REFERENCE)
        sys.path.remove(str(tmp_path))
