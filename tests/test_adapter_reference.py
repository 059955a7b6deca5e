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
