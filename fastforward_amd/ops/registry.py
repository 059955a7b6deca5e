"""The torch operator library ``fastforward_amd::*``: the reference's four schemas (quantization/_quantizer_impl.py:127-134) plus the hot
entry points behind the dispatcher and the range estimator, their Python (CompositeExplicitAutograd) and Meta implementations, and the
C++ dispatch-key kernels of csrc/libffq_torch.so when that extension loads."""

from __future__ import annotations

import torch

from fastforward_amd import _native
from fastforward_amd.ops import _base
from fastforward_amd.ops.static import dequantize_by_tile, quantize_by_tile, quantize_by_tile_backward, quantize_dynamic_by_tile
from fastforward_amd.ops.reductions import _running_minmax_step
from fastforward_amd.ops.gemm import _bmm_w8a8, _linear_w8a8
from fastforward_amd.ops.wq import _linear_wq


# ---------------------------------------------------------------------------------------------
# torch custom-op registration: same four schemas as the reference's `fastforward::` ops, in this
# package's own namespace (defining `fastforward::*` twice in one process is an error).
# ---------------------------------------------------------------------------------------------
_LIBRARY = torch.library.Library("fastforward_amd", "DEF")
_LIBRARY.define(
    "quantize_by_tile(Tensor data, Tensor scale, SymInt[] tile_size, float num_bits, "
    "ScalarType? output_dtype, Tensor? offset=None) -> Tensor"
)
_LIBRARY.define(
    "dequantize_by_tile(Tensor data, Tensor scale, SymInt[] tile_size, Tensor? offset=None, "
    "ScalarType? output_dtype=None) -> Tensor"
)
_LIBRARY.define(
    "quantize_dynamic_by_tile(Tensor data, SymInt[] tile_size, float num_bits, bool symmetric, "
    "bool allow_one_sided, ScalarType? output_dtype) -> (Tensor, Tensor, Tensor)"
)
_LIBRARY.define(
    "quantize_by_tile_backward(Tensor data, Tensor output_grad, Tensor scale, SymInt[] tile_size, "
    "float num_bits, Tensor? offset=None) -> Tensor[]"
)
# ... and the hot entry points behind the dispatcher and the range estimator as operators of the same library (round 5), so that
# they too have C++ device kernels (csrc/ffq_torch.cpp); the Python bodies stay registered and serve whenever the extension is absent
_LIBRARY.define(
    "running_minmax_step(Tensor data, SymInt[] tile_size, Tensor(a!) running_min, Tensor(b!) running_max, Tensor(c!)? status_flags, "
    "float num_bits, bool symmetric, bool allow_one_sided, Tensor(d!) scale_out, Tensor(e!)? offset_out) -> ()"
)
_LIBRARY.define(
    "linear_w8a8(Tensor x_codes, Tensor w_codes, Tensor x_scale, Tensor? x_offset, Tensor w_scale, Tensor? w_offset, Tensor? bias, "
    "ScalarType out_dtype, Tensor? out_scale, Tensor? out_offset, float out_num_bits, Tensor? w_rowsum, ScalarType? requant_from) -> Tensor"
)
_LIBRARY.define(
    "bmm_w8a8(Tensor x_codes, Tensor w_codes, Tensor x_scale, Tensor? x_offset, Tensor w_scale, Tensor? w_offset, ScalarType out_dtype, "
    "Tensor? out_scale, Tensor? out_offset, float out_num_bits, ScalarType? requant_from) -> Tensor"
)
_LIBRARY.define(
    "linear_wq(Tensor x, Tensor w_codes, Tensor w_scale, Tensor? w_offset, int group, Tensor? bias, ScalarType out_dtype, int pack_block, "
    "int two_pass, int split) -> Tensor"
)
_LIBRARY.impl("running_minmax_step", _running_minmax_step, "CompositeExplicitAutograd")
_LIBRARY.impl("linear_w8a8", _linear_w8a8, "CompositeExplicitAutograd")
_LIBRARY.impl("bmm_w8a8", _bmm_w8a8, "CompositeExplicitAutograd")
_LIBRARY.impl("linear_wq", _linear_wq, "CompositeExplicitAutograd")
_LIBRARY.impl("quantize_by_tile", quantize_by_tile, "CompositeExplicitAutograd")
_LIBRARY.impl("dequantize_by_tile", dequantize_by_tile, "CompositeExplicitAutograd")
_LIBRARY.impl("quantize_dynamic_by_tile", quantize_dynamic_by_tile, "CompositeExplicitAutograd")
_LIBRARY.impl("quantize_by_tile_backward", quantize_by_tile_backward, "CompositeExplicitAutograd")


# Fake / Meta implementations of all four ops (reference _quantizer_impl.py:288-339): shapes, dtypes and devices of the
# real outputs without touching data, so the ops trace under FakeTensor / torch.compile / torch.export.
def _float_result(*dtypes: torch.dtype) -> torch.dtype:
    out = dtypes[0]
    for d in dtypes[1:]:
        out = torch.promote_types(out, d)
    return out if out.is_floating_point else torch.float32


def _meta_quantize_by_tile(data, scale, tile_size, num_bits, output_dtype, offset=None):  # type: ignore[no-untyped-def]
    if output_dtype is None:
        output_dtype = _float_result(data.dtype, scale.dtype, (offset if offset is not None else scale).dtype)
    return torch.empty(data.shape, dtype=output_dtype, device=data.device)


def _meta_dequantize_by_tile(data, scale, tile_size, offset=None, output_dtype=None):  # type: ignore[no-untyped-def]
    if output_dtype is None:
        output_dtype = _float_result(data.dtype, scale.dtype, *(() if offset is None else (offset.dtype,)))
    return torch.empty(data.shape, dtype=output_dtype, device=data.device)


def _meta_quantize_dynamic_by_tile(data, tile_size, num_bits, symmetric, allow_one_sided, output_dtype):  # type: ignore[no-untyped-def]
    tile = 1
    for extent in tile_size:
        tile *= extent
    ntiles = data.numel() // tile if tile else 0
    if output_dtype is None:
        output_dtype = data.dtype if data.dtype in (torch.float32, torch.float64) else torch.float32
    params = lambda: torch.empty(ntiles, dtype=torch.float32, device=data.device)  # noqa: E731
    return torch.empty(data.shape, dtype=output_dtype, device=data.device), params(), params()


def _meta_quantize_by_tile_backward(data, output_grad, scale, tile_size, num_bits, offset=None):  # type: ignore[no-untyped-def]
    doffset = scale.new_empty(0) if offset is None else torch.empty_like(scale)
    return [torch.empty(data.shape, dtype=data.dtype, device=data.device), torch.empty_like(scale), doffset]


def _meta_linear_w8a8(x_codes, w_codes, x_scale, x_offset, w_scale, w_offset, bias, out_dtype, out_scale, out_offset, out_num_bits, w_rowsum, requant_from):  # type: ignore[no-untyped-def]
    return torch.empty((*x_codes.shape[:-1], w_codes.shape[0]), dtype=out_dtype, device=x_codes.device)


def _meta_bmm_w8a8(x_codes, w_codes, x_scale, x_offset, w_scale, w_offset, out_dtype, out_scale, out_offset, out_num_bits, requant_from):  # type: ignore[no-untyped-def]
    return torch.empty((x_codes.shape[0], x_codes.shape[1], w_codes.shape[1]), dtype=out_dtype, device=x_codes.device)


def _meta_linear_wq(x, w_codes, w_scale, w_offset, group, bias, out_dtype, pack_block, two_pass, split):  # type: ignore[no-untyped-def]
    n = w_codes.numel() * 2 // x.shape[-1] if pack_block > 0 else w_codes.shape[0]
    return torch.empty((*x.shape[:-1], n), dtype=out_dtype, device=x.device)


_LIBRARY.impl("running_minmax_step", lambda *args: None, "Meta")
_LIBRARY.impl("linear_w8a8", _meta_linear_w8a8, "Meta")
_LIBRARY.impl("bmm_w8a8", _meta_bmm_w8a8, "Meta")
_LIBRARY.impl("linear_wq", _meta_linear_wq, "Meta")
_LIBRARY.impl("quantize_by_tile", _meta_quantize_by_tile, "Meta")
_LIBRARY.impl("dequantize_by_tile", _meta_dequantize_by_tile, "Meta")
_LIBRARY.impl("quantize_dynamic_by_tile", _meta_quantize_dynamic_by_tile, "Meta")
_LIBRARY.impl("quantize_by_tile_backward", _meta_quantize_by_tile_backward, "Meta")


# Every operator above also has a C++ device kernel (csrc/ffq_torch.cpp -> csrc/libffq_torch.so, registered for the HIP
# dispatch key): torch.ops.fastforward_amd.* on a HIP tensor then runs dispatcher -> C++ -> the C ABI without entering the
# interpreter. Same library, same kernels, same results as the Python implementations above, which stay registered (and serve
# when the extension is absent or FFQ_NO_TORCH_EXT=1 — they are the HIP path too).
TORCH_EXTENSION_PATH = _native.LIBRARY_PATH.with_name("libffq_torch.so")


def _load_torch_extension() -> bool:
    import os
    import warnings

    if os.environ.get("FFQ_NO_TORCH_EXT") == "1" or not TORCH_EXTENSION_PATH.exists() or not _native.is_available():
        return False
    try:
        torch.ops.load_library(str(TORCH_EXTENSION_PATH))
    except OSError as e:  # built against another PyTorch
        warnings.warn(f"fastforward_amd: cannot load {TORCH_EXTENSION_PATH} ({e}); the operators run through the Python implementations")
        return False
    return True


NATIVE_DISPATCH: bool = _load_torch_extension()
_base.NATIVE_DISPATCH = NATIVE_DISPATCH
