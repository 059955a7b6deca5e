"""Tensor-level entry points of the hot path: torch tensors in, C-ABI calls out.

These functions are the bodies of the torch custom ops ``fastforward_amd::quantize_by_tile``,
``dequantize_by_tile``, ``quantize_dynamic_by_tile`` and ``quantize_by_tile_backward`` — the same
four schemas the reference registers under ``fastforward::`` (reference:
src/fastforward/quantization/_quantizer_impl.py:144-285) — plus the reduction / parameter / packing /
linear kernels the reference expresses as ATen chains. Every call is one enqueue on torch's current
HIP stream through the C ABI of ``include/ffq.h``: no host synchronisation, no hidden allocation in
the library (outputs and scratch are torch allocations), legal under hipGraph capture.


The package is split by family (round 6; one 1,700-line module before): ``static`` (the four registry operators and the row-batched
forms of A1), ``reductions`` (A4 / A5), ``packing`` (A7, GGUF, GPTQ, the grid estimator), ``gemm`` (A6 on int8 codes), ``wq`` (A6 with a
quantized weight and a plain input), ``producers`` (RMSNorm / SiLU*up / rotary / attention with A1 fused), ``registry`` (the torch operator
library and the C++ extension), ``_base`` (device check, tags, scratch). Every public name is re-exported here: ``ops.linear_wq`` etc.
"""

from __future__ import annotations

from fastforward_amd import _native  # noqa: F401  (tests and tools reach the loaded library through ops._native)
from fastforward_amd._cabi import FLAG_INF, FLAG_NAN  # noqa: F401
from fastforward_amd.ops import _base
from fastforward_amd.ops._base import (  # noqa: F401
    _DTYPES, _EXTREMA_WORDS, _PRODUCT_PREPARE, _TAGS, _TICKETS, _extrema_words, _flat, _host_route, _native_route, _ptr, _tag, _tickets, _tile_of,
    _workspace,
)
from fastforward_amd.ops.static import (  # noqa: F401
    _quantize_by_tile_backward_composite, dequantize_by_tile, quantize_by_tile, quantize_by_tile_backward, quantize_by_tile_unless_same, quantize_dynamic_by_tile, quantize_rows_batch, quantize_rows_rowsum,
)
from fastforward_amd.ops.packing import (  # noqa: F401
    _pack_gguf, gptq_block, grid_sqerror_by_tile, pack_int4, pack_q4_0_blocks, pack_q8_0_blocks, quantize_pack_int4, unpack_dequantize_int4, unpack_int4,
)
from fastforward_amd.ops.reductions import (  # noqa: F401
    _running_minmax_step, minmax_by_tile, parameters_for_range, running_minmax_quantize, running_minmax_step,
)
from fastforward_amd.ops.gemm import (  # noqa: F401
    _bmm_w8a8, _linear_w8a8, bmm_w8a8, linear_w8a8, linear_w8a8_earlier, linear_w8a8_gated, linear_w8a8_multi, linear_w8a8_takes_earlier, mlp_gate_up_w8a8, mlp_gate_up_w8a8_estimating,
)
from fastforward_amd.ops.wq import (  # noqa: F401
    _linear_wq, _wq_scratch, linear_wq, linear_wq_multi, mlp_gate_up_wq,
)
from fastforward_amd.ops.producers import (  # noqa: F401
    _fan, add_rmsnorm_quantize, attention, rope_, silu_mul_quantize,
)
from fastforward_amd.ops.registry import NATIVE_DISPATCH, TORCH_EXTENSION_PATH, _LIBRARY  # noqa: F401,E402

__all__ = [
    "quantize_by_tile",
    "dequantize_by_tile",
    "quantize_dynamic_by_tile",
    "quantize_by_tile_backward",
    "minmax_by_tile",
    "running_minmax_step",
    "parameters_for_range",
    "pack_int4",
    "unpack_int4",
    "pack_q4_0_blocks",
    "pack_q8_0_blocks",
    "quantize_pack_int4",
    "unpack_dequantize_int4",
    "gptq_block",
    "grid_sqerror_by_tile",
    "linear_w8a8",
    "linear_w8a8_multi",
    "bmm_w8a8",
    "linear_wq",
    "linear_wq_multi",
    "mlp_gate_up_w8a8",
    "add_rmsnorm_quantize",
    "silu_mul_quantize",
    "rope_",
    "quantize_rows_rowsum",
    "quantize_rows_batch",
    "attention",
    "FLAG_INF",
    "FLAG_NAN",
]


def __getattr__(name: str):
    # `_prepare` is looked up at call time by every module of the package (``_base._prepare``): the one seam oracle/inject.py substitutes
    if name == "_prepare":
        return _base._prepare
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
