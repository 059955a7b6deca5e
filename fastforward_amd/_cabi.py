"""ctypes view of the C ABI declared in ``include/ffq.h``.

Any shared library that exports that ABI can be wrapped by :class:`FFQLibrary`. The product wraps
``fastforward_amd/csrc/libffq_hip.so`` (see :mod:`fastforward_amd._native`); the test-suite wraps
the CPU oracle with the same class so both sides are driven through identical call sites.

Nothing in this module touches torch: arguments are raw addresses and integers.
"""

from __future__ import annotations

import ctypes
import enum
import os

from typing import Sequence

from fastforward_amd.exceptions import QuantizationError

FFQ_MAX_DIMS = 8
FFQ_MAX_FANOUT = 3
FFQ_ABI_VERSION = 9


class Status(enum.IntEnum):
    OK = 0
    ERR_TILE_RANK = 1
    ERR_TILE_DIVIDE = 2
    ERR_PARAM_NUMEL = 3
    ERR_PRECISION = 4
    ERR_EMPTY = 5
    ERR_DTYPE = 6
    ERR_ARG = 7
    ERR_WORKSPACE = 8
    ERR_LAUNCH = 9
    ERR_PARAM_ROWS = 10


class DType(enum.IntEnum):
    F32 = 0
    BF16 = 1
    F16 = 2
    F64 = 3
    I8 = 4
    I16 = 5
    I32 = 6
    I64 = 7
    U8 = 8


FLAG_INF = 1
FLAG_NAN = 2

# Exception type the reference raises for each failure (see the comments in include/ffq.h).
_EXCEPTIONS: dict[int, type[Exception]] = {
    Status.ERR_TILE_RANK: ValueError,
    Status.ERR_TILE_DIVIDE: ValueError,
    Status.ERR_PARAM_NUMEL: RuntimeError,
    Status.ERR_PRECISION: RuntimeError,
    Status.ERR_EMPTY: QuantizationError,
    Status.ERR_DTYPE: NotImplementedError,
    Status.ERR_ARG: ValueError,
    Status.ERR_WORKSPACE: RuntimeError,
    Status.ERR_LAUNCH: RuntimeError,
    Status.ERR_PARAM_ROWS: ValueError,
}


class Tiling(ctypes.Structure):
    """``ffq_tiling``: data shape and parameter-sharing tile."""

    _fields_ = [
        ("ndim", ctypes.c_int32),
        ("shape", ctypes.c_int64 * FFQ_MAX_DIMS),
        ("tile", ctypes.c_int64 * FFQ_MAX_DIMS),
    ]

    @classmethod
    def make(cls, shape: Sequence[int], tile: Sequence[int]) -> "Tiling":
        if len(shape) != len(tile):
            # check_tile_compatibility, quantization/tiled_tensor.py:24-29
            msg = (
                "Input dimensionality must match tile_size dimensionality got "
                f"{len(shape)} and {len(tile)}"
            )
            raise ValueError(msg)
        if len(shape) > FFQ_MAX_DIMS:
            raise NotImplementedError(f"tensors of rank > {FFQ_MAX_DIMS} are not supported")
        t = cls()
        t.ndim = len(shape)
        for i, (s, b) in enumerate(zip(shape, tile)):
            t.shape[i] = int(s)
            t.tile[i] = int(b)
        return t


class FanOut(ctypes.Structure):
    """``ffq_fanout``: the static per-tensor int8 quantizers fed by one fused producer."""

    _fields_ = [
        ("count", ctypes.c_int32),
        ("num_bits", ctypes.c_double),
        ("scale", ctypes.c_void_p * FFQ_MAX_FANOUT),
        ("offset", ctypes.c_void_p * FFQ_MAX_FANOUT),
        ("codes", ctypes.c_void_p * FFQ_MAX_FANOUT),
    ]

    @classmethod
    def make(cls, num_bits: float, scales: Sequence[int], offsets: Sequence[int | None], codes: Sequence[int]) -> "FanOut":
        if not (len(scales) == len(offsets) == len(codes)) or len(scales) > FFQ_MAX_FANOUT:
            raise ValueError(f"a fused producer feeds at most {FFQ_MAX_FANOUT} quantizers")
        f = cls()
        f.count = len(scales)
        f.num_bits = float(num_bits)
        for j, (s, o, c) in enumerate(zip(scales, offsets, codes)):
            f.scale[j], f.offset[j], f.codes[j] = s, o, c
        return f


FFQ_MAX_BATCH = 8


class RowsBatch(ctypes.Structure):
    """``ffq_rows_batch``: several row-quantized bf16 weights for one A1 launch."""

    _fields_ = [
        ("count", ctypes.c_int32),
        ("num_bits", ctypes.c_double),
        ("data", ctypes.c_void_p * FFQ_MAX_BATCH),
        ("scale", ctypes.c_void_p * FFQ_MAX_BATCH),
        ("offset", ctypes.c_void_p * FFQ_MAX_BATCH),
        ("codes", ctypes.c_void_p * FFQ_MAX_BATCH),
        ("rows", ctypes.c_int64 * FFQ_MAX_BATCH),
        ("cols", ctypes.c_int64 * FFQ_MAX_BATCH),
        ("rowsum", ctypes.c_void_p * FFQ_MAX_BATCH),
    ]


_vp = ctypes.c_void_p
_i = ctypes.c_int
_i64 = ctypes.c_int64
_d = ctypes.c_double
_sz = ctypes.c_size_t
_tp = ctypes.POINTER(Tiling)
_fp = ctypes.POINTER(FanOut)

# name -> (restype, argtypes); mirrors include/ffq.h one to one.
SIGNATURES: dict[str, tuple[object, list[object]]] = {
    "ffq_abi_version": (_i, []),
    "ffq_last_error": (ctypes.c_char_p, []),
    "ffq_backend_name": (ctypes.c_char_p, []),
    "ffq_num_tiles": (_i64, [_tp]),
    "ffq_can_support_bitwidth": (_i, [_i, _d]),
    "ffq_promote_types": (_i, [_i, _i]),
    "ffq_quantize_by_tile": (_i, [_vp, _i, _vp, _i, _i64, _vp, _i, _i64, _tp, _d, _vp, _i, _vp]),
    "ffq_dequantize_by_tile": (_i, [_vp, _i, _vp, _i, _i64, _vp, _i, _i64, _tp, _vp, _i, _vp]),
    "ffq_dequantize_result_dtype": (_i, [_i, _i, _i, _i]),
    "ffq_minmax_workspace_bytes": (_sz, [_tp, _i]),
    "ffq_minmax_by_tile": (_i, [_vp, _i, _tp, _vp, _vp, _i, _vp, _vp, _sz, _vp, _vp]),
    "ffq_running_minmax_step": (_i, [_vp, _i, _tp, _vp, _vp, _vp, _d, _i, _i, _vp, _i, _vp, _i, _vp, _sz, _vp, _vp]),
    "ffq_running_minmax_quantize": (_i, [_vp, _i, _tp, _vp, _vp, _vp, _d, _i, _i, _vp, _vp, _vp, _i, _vp, _vp]),
    "ffq_parameters_for_range_workspace_bytes": (_sz, [_i64, _i, _i]),
    "ffq_parameters_for_range": (_i, [_vp, _vp, _i, _i64, _d, _i, _i, _vp, _i, _vp, _i, _vp, _sz, _vp]),
    "ffq_quantize_dynamic_workspace_bytes": (_sz, [_tp, _i]),
    "ffq_quantize_dynamic_by_tile": (_i, [_vp, _i, _tp, _d, _i, _i, _vp, _i, _vp, _vp, _vp, _sz, _vp, _vp]),
    "ffq_pack_int4": (_i, [_vp, _i, _i64, _i64, _vp, _vp]),
    "ffq_unpack_int4": (_i, [_vp, _i64, _i64, _vp, _i, _vp]),
    "ffq_linear_w8a8_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "ffq_bmm_w8a8_workspace_bytes": (_sz, [_i64, _i64, _i64, _i64]),
    "ffq_bmm_w8a8": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _d, _i, _i64, _i64, _i64, _i64, _vp, _sz, _vp]),
    "ffq_linear_w8a8": (
        _i,
        [_vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _vp, _i, _vp, _i, _vp, _vp, _d, _i, _i64, _i64, _i64, _vp, _sz, _vp],
    ),
    "ffq_linear_w8a8_multi": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _vp, _i, _i64, _vp, _i64, _vp, _sz, _vp]),
    "ffq_quantize_by_tile_unless_same": (_i, [_vp, _i, _vp, _vp, _i64, _d, _vp, _vp, _vp, _vp]),
    "ffq_linear_w8a8_takes_earlier": (_i, [_i64, _i64, _i64]),
    "ffq_linear_w8a8_earlier": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _i64, _i64, _i64, _vp, _sz, _vp]),
    "ffq_linear_w8a8_gated": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _i64, _i64, _i64, _vp, _sz, _vp, _vp, _vp]),
    "ffq_gptq_block": (_i, [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _d, _vp]),
    "ffq_pack_gguf_blocks": (_i, [_vp, _vp, _i64, _i, _vp, _vp]),
    "ffq_quantize_pack_int4": (_i, [_vp, _i, _vp, _i64, _vp, _i64, _tp, _i64, _vp, _vp]),
    "ffq_unpack_dequantize_int4": (_i, [_vp, _vp, _i64, _vp, _i64, _tp, _i64, _vp, _i, _vp]),
    "ffq_grid_sqerror_workspace_bytes": (_sz, [_tp, _i64]),
    "ffq_grid_sqerror_by_tile": (_i, [_vp, _i, _vp, _vp, _i64, _tp, _d, _vp, _i, _vp, _sz, _vp]),
    "ffq_quantize_backward_workspace_bytes": (_sz, [_tp]),
    "ffq_quantize_by_tile_backward": (_i, [_vp, _vp, _i, _vp, _i64, _vp, _i64, _tp, _d, _vp, _vp, _vp, _vp, _sz, _vp]),
    "ffq_mlp_gate_up_wq_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "ffq_mlp_gate_up_wq": (_i, [_vp, _i, _vp, _vp, _i, _i64, _vp, _vp, _vp, _vp, _i64, _i64, _vp, _i64, _i64, _i64, _vp, _sz, _vp, _i64, _vp]),
    "ffq_mlp_gate_up_w8a8_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "ffq_mlp_gate_up_w8a8": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _d, _i64, _i64, _i64, _vp, _sz, _vp]),
    "ffq_mlp_gate_up_w8a8_estimating_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "ffq_mlp_gate_up_w8a8_estimating": (_i, [_vp] * 14 + [_i64, _i64, _i64, _vp, _sz, _vp, _vp, _vp]),
    "ffq_add_rmsnorm_quantize": (_i, [_vp, _vp, _vp, _vp, _i, _i64, _i64, _d, _vp, _fp, _vp]),
    "ffq_silu_mul_quantize": (_i, [_vp, _vp, _i, _i64, _vp, _fp, _vp]),
    "ffq_rope_inplace": (_i, [_vp, _i64, _vp, _i64, _i, _i64, _i64, _i64, _vp, _vp, _vp]),
    "ffq_quantize_rows_rowsum": (_i, [_vp, _i, _vp, _vp, _i64, _i64, _d, _vp, _vp, _vp]),
    "ffq_linear_wq_supported": (_i, [_i, _i, _i, _i64, _i64, _i64, _i64, _i64]),
    "ffq_linear_wq_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "ffq_linear_wq_multi": (_i, [_vp, _i, _i, _vp, _i, _i64, _vp, _vp, _i, _i64, _vp, _i, _i64, _vp, _i64, _vp, _sz, _vp, _i64, _vp]),
    "ffq_linear_wq_split": (_i64, [_i64, _i64, _i64, _i]),
    "ffq_linear_wq_tickets": (_i64, [_i64, _i64, _i64, _i]),
    "ffq_linear_wq_slab_bytes": (_sz, [_i64, _i64, _i64, _i, _i64]),
    "ffq_linear_wq": (_i, [_vp, _i, _vp, _i, _i64, _vp, _vp, _i64, _i64, _vp, _i, _vp, _i, _i64, _i64, _i64, _vp, _sz, _vp, _i64, _vp]),
    "ffq_force_generic_kernels": (_i, [_i]),
    "ffq_quantize_rows_batch": (_i, [ctypes.POINTER(RowsBatch), _i, _vp]),
    "ffq_attention": (_i, [_vp, _vp, _vp, _i, _i64, _i64, _i64, _i64, _i64, _d, _i, _vp, _vp, _vp, _vp, _d, _vp, _vp, _vp]),
}


class FFQLibrary:
    """A loaded implementation of the ``ffq_*`` ABI."""

    def __init__(self, path: str | os.PathLike[str]):
        self.path = os.fspath(path)
        self._dll = ctypes.CDLL(self.path)
        for name, (restype, argtypes) in SIGNATURES.items():
            try:
                fn = getattr(self._dll, name)
            except AttributeError as e:
                raise ImportError(f"{self.path} does not export {name}") from e
            fn.restype = restype
            fn.argtypes = argtypes
            setattr(self, name, fn)
        version = self.ffq_abi_version()  # type: ignore[attr-defined]
        if version != FFQ_ABI_VERSION:
            raise ImportError(f"{self.path}: ABI version {version}, expected {FFQ_ABI_VERSION}")
        self.backend_name: str = self.ffq_backend_name().decode()  # type: ignore[attr-defined]

    @property
    def is_device(self) -> bool:
        """True when the library expects device pointers."""
        return self.backend_name.startswith("hip")

    def check(self, status: int) -> None:
        """Raise the exception the reference raises for `status`."""
        if status == Status.OK:
            return
        message = self.ffq_last_error().decode(errors="replace")  # type: ignore[attr-defined]
        exc = _EXCEPTIONS.get(status, RuntimeError)
        raise exc(message or f"ffq status {status}")
