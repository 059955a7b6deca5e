"""Attach the HIP kernels to a real ``fastforward`` installation (when it is importable).

The reference defines its four hot-path ops with ``torch.library.custom_op`` (reference
src/fastforward/quantization/_quantizer_impl.py:127-134); each resulting ``CustomOpDef`` exposes
``register_kernel(device_type)``. ``install()`` registers this package's C-ABI-backed functions as the
``"cuda"`` (= HIP on ROCm) kernels of those ops, and registers the quantized linear / mm / matmul / bmm kernels in the
reference's own operator dispatcher (reference src/fastforward/dispatcher.py:233-265). After that an
unmodified FastForward program runs its fake-quantization hot path on the MI355X kernels:

    import fastforward as ff, fastforward_amd.adapter
    fastforward_amd.adapter.install()
    ff.quantize_model(model) ...           # the usual recipe, unchanged

The dispatcher hooks are the SAME implementation this package registers in its own dispatcher
(:class:`fastforward_amd.fused_linear.DispatcherKernels`), instantiated against the reference's types: one set of
predicates, one set of launches (zero weight-offset buffers decided on the device, the output quantizer inside the GEMM's
epilogue, the hand-written weight-only GEMM) — a reference user and a user of this package's own surface get the same
backend.

Nothing here is needed (or importable) on a machine without the reference; this package's own
surface (``fastforward_amd.nn`` etc.) calls the same kernels directly.
"""

from __future__ import annotations

from typing import Sequence

from fastforward_amd import ops
from fastforward_amd.fused_linear import DispatcherKernels, Surface


def reference_surface() -> Surface:
    import fastforward as ff

    from fastforward.nn.linear_quantizer import LinearQuantizer
    from fastforward.quantization.affine import AffineQuantizationFunction, StaticAffineQuantParams
    from fastforward.quantization.function import QuantizationContext

    return Surface(ff.QuantizedTensor, AffineQuantizationFunction, StaticAffineQuantParams, LinearQuantizer, QuantizationContext,
                   ff.exceptions.QuantizationError, ff.get_export_mode)


# the dispatcher kernels written against the reference's QuantizedTensor / LinearQuantizer / QuantizationError
REFERENCE_KERNELS = DispatcherKernels(reference_surface)


def install(device_types: Sequence[str] = ("cuda",), register_linear: bool = True) -> list[str]:
    """Returns the names of the reference hooks that were attached."""
    import fastforward as ff

    from fastforward.quantization import _quantizer_impl as impl

    attached = []
    table = {
        "quantize_by_tile_impl": ops.quantize_by_tile,
        "dequantize_by_tile_impl": ops.dequantize_by_tile,
        "quantize_dynamic_by_tile_impl": ops.quantize_dynamic_by_tile,
        "quant_dequant_by_tile_grad_impl": ops.quantize_by_tile_backward,
    }
    native = False
    if ops.NATIVE_DISPATCH and tuple(device_types) == ("cuda",):
        # the C++ dispatch-key kernels of libffq_torch.so under the reference's operator names: fastforward::* on a HIP tensor
        # then runs dispatcher -> C++ -> C ABI (csrc/ffq_torch.cpp::ffq_torch_install_reference_kernels), no Python in between
        import ctypes

        native = ctypes.CDLL(str(ops.TORCH_EXTENSION_PATH)).ffq_torch_install_reference_kernels() >= 0
    for attr, fn in table.items():
        op_def = getattr(impl, attr)
        if not native:  # Python functions (ctypes -> C ABI) as the ops' device kernels
            for device_type in device_types:
                op_def.register_kernel(device_type)(fn)
        attached.append(f"fastforward::{op_def._opname if hasattr(op_def, '_opname') else attr}")
    if register_linear:
        hooks = REFERENCE_KERNELS.register_all(lambda name, predicate, kernel: ff.dispatcher.register(name, predicate, kernel), ff.dispatcher.Predicate)
        attached += [f"dispatcher:{name}" for name in hooks]
    return attached
