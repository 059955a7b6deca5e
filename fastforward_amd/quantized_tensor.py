"""``QuantizedTensor``: integer codes + the context needed to interpret them.

Host-side mirror of src/fastforward/quantized_tensor.py:290-538. A ``torch.Tensor`` subclass built
with ``as_subclass`` (:304-315): the storage IS the raw code tensor, the subclass only adds a
``QuantizationContext``. Every torch function applied to it goes through ``__torch_function__``
(:461-477):

  1. attribute-like functions listed in ``_PASSTHROUGH`` run on the raw tensor unchanged (:128-232);
  2. otherwise the quantized-operator dispatcher is asked for a kernel by op name
     (``fastforward_amd.dispatcher.dispatch``) — this is where the fused W8A8 linear is found;
  3. otherwise every quantized argument is dequantized and the float op runs, unless
     ``strict_quantization`` is on, in which case a ``QuantizationError`` is raised (:548-563).

In-place torch ops have no default implementation on a QuantizedTensor (:115-125): results of an
operation generally leave the quantization grid, so ``a += b`` silently rebinding to a float tensor
is the only defined behaviour (:493-507).
"""

from __future__ import annotations

import contextlib
import copy
import functools
import warnings

from typing import TYPE_CHECKING, Any, Callable, Generator, Sequence

import torch

from torch._C import DisableTorchFunctionSubclass
from torch._C._nn import _parse_to as _parse_to_args
from torch.utils import _pytree as pytree

from fastforward_amd import flags
from fastforward_amd.dispatcher import DispatcherPriority, Predicate, dispatch, register
from fastforward_amd.exceptions import QuantizationError

if TYPE_CHECKING:
    from fastforward_amd.quantization.function import QuantizationContext, QuantizationFunction, QuantizationParameters


# -- casts: `qt.float()`, `qt.half()`, ... dequantize first ---------------------------------------
def _dequantize_then_cast(dtype: torch.dtype, qtensor: "QuantizedTensor") -> torch.Tensor:
    return qtensor.dequantize().to(dtype)


for _name, _dtype in {
    "double": torch.double, "float": torch.float, "half": torch.half, "bfloat16": torch.bfloat16,
    "long": torch.int64, "int": torch.int32, "short": torch.int16, "char": torch.int8,
    "cdouble": torch.complex128, "cfloat": torch.complex64, "chalf": torch.complex32,
    "bool": torch.bool, "byte": torch.uint8,
}.items():
    register(_name, None, functools.partial(_dequantize_then_cast, _dtype))


# -- ops without a meaningful default on quantized data -------------------------------------------
def _forbid(func: Callable[..., Any], message: str | None = None) -> None:
    name = func.__name__
    text = message or (
        f"{name} is not implemented for QuantizedTensor. This can happen even when torch.Tensor does "
        "have an implementation as it may not generalize to the quantized representation. A user "
        f"implementation of {name} can be registered through the QuantizedTensor dispatcher system. "
        "See the documentation of `fastforward_amd.dispatcher.register` for more details."
    )

    def raiser(*args: Any, **kwargs: Any) -> Any:
        raise NotImplementedError(text)

    raiser.__name__ = f"{name}_not_implemented"
    register(name, None, raiser, DispatcherPriority.NOT_IMPLEMENTED_FALLBACK)


for _func in (
    torch.Tensor.__getitem__,
    torch.Tensor.__reversed__,
    torch.Tensor.__setitem__,
    torch.Tensor._autocast_to_full_precision,
    torch.Tensor._autocast_to_reduced_precision,
):
    _forbid(_func)

for _attr in dir(torch.Tensor):
    if _attr.endswith("_") and not _attr.endswith("__"):
        _forbid(
            getattr(torch.Tensor, _attr),
            f"The in-place operation '{_attr}' is not implemented for QuantizedTensor. A user "
            f"implementation of {_attr} can be registered through the QuantizedTensor dispatcher "
            "system. See the documentation of `fastforward_amd.dispatcher.register` for more details.",
        )


# -- functions that act on the raw tensor without dispatch or dequantization ----------------------
_PASSTHROUGH: set[Callable[..., Any]] = set()
_GETSET_DESCRIPTOR = type(torch.Tensor.grad)


def _passthrough(attr_name: str) -> None:
    attr = getattr(torch.Tensor, attr_name, None)
    if attr is None:
        return
    if isinstance(attr, _GETSET_DESCRIPTOR):
        _PASSTHROUGH.update((attr.__get__, attr.__set__))
    else:
        _PASSTHROUGH.add(attr)
    # `torch.is_floating_point(t)` is a different callable than `Tensor.is_floating_point`
    functional = getattr(torch, attr_name, None)
    if callable(functional) and functional is not attr:
        _PASSTHROUGH.add(functional)


for _attr in (
    "__cuda_array_interface__ __repr__ __setstate__ __dir__ _backward_hooks _base _cdata _grad _grad_fn "
    "_indices _is_view _nested_tensor_size _nested_tensor_strides _version as_subclass backward data_ptr "
    "dim ndim dtype get_device device grad grad_fn indices layout name ndimension nelement numel output_nr "
    "pin_memory record_stream register_hook requires_grad requires_grad_ retain_grad shape size sparse_dim "
    "sparse_mask storage storage_offset storage_type has_names names refine_names "
    "is_coalesced is_complex is_conj is_contiguous is_cpu is_cuda is_distributed is_floating_point "
    "is_inference is_ipu is_leaf is_meta is_mkldnn is_mps is_neg is_nested is_nonzero is_pinned "
    "is_same_size is_set_to is_shared is_signed is_sparse is_sparse_csr is_vulkan is_xpu"
).split():
    _passthrough(_attr)


def apply_and_reattach(
    func: Callable[[torch.Tensor], torch.Tensor], quantized: "QuantizedTensor | None" = None
) -> Any:
    """Run `func` on the raw codes and re-wrap the result with the same context (reference :247-281).

    Without `quantized` it returns a one-argument wrapper, so it also works as a decorator.
    """
    if quantized is not None:
        return quantized._quantization_context.attach(func(quantized.raw_data))

    @functools.wraps(func)
    def wrapper(q: "QuantizedTensor") -> "QuantizedTensor":
        return apply_and_reattach(func, q)

    return wrapper


def _rebuild(data: torch.Tensor, context: "QuantizationContext[Any]") -> "QuantizedTensor":
    return QuantizedTensor(data, context)


@contextlib.contextmanager
def _quiet() -> Generator[None, None, None]:
    with warnings.catch_warnings():
        warnings.filterwarnings("ignore", module=__name__)
        yield


class QuantizedTensor(torch.Tensor):
    """Raw quantized data plus the ``QuantizationContext`` that produced it."""

    _quantization_context: "QuantizationContext[Any]"

    def __new__(cls, data: torch.Tensor, *args: Any, **kwargs: Any) -> "QuantizedTensor":
        return data.as_subclass(cls)

    def __init__(self, data: torch.Tensor, quantization_context: "QuantizationContext[Any]") -> None:
        super().__init__()
        self._quantization_context = quantization_context

    # ---- device / dtype movement -------------------------------------------------------------
    def to(self, *args: Any, **kwargs: Any) -> torch.Tensor:  # type: ignore[override]
        """dtype conversions dequantize; device moves carry the parameters along (reference :330-359)."""
        if (args and isinstance(args[0], torch.Tensor)) or "other" in kwargs:
            raise ValueError(f"{type(self).__name__}.to(other: Tensor, ...) is not supported")
        device, dtype, non_blocking, memory_format = _parse_to_args(*args, **kwargs)
        if dtype is not None:
            return self.dequantize().to(device=device, dtype=dtype, non_blocking=non_blocking, memory_format=memory_format)
        with DisableTorchFunctionSubclass():
            moved = super().to(device=device, non_blocking=non_blocking, memory_format=memory_format)
        return type(self)(moved, quantization_context=self._quantization_context.to(device))

    def cuda(self, device: Any = None, non_blocking: bool = False) -> "QuantizedTensor":  # type: ignore[override]
        return self.to(device=device or "cuda", non_blocking=non_blocking)  # type: ignore[return-value]

    def cpu(self) -> "QuantizedTensor":  # type: ignore[override]
        return self.to("cpu")  # type: ignore[return-value]

    # ---- copies ------------------------------------------------------------------------------
    def __deepcopy__(self, memo: dict[Any, Any]) -> "QuantizedTensor":
        if not self.is_leaf:
            raise RuntimeError(
                "Only Tensors created explicitly by the user (graph leaves) support the deepcopy protocol at the moment"
            )
        context = copy.deepcopy(self._quantization_context, memo)
        return type(self)(copy.deepcopy(self.raw_data.detach(), memo), context)

    def __reduce_ex__(self, proto: int) -> Any:  # type: ignore[override]
        return _rebuild, (self.raw_data.detach(), self._quantization_context)

    def clone(self) -> "QuantizedTensor":  # type: ignore[override]
        """Copy of the codes and of every tensor parameter."""
        context = self._quantization_context.clone_parameters()
        with DisableTorchFunctionSubclass():
            data = super().clone()
        return context.attach(data)

    def detach(self) -> "QuantizedTensor":  # type: ignore[override]
        context = self._quantization_context.detach_parameters()
        with DisableTorchFunctionSubclass():
            data = super().detach()
        return context.attach(data)

    def contiguous(self, memory_format: Any = torch.contiguous_format) -> "QuantizedTensor":  # type: ignore[override]
        raw = self.raw_data.contiguous(memory_format=memory_format)
        context = self._quantization_context.contiguous_parameters()
        if raw is self.raw_data and context is self._quantization_context:
            return self
        return context.attach(raw)

    # ---- views of the content ----------------------------------------------------------------
    def dequantize(self) -> torch.Tensor:  # type: ignore[override]
        """Real-valued tensor (A2 for affine contexts)."""
        return self._quantization_context.quantization_fn.dequantize(self.raw_data, self.quant_args())

    @property
    def raw_data(self) -> torch.Tensor:
        """The codes as a plain tensor (no copy)."""
        return self.as_subclass(torch.Tensor)

    def int_repr(self) -> torch.Tensor:  # type: ignore[override]
        return self.raw_data

    def quant_args(self) -> "QuantizationParameters":
        return self._quantization_context.quantization_params

    @property
    def quantization_context(self) -> "QuantizationContext[Any]":
        return self._quantization_context

    @property
    def quant_func(self) -> "type[QuantizationFunction[Any]]":
        return self._quantization_context.quantization_fn

    @property
    def is_quantized(self) -> bool:  # type: ignore[override]
        """Always False — torch would otherwise treat this as one of ITS quantized tensors (reference :520-538)."""
        warnings.warn(
            "QuantizedTensor.is_quantized is used. This property evaluates to False for "
            "QuantizedTensors as it is otherwise identified as a PyTorch native Quantized tensor. "
            "The recommended approach to test for QuantizedTensor's is using isinstance."
        )
        return False

    @is_quantized.setter
    def is_quantized(self, _value: bool) -> None:
        raise AttributeError("AttributeError: can't set attribute 'is_quantized'")

    # ---- torch function protocol -------------------------------------------------------------
    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):  # type: ignore[no-untyped-def]
        kwargs = kwargs or {}
        with DisableTorchFunctionSubclass():
            if func in _PASSTHROUGH:
                return func(*args, **kwargs)
            op_name = func.__name__
            if op_name:
                kernel = dispatch(op_name, *args, **kwargs)
                if kernel:
                    return kernel(*args, **kwargs)
            return _dequantization_fallback(func, *args, **kwargs)

    def __repr__(self, **kwargs: Any) -> str:  # type: ignore[override]
        with torch._C.DisableTorchFunction(), _quiet():
            text = super().__repr__(**kwargs)
        pad = " " * (len(type(self).__name__) + 1)
        return f"{text[:-1]},\n{pad}quant_func={self.quant_func.__name__}, quant_args={self.quant_args()})"

    # in-place operators: no in-place implementation, Python falls back to `a = a <op> b`
    def _not_in_place(self, *args: Any, **kwargs: Any) -> Any:
        return NotImplemented

    __iadd__ = __isub__ = __imul__ = __imatmul__ = __itruediv__ = __ifloordiv__ = _not_in_place  # type: ignore[assignment]
    __imod__ = __ilshift__ = __irshift__ = __iand__ = __ixor__ = __ior__ = __ipow__ = _not_in_place  # type: ignore[assignment]


def _dequantize_if_quantized(obj: Any) -> Any:
    return obj.dequantize() if isinstance(obj, QuantizedTensor) else obj


def _dequantization_fallback(func: Callable[..., Any], *args: Any, **kwargs: Any) -> Any:
    if flags.get_strict_quantization():
        raise QuantizationError(
            f"{func} was called while `fastforward_amd.get_strict_quantization() == True`. Because of this, "
            "implicit dequantization is not allowed. Implicit dequantization occurs when a non-quantized "
            "operator is applied to one or more quantized tensors. This error can be resolved by changing "
            "the global config, performing the operation in a temporary config context, by explicitly "
            "dequantizing the quantized tensors before the operations, or by registering a quantized "
            "operator that handles this specific case."
        )
    with DisableTorchFunctionSubclass():
        args = pytree.tree_map(_dequantize_if_quantized, args)
        kwargs = pytree.tree_map(_dequantize_if_quantized, kwargs)
    return func(*args, **kwargs)


# Same-shape view / view_as work for any quantizer (autograd.Function may call them when an output
# aliases an input) — reference :566-598.
def _as_size(shape: tuple[Any, ...]) -> torch.Size:
    if len(shape) == 1 and isinstance(shape[0], (torch.Size, Sequence)):
        return torch.Size(shape[0])
    return torch.Size(shape)


@Predicate
def _same_shape(self: QuantizedTensor, *shape: Any) -> bool:
    return _as_size(shape) == self.shape


@Predicate
def _same_shape_as(self: QuantizedTensor, other: Any) -> bool:
    return isinstance(other, torch.Tensor) and other.shape == self.shape


@register("view", _same_shape, priority=DispatcherPriority.FALLBACK)
def _view_same_shape(self: QuantizedTensor, *shape: Any) -> torch.Tensor:
    size = _as_size(shape)
    return apply_and_reattach(lambda x: x.view(*size), self)


@register("view_as", _same_shape_as, priority=DispatcherPriority.FALLBACK)
def _view_as_same_shape(self: QuantizedTensor, other: torch.Tensor) -> torch.Tensor:
    return apply_and_reattach(lambda x: x.view_as(other), self)
