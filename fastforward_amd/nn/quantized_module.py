"""``QuantizedModule`` and ``quantize_model`` — conversion seam #3 of the reference.

Reference: src/fastforward/nn/quantized_module.py. Conversion swaps ``module.__class__`` for the
registered quantized subclass and calls ``__init_quantization__`` (:542-564); the registry is filled
automatically when a class inherits from ``QuantizedModule`` and one ``torch.nn.Module`` type
(:95-127, :178-191). No tensor is touched by conversion.
"""

from __future__ import annotations

import logging
import textwrap
import warnings

from typing import Any, Iterator, TypeAlias, Union, cast

import torch

from fastforward_amd.exceptions import QuantizationError
from fastforward_amd.nn.quantizer import Quantizer, QuantizerMetadata, QuantizerStub

ModuleType: TypeAlias = type[torch.nn.Module]
QuantizedModuleType: TypeAlias = type["QuantizedModule"]
ModuleConversionDict: TypeAlias = dict[ModuleType, Union[QuantizedModuleType, "SkipQuantization"]]

logger = logging.getLogger(__name__)


def named_quantizers(
    module: torch.nn.Module,
    prefix: str = "",
    recurse: bool = True,
    remove_duplicate: bool = True,
    skip_stubs: bool = True,
) -> Iterator[tuple[str, Quantizer]]:
    """(name, quantizer) pairs below `module`; stubs are skipped by default (reference :38-72)."""
    children = module.named_modules(prefix="", remove_duplicate=remove_duplicate) if recurse else module.named_children()
    for name, child in children:
        if not isinstance(child, Quantizer) or (skip_stubs and isinstance(child, QuantizerStub)):
            continue
        yield (f"{prefix}.{name}" if prefix else name), child


def quantizer_state_dict(module: torch.nn.Module) -> dict[str, Any]:
    """state_dict restricted to quantizers; load with ``load_state_dict(..., strict=False)``."""
    state: dict[str, torch.Tensor] = {}
    for name, quantizer in named_quantizers(module):
        quantizer.state_dict(destination=state, prefix=f"{name}." if name else "")
    return state


_REGISTRY: dict[ModuleType, list[QuantizedModuleType]] = {}


def _register_quantized_class(cls: QuantizedModuleType) -> None:
    """Associate `cls` with the single non-quantized Module type it derives from (reference :99-127)."""
    plain = [b for b in cls.__bases__ if issubclass(b, torch.nn.Module) and not issubclass(b, QuantizedModule)]
    if not plain:
        plain = [
            b
            for b in cls.__mro__[1:]
            if issubclass(b, torch.nn.Module) and b is not torch.nn.Module and not issubclass(b, QuantizedModule)
        ]
    if len(plain) == 1:
        _REGISTRY.setdefault(plain[0], []).append(cls)


class _InitQuantizationAfterInit(type):
    def __call__(cls, *args: Any, **kwargs: Any) -> Any:
        instance = super().__call__(*args, **kwargs)
        instance.__init_quantization__()
        return instance


class QuantizedModule(torch.nn.Module, metaclass=_InitQuantizationAfterInit):
    """Base of quantized modules: all quantization set-up lives in ``__init_quantization__``.

    A quantized module defines ``QuantizerStub``s only; users replace them with concrete quantizers.
    Subclass with ``include_in_module_map=False`` to stay out of the conversion registry.
    """

    _quantizer_metadata: dict[str, QuantizerMetadata]

    def __init_quantization__(self) -> None:
        super(torch.nn.Module, self).__setattr__("_quantizer_metadata", {})

    def __init_subclass__(cls, include_in_module_map: bool = True) -> None:
        if include_in_module_map:
            _register_quantized_class(cls)

    def quantize_children(
        self: torch.nn.Module,
        extra_conversion: ModuleConversionDict | None = None,
        skip_quantized_modules: bool = False,
        *,
        ignore_global_module_map: bool = False,
    ) -> None:
        for _, child in self.named_children():
            if not isinstance(child, Quantizer):
                quantize_model(
                    child,
                    extra_conversion=extra_conversion,
                    skip_quantized_modules=skip_quantized_modules,
                    ignore_global_module_map=ignore_global_module_map,
                )

    def register_quantizer(self, name: str, quantizer: Quantizer | None, *, _register_module: bool = True) -> None:
        """Register a quantizer slot and reconcile its metadata with the module's (reference :232-289)."""
        if quantizer is not None and not isinstance(quantizer, Quantizer):
            raise TypeError(f"{quantizer} is not a Quantizer subclass")
        metadata = self.__dict__.get("_quantizer_metadata")
        if metadata is None:
            raise AttributeError(f"Cannot assign quantizer before {type(self).__name__}.__init_quantization__() call")
        if _register_module:
            self.register_module(name, quantizer)
        if quantizer is None:
            return
        known = name in metadata
        if quantizer.quant_metadata is not None:
            if not known:
                metadata[name] = quantizer.quant_metadata
            elif not metadata[name].is_extension(quantizer.quant_metadata):
                warnings.warn(
                    f"Quantizer metadata for {name} is not a consistent extension with stored quantization "
                    f"metadata for {name}. The quantizer metadata is updated to match the module. Because of "
                    "this, the quantization state may become inconsistent, for example, when the same "
                    "quantizer is shared.",
                    RuntimeWarning,
                )
        elif not known:
            metadata[name] = QuantizerMetadata()
        quantizer.quant_metadata = metadata[name]

    def __setattr__(self, name: str, value: Any) -> None:
        super().__setattr__(name, value)
        if isinstance(value, Quantizer):
            self.register_quantizer(name, value, _register_module=False)

    def named_quantizers(self, prefix: str = "", recurse: bool = True, remove_duplicate: bool = True, skip_stubs: bool = True) -> Iterator[tuple[str, Quantizer]]:
        yield from named_quantizers(self, prefix, recurse, remove_duplicate, skip_stubs=skip_stubs)

    def quantizers(self, recurse: bool = True, skip_stubs: bool = True) -> Iterator[Quantizer]:
        for _, quantizer in self.named_quantizers(recurse=recurse, skip_stubs=skip_stubs):
            yield quantizer

    def quantizer_state_dict(self) -> dict[str, Any]:
        return quantizer_state_dict(self)


class SkipQuantization:
    """Conversion-map value meaning "leave modules of this type alone"."""

    def __repr__(self) -> str:
        return "<skip quantization>"


SKIP_QUANTIZATION = SkipQuantization()


def quantized_module_map() -> dict[ModuleType, QuantizedModuleType]:
    """Module type -> most recently defined quantized counterpart (reference :567-598)."""
    mapping: dict[ModuleType, QuantizedModuleType] = {}
    for module_type, candidates in _REGISTRY.items():
        if len(candidates) > 1:
            logger.warning(
                "Multiple quantized versions of '%s.%s' exists. Defaulting to '%s.%s' which was created last",
                module_type.__module__, module_type.__qualname__, candidates[-1].__module__, candidates[-1].__qualname__,
            )
        mapping[module_type] = candidates[-1]
    return mapping


def _missing_modules(model: torch.nn.Module, module_map: ModuleConversionDict, skip_quantized_modules: bool = False) -> list[ModuleType]:
    missing = []
    for module_type in {type(m) for m in model.modules()} | {type(model)}:
        if module_type in module_map or module_type is torch.nn.Module or issubclass(module_type, Quantizer):
            continue
        if skip_quantized_modules and issubclass(module_type, QuantizedModule):
            continue
        missing.append(module_type)
    return missing


def surrogate_quantized_modules(model: torch.nn.Module, *, extra_conversion: ModuleConversionDict | None = None, ignore_global_module_map: bool = False) -> ModuleConversionDict:
    """Pass-through quantized classes for module types without a counterpart (reference :422-488)."""
    known: ModuleConversionDict = dict(extra_conversion or {})
    if not ignore_global_module_map:
        known = {**quantized_module_map(), **known}
    surrogates: ModuleConversionDict = {}
    for module_type in _missing_modules(model, known, skip_quantized_modules=True):
        surrogate = type(f"Quantized{module_type.__name__}Surrogate", (QuantizedModule, module_type), {}, include_in_module_map=False)
        surrogates[module_type] = cast(QuantizedModuleType, surrogate)
    return surrogates


def quantize_model(
    model: torch.nn.Module,
    recursive: bool = True,
    extra_conversion: ModuleConversionDict | None = None,
    skip_quantized_modules: bool = False,
    *,
    ignore_global_module_map: bool = False,
) -> torch.nn.Module:
    """Convert `model` (and children) in place to quantized counterparts (reference :491-539)."""
    module_map: ModuleConversionDict = dict(extra_conversion or {})
    if not ignore_global_module_map:
        module_map = {**quantized_module_map(), **module_map}
    missing = _missing_modules(model, module_map, skip_quantized_modules)
    if missing:
        listing = "\n".join(f"      - {m.__module__}.{m.__qualname__}" for m in missing)
        raise QuantizationError(
            textwrap.dedent(
                f"""
    Cannot quantize model because no quantized version of the following modules is known:
{listing}
    It is possible that quantized definitions of one or more of these models
    exists, but have not been imported."
    """
            ).strip()
        )
    if skip_quantized_modules and isinstance(model, QuantizedModule):
        logger.info("Skipping requantization of '%s' because skip_quantized_modules=True", type(model))
    else:
        target = module_map.get(type(model))
        if target is None:
            raise QuantizationError(f"Quantization is not supported for '{type(model)}'.")
        if not isinstance(target, SkipQuantization):
            model.__class__ = target
            cast(QuantizedModule, model).__init_quantization__()
    if isinstance(model, QuantizedModule) and recursive:
        model.quantize_children(
            extra_conversion=extra_conversion,
            skip_quantized_modules=skip_quantized_modules,
            ignore_global_module_map=ignore_global_module_map,
        )
    return model
