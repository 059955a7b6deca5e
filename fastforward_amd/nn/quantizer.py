"""Quantizer base classes, stubs and tags (reference: src/fastforward/nn/quantizer.py).

``Quantizer`` is an ``nn.Module`` whose ``forward`` runs the override stack around ``quantize``
(:413-416). ``QuantizerStub`` is the identity placeholder that ``QuantizedModule``s create and that
users later replace by real quantizers (:471-535). ``Tag`` / ``QuantizerMetadata`` describe what a
quantizer quantizes (``parameter/weight``, ``activation/input`` ...; :22-251).
"""

from __future__ import annotations

import collections
import copy
import logging

from types import SimpleNamespace
from typing import Any, Callable, Iterator

import torch

from fastforward_amd import forward_override as override

logger = logging.getLogger(__name__)


class Tag:
    """Interned, '/'-hierarchical symbol: ``Tag("parameter") / "weight"`` is ``Tag("parameter/weight")``."""

    _interned: dict[str, "Tag"] = {}
    _symbol: str

    def __new__(cls, symbol: "str | Tag") -> "Tag":
        if isinstance(symbol, Tag):
            return symbol
        tag = cls._interned.get(symbol)
        if tag is None:
            tag = super().__new__(cls)
            tag._symbol = symbol
            cls._interned[symbol] = tag
        return tag

    def __deepcopy__(self, memo: dict[Any, Any]) -> "Tag":
        return self

    def __copy__(self) -> "Tag":
        return self

    def __str__(self) -> str:
        return f"#{self._symbol}"

    def __repr__(self) -> str:
        return self._symbol

    def hierarchy(self) -> Iterator["Tag"]:
        """All prefixes: ``a/b/c`` yields ``a``, ``a/b``, ``a/b/c``."""
        parts = self._symbol.split("/")
        for i in range(1, len(parts) + 1):
            yield type(self)("/".join(parts[:i]))

    def __truediv__(self, rhs: "str | Tag") -> "Tag":
        if isinstance(rhs, Tag):
            rhs = rhs._symbol
        if isinstance(rhs, str):
            return type(self)(f"{self._symbol}/{rhs}")
        return NotImplemented

    def __rtruediv__(self, lhs: str) -> "Tag":
        if isinstance(lhs, str):
            return type(self)(lhs) / self
        return NotImplemented


_PARAMETER, _ACTIVATION = Tag("parameter"), Tag("activation")
default_tags = SimpleNamespace(
    parameter_quantizer=_PARAMETER,
    activation_quantizer=_ACTIVATION,
    weight_quantizer=_PARAMETER / "weight",
    bias_quantizer=_PARAMETER / "bias",
    input_quantizer=_ACTIVATION / "input",
    output_quantizer=_ACTIVATION / "output",
)


class _HasTag:
    def __init__(self, tag: Tag) -> None:
        self._tag = tag

    def __get__(self, instance: "QuantizerMetadata", owner: Any = None) -> bool:
        return self._tag in instance


class QuantizerMetadata:
    """Tags plus free-form attributes (``shape`` ...) attached to a quantizer slot (reference :138-251)."""

    parameter_quantizer = _HasTag(default_tags.parameter_quantizer)
    weight_quantizer = _HasTag(default_tags.weight_quantizer)
    bias_quantizer = _HasTag(default_tags.bias_quantizer)
    input_quantizer = _HasTag(default_tags.input_quantizer)
    activation_quantizer = _HasTag(default_tags.activation_quantizer)
    output_quantizer = _HasTag(default_tags.output_quantizer)

    def __init__(
        self,
        *tags: str | Tag,
        weight_quantizer: bool = False,
        bias_quantizer: bool = False,
        input_quantizer: bool = False,
        output_quantizer: bool = False,
        shape: tuple[int, ...] | torch.Size | None = None,
        **kwargs: Any,
    ) -> None:
        self._tags: set[Tag] = set()
        self._kwargs = dict(kwargs, shape=shape)
        for tag in tags:
            self.add_tag(tag)
        for wanted, tag in (
            (weight_quantizer, default_tags.weight_quantizer),
            (bias_quantizer, default_tags.bias_quantizer),
            (input_quantizer, default_tags.input_quantizer),
            (output_quantizer, default_tags.output_quantizer),
        ):
            if wanted:
                self.add_tag(tag)

    def __getstate__(self) -> dict[str, Any]:
        return self.__dict__.copy()

    def __setstate__(self, state: dict[str, Any]) -> None:
        self.__dict__.update(state)

    def add_tag(self, tag: Tag | str) -> None:
        self._tags.update(Tag(tag).hierarchy())

    def __repr__(self) -> str:
        extra = ", ".join(f"{k}={v}" for k, v in self._kwargs.items())
        return f"{type(self).__name__}(tags={self._tags}, {extra})"

    def __contains__(self, tag: str | Tag) -> bool:
        return Tag(tag) in self._tags

    def __getattr__(self, key: str) -> Any:
        kwargs = self.__dict__.get("_kwargs", {})
        if key in kwargs:
            return kwargs[key]
        raise AttributeError(key)

    @property
    def shape(self) -> tuple[int, ...] | torch.Size | None:
        return self._kwargs.get("shape")

    def is_extension(self, other: "QuantizerMetadata") -> bool:
        """True if `other` has all of self's tags and agrees on all of self's attributes (reference :220-243)."""
        if not self._tags.issubset(other._tags):
            return False
        for key, value in self._kwargs.items():
            if key == "shape" and value is None:
                continue
            if other._kwargs[key] != value:
                return False
        return True

    def to_stub(self) -> "QuantizerStub":
        """Deprecated in the reference (:245-251); kept because its Llama helpers use it."""
        return QuantizerStub(_metadata=self)


class Quantizer(torch.nn.Module):
    """Base class of quantizers: ``forward = overrides(quantize)``."""

    quant_metadata: QuantizerMetadata | None

    def __init__(self) -> None:
        super().__init__()
        # OrderedDict instead of dict: plain dicts cannot be weak-referenced (reference :268-270)
        super(torch.nn.Module, self).__setattr__("_quantizer_overrides", collections.OrderedDict())
        self.quant_metadata = None
        self._register_load_state_dict_pre_hook(self._materialize_before_load)

    @classmethod
    def factory(cls, *args: Any, **kwargs: Any) -> Callable[[str, "Quantizer"], "Quantizer"]:
        """``(name, current_quantizer) -> cls(*args, **kwargs)`` for bulk replacement (reference :274-298)."""

        def make(_name: str, _current: "Quantizer") -> "Quantizer":
            return cls(*args, **kwargs)

        make.__name__ = f"{cls.__name__}_factory"
        return make

    def __deepcopy__(self, memo: dict[int, Any]) -> "Quantizer":
        if id(self) in memo:
            return memo[id(self)]
        new = type(self).__new__(type(self))
        torch.nn.Module.__setstate__(new, copy.deepcopy(torch.nn.Module.__getstate__(self), memo))
        memo[id(self)] = new
        return new

    def quantize(self, data: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError

    def register_override(self, override_fn: override.OverrideFn[torch.Tensor]) -> override.OverrideHandle:
        """Push `override_fn`; it runs instead of (and may call) the current forward (reference :373-392)."""
        handle = override.OverrideHandle(self)
        self._quantizer_overrides[handle.handle_id] = override_fn
        return handle

    def remove_override(self, override_id: int) -> override.OverrideFn[torch.Tensor] | None:
        return self._quantizer_overrides.pop(override_id, None)

    @property
    def overrides(self) -> Iterator[override.OverrideFn[torch.Tensor]]:
        yield from self._quantizer_overrides.values()

    def forward(self, data: torch.Tensor) -> torch.Tensor:
        return override.apply_overrides(self, self.quantize, self._quantizer_overrides)(data)

    def extra_repr(self) -> str:
        text = super().extra_repr()
        if self._quantizer_overrides:
            text += "\n(overrides): \n"
            for i, fn in enumerate(self._quantizer_overrides.values()):
                text += f"  ({i}): {fn}\n"
        return text

    def is_stub(self) -> bool:
        return False

    def _materialize_before_load(self, state_dict: dict[str, Any], prefix: str, *_: Any) -> None:
        """Give lazily-shaped parameters the shape found in the checkpoint (reference :438-463)."""
        lazy = torch.nn.parameter.UninitializedTensorMixin
        for full_name, loaded in state_dict.items():
            param = getattr(self, full_name.removeprefix(prefix), None)
            if loaded is None or param is None:
                continue
            if isinstance(param, lazy) and not isinstance(loaded, lazy):
                with torch.no_grad():
                    param.materialize(loaded.shape)

    def reset_parameters(self) -> None:
        raise NotImplementedError(f"{type(self).__name__} does not implement 'reset_parameters'")


class QuantizerStub(Quantizer):
    """Identity quantizer carrying metadata; placeholder until a real quantizer is installed."""

    quant_metadata: QuantizerMetadata

    def __init__(
        self,
        *tags: str | Tag,
        weight_quantizer: bool = False,
        bias_quantizer: bool = False,
        input_quantizer: bool = False,
        output_quantizer: bool = False,
        shape: tuple[int, ...] | torch.Size | None = None,
        _metadata: QuantizerMetadata | None = None,
        **kwargs: Any,
    ) -> None:
        super().__init__()
        self.quant_metadata = _metadata or QuantizerMetadata(
            *tags,
            weight_quantizer=weight_quantizer,
            bias_quantizer=bias_quantizer,
            input_quantizer=input_quantizer,
            output_quantizer=output_quantizer,
            shape=shape,
            **kwargs,
        )

    def quantize(self, data: torch.Tensor) -> torch.Tensor:
        return data

    def is_stub(self) -> bool:
        return True

    def reset_parameters(self) -> None:
        pass
