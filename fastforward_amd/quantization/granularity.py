"""Parameter-sharing granularities (reference: src/fastforward/quantization/granularity.py).

A granularity maps a data shape to the tile over which one (scale, offset) pair is shared. The
tile is what selects the kernel specialisation in csrc/ffq_core.hip::analyse:

    PerTensor                 -> one tile                 (parameters wave-uniform)
    PerChannel(0) on [O, I]   -> contiguous rows          (tile = flat / I)
    PerBlock(1, 128, 0)       -> contiguous runs of 128   (tile = flat / 128)
    PerChannel(-1) on [R, C]  -> strided columns          (tile = flat % C, register-resident)
    anything else             -> generic N-d tile lookup
"""

from __future__ import annotations

import abc
import logging

from typing import Any, Literal, Sequence

import torch

from fastforward_amd.quantization.tiled_tensor import check_tile_compatibility

logger = logging.getLogger(__name__)


class Granularity(abc.ABC):
    """Base class: ``tile_size(data_shape)`` returns the tile, or "data_shape" for one whole tile."""

    @abc.abstractmethod
    def tile_size(self, data_shape: torch.Size) -> torch.Size | Literal["data_shape"]:
        raise NotImplementedError

    def parameter_dimensionality(self, data_shape: torch.Size) -> int:
        """Number of (scale, offset) pairs for data of `data_shape` (reference :64-75)."""
        tile = self.tile_size(data_shape)
        if isinstance(tile, str):
            return 1
        return data_shape.numel() // tile.numel()

    def repr_args(self) -> dict[str, Any]:
        return {}

    def __repr__(self) -> str:
        inner = ", ".join(f"{k}={v}" for k, v in self.repr_args().items())
        return f"{type(self).__name__}({inner})"

    def __eq__(self, other: object) -> bool:
        if type(self) is not type(other):
            return False
        return all(getattr(self, k) == getattr(other, k) for k in getattr(type(self), "__match_args__", ()))

    def __hash__(self) -> int:
        return hash((type(self).__name__, tuple(repr(getattr(self, k)) for k in getattr(type(self), "__match_args__", ()))))


class PerTensor(Granularity):
    """One parameter pair for the whole tensor (reference :102-118)."""

    def tile_size(self, data_shape: torch.Size) -> Literal["data_shape"]:
        return "data_shape"


class PerChannel(Granularity):
    """One parameter pair per index of `channel_dim` (default 0; several dims allowed; reference :121-156)."""

    __match_args__ = ("channel_dims",)

    def __init__(self, channel_dim: int | tuple[int, ...] = 0) -> None:
        self.channel_dims = (channel_dim,) if isinstance(channel_dim, int) else tuple(channel_dim)

    def tile_size(self, data_shape: torch.Size) -> torch.Size:
        tile = list(data_shape)
        for dim in self.channel_dims:
            tile[dim] = 1
        return torch.Size(tile)

    def repr_args(self) -> dict[str, Any]:
        return {"channel": self.channel_dims[0] if len(self.channel_dims) == 1 else self.channel_dims}


def _tuple(value: int | Sequence[int]) -> tuple[int, ...]:
    return (value,) if isinstance(value, int) else tuple(value)


class PerBlock(Granularity):
    """Blocks of `block_sizes` along `block_dims`, one channel per index of `per_channel_dims` (reference :159-225)."""

    __match_args__ = ("block_dims", "block_sizes", "per_channel_dims", "strict_blocks")

    def __init__(
        self,
        block_dims: int | Sequence[int],
        block_sizes: int | Sequence[int],
        per_channel_dims: int | Sequence[int] = (),
        strict_blocks: bool = True,
    ) -> None:
        self.block_dims = _tuple(block_dims)
        self.block_sizes = _tuple(block_sizes)
        self.per_channel_dims = _tuple(per_channel_dims)
        self.strict_blocks = strict_blocks
        if len(self.block_dims) != len(self.block_sizes):
            raise ValueError("block_sizes and block_dims must be of equal length")
        overlap = [str(d) for d in self.per_channel_dims if d in self.block_dims]
        if overlap:
            logger.warning(
                "Dimensions %s are in both 'block_dims' and 'per_channel_dims'. They will be "
                "quantized as per-block following 'block_sizes'",
                ", ".join(overlap),
            )

    def tile_size(self, data_shape: torch.Size) -> torch.Size:
        tile = list(data_shape)
        for dim in self.per_channel_dims:
            tile[dim] = 1
        for dim, size in zip(self.block_dims, self.block_sizes):
            if size > data_shape[dim]:
                raise ValueError(
                    f"Can't apply per block quantization using block-size={size} over dimension "
                    f"{dim} for a tensor with shape {data_shape}. "
                )
            if self.strict_blocks and data_shape[dim] % size != 0:
                raise ValueError(
                    f"Block dim {dim} of size {size} does not divide the data dim {data_shape[dim]} "
                    "exactly. This is required because strict_blocks=True"
                )
            tile[dim] = size
        return torch.Size(tile)

    def repr_args(self) -> dict[str, Any]:
        return {
            "block_dims": self.block_dims,
            "block_sizes": self.block_sizes,
            "per_channel_dims": self.per_channel_dims,
            "strict_blocks": self.strict_blocks,
        }


class PerTile(Granularity):
    """An explicit tile shape (reference :228-262)."""

    __match_args__ = ("tile_shape",)

    def __init__(self, tile_shape: Sequence[int]) -> None:
        self.tile_shape = torch.Size(tile_shape)

    def tile_size(self, data_shape: torch.Size) -> torch.Size:
        check_tile_compatibility(data_shape, self.tile_shape)
        return self.tile_shape

    def repr_args(self) -> dict[str, Any]:
        return {"tile_shape": self.tile_shape}


def is_per_tensor(granularity: Granularity) -> bool:
    return isinstance(granularity, PerTensor)


def is_per_channel(granularity: Granularity) -> bool:
    return isinstance(granularity, PerChannel)


def is_per_block(granularity: Granularity) -> bool:
    return isinstance(granularity, PerBlock)


def granularity_from_sizes(data_size: torch.Size, tile_size: torch.Size) -> Granularity:
    """Simplest granularity whose tile for `data_size` is `tile_size` (reference :308-332)."""
    if tuple(data_size) == tuple(tile_size):
        return PerTensor()
    dims = range(len(data_size))
    whole_or_one = all(tile_size[i] in (1, data_size[i]) for i in dims)
    channel_dims = tuple(i for i in dims if tile_size[i] == 1 and data_size[i] > 1)
    if whole_or_one:
        return PerChannel(channel_dims)
    block_dims = tuple(i for i in dims if tile_size[i] not in (1, data_size[i]))
    block_sizes = tuple(tile_size[i] for i in block_dims)
    strict = all(data_size[i] % tile_size[i] == 0 for i in dims)
    return PerBlock(block_dims, block_sizes, channel_dims, strict_blocks=strict)
