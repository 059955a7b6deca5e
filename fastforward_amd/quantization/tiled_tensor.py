"""Tile <-> row views (reference: src/fastforward/quantization/tiled_tensor.py).

The HIP kernels never materialise these views — the tile index is computed from the flat index —
but the layout they define IS the parameter order of every kernel (`ffq_tiling` in include/ffq.h),
and the composite backward op and the tests use them directly.

Row order: the data is split into a grid of tiles, tiles are numbered row-major over that grid, and
inside a tile elements keep their row-major order (reference :90-98).
"""

from __future__ import annotations

from typing import Literal, Sequence

import torch


def check_tile_compatibility(input_size: Sequence[int], tile_size: Sequence[int]) -> None:
    """Raise ValueError unless `tile_size` has the rank of `input_size` and divides it (reference :19-42)."""
    if len(input_size) != len(tile_size):
        raise ValueError(
            "Input dimensionality must match tile_size dimensionality got "
            f"{len(input_size)} and {len(tile_size)}"
        )
    bad = [i for i, (n, t) in enumerate(zip(input_size, tile_size)) if t > 0 and n % t != 0]
    if bad:
        parts = ", ".join(f"{input_size[i]} and {tile_size[i]} for dimension {i}" for i in bad)
        raise ValueError(
            f"Each dimension of tile_size must divide the corresponding input dimension. Got {parts}."
        )


def _grid_and_perm(shape: Sequence[int], tile: Sequence[int]) -> tuple[list[int], list[int]]:
    split: list[int] = []
    for n, t in zip(shape, tile):
        split += [n // t, t]
    rank = len(shape)
    perm = [2 * k for k in range(rank)] + [2 * k + 1 for k in range(rank)]
    return split, perm


def tiles_to_rows(data: torch.Tensor, tile_size: Sequence[int] | Literal["data_shape"]) -> torch.Tensor:
    """View `data` as [num_tiles, tile_numel]; a copy only when the tiles are strided."""
    if data.numel() == 0:
        return data.reshape(1, 0)
    tile = tuple(data.shape) if isinstance(tile_size, str) else tuple(tile_size)
    check_tile_compatibility(tuple(data.shape), tile)
    split, perm = _grid_and_perm(tuple(data.shape), tile)
    tile_numel = 1
    for t in tile:
        tile_numel *= t
    return data.reshape(split).permute(perm).reshape(data.numel() // tile_numel, -1)


def rows_to_tiles(
    rows: torch.Tensor, data_size: Sequence[int], tile_size: Sequence[int] | Literal["data_shape"]
) -> torch.Tensor:
    """Inverse of :func:`tiles_to_rows` (reference :101-144)."""
    if rows.numel() == 0:
        return rows.reshape(tuple(data_size))
    shape = tuple(data_size)
    tile = shape if isinstance(tile_size, str) else tuple(tile_size)
    check_tile_compatibility(shape, tile)
    numel, tile_numel = 1, 1
    for n, t in zip(shape, tile):
        numel *= n
        tile_numel *= t
    expected = (numel // tile_numel, tile_numel)
    if tuple(rows.shape) != expected:
        raise ValueError(
            f"tiled_data is expected to be of size {torch.Size(expected)} but found {rows.size()}"
        )
    split, perm = _grid_and_perm(shape, tile)
    permuted_shape = [split[p] for p in perm]
    inverse = [0] * len(perm)
    for dst, src in enumerate(perm):
        inverse[src] = dst
    return rows.reshape(permuted_shape).permute(inverse).reshape(shape)
