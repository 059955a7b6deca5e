"""A3 (dynamic quantize) and A5 (parameters_for_range) timings: the one-launch / grid forms of round 5 against the composed /
one-block forms (selected with ffq_force_generic_kernels for A5; for A3 the composed form is minmax + A5 + A1 called separately).

    python3 tools/a3_time.py            # prints one line per case; HIP-event time of hipGraph-replayed launches
"""

from __future__ import annotations

import pathlib
import sys

import torch

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

import os  # noqa: E402

from fastforward_amd import _native, ops  # noqa: E402

if os.environ.get("FFQ_LIB"):  # a variant build (tools/build_variant.sh)
    from fastforward_amd._cabi import FFQLibrary

    _native._LIB = FFQLibrary(os.environ["FFQ_LIB"])
from bench import event_time_ms  # noqa: E402

dev = "cuda"


def a3() -> None:
    for label, shape, tile in (("per-token [8,2048,4096]", (8, 2048, 4096), (1, 1, 4096)), ("per-token [8,2048,14336]", (8, 2048, 14336), (1, 1, 14336)),
                               ("group-128 [14336,4096]", (14336, 4096), (1, 128)), ("per-channel [4096,4096]", (4096, 4096), (1, 4096)),
                               ("per-tensor [8,2048,4096]", (8, 2048, 4096), (8, 2048, 4096))):
        n = 1
        for s in shape:
            n *= s
        xs = [torch.randn(shape, device=dev, dtype=torch.bfloat16) for _ in range(max(2, int(6e8 // (2 * n))))]
        k = len(xs)
        for sym, one in ((False, True), (True, True)):
            ms = event_time_ms(lambda r: ops.quantize_dynamic_by_tile(xs[r % k], tile, 8, sym, one, torch.int8), iters=10, reps=12)

            def composed(r):
                lo, hi = ops.minmax_by_tile(xs[r % k], tile)
                s, o = ops.parameters_for_range(lo, hi, 8, sym, one)
                return ops.quantize_by_tile(xs[r % k], s, tile, 8, torch.int8, o)

            ms_c = event_time_ms(composed, iters=10, reps=12)
            print(f"A3 {label:28s} symmetric={sym!s:5s} one_sided={one!s:5s}: {ms * 1e3:8.1f} us = {3 * n / ms / 1e6:7.0f} GB/s (3 B/elem) = {3 * n / ms / 8e9:.3f} of 8 TB/s"
                  f" | A4 + A5 + A1 as three calls {ms_c * 1e3:8.1f} us", flush=True)
        del xs


def a5() -> None:
    lib = _native.library()
    for ntiles in (4096, 8193, 131072, 458752):
        lo = (torch.randn(ntiles, device=dev) - 0.5).to(torch.bfloat16)
        hi = (lo.float() + torch.rand(ntiles, device=dev)).to(torch.bfloat16)
        s, o = torch.empty(ntiles, device=dev), torch.empty(ntiles, device=dev)
        for sym, one in ((True, True), (False, True)):
            ms = event_time_ms(lambda r: ops.parameters_for_range(lo, hi, 8, sym, one, s, o), iters=10, reps=12)
            prev = lib.ffq_force_generic_kernels(1)
            try:
                ms_1 = event_time_ms(lambda r: ops.parameters_for_range(lo, hi, 8, sym, one, s, o), iters=10, reps=12)
            finally:
                lib.ffq_force_generic_kernels(prev)
            print(f"A5 {ntiles:7d} tiles symmetric={sym!s:5s}: {ms * 1e3:7.1f} us | one block {ms_1 * 1e3:7.1f} us", flush=True)


if __name__ == "__main__":
    if "a5" in sys.argv[1:] or len(sys.argv) == 1:
        a5()
    if "a3" in sys.argv[1:] or len(sys.argv) == 1:
        a3()
