"""The weight-only linear at T token rows with the LIBRARY'S OWN PLAN (whatever form that is: skinny, 128-column tiles, 256-row tiles)
next to the 256-row tiles (ffq_force_generic_kernels), the vendor's bf16 GEMM on the dequantized weight and A2 + that GEMM (the
reference's route, fallback.py:86-112), int8 containers and packed W4 g128, Llama-3-8B shapes.
usage: [FFQ_LIB=...] python tools/wq_rows_sweep.py [T ...]"""
import os, pathlib, sys
import torch
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from fastforward_amd import ops, _native
if os.environ.get("FFQ_LIB"):
    from fastforward_amd._cabi import FFQLibrary
    _native._LIB = FFQLibrary(os.environ["FFQ_LIB"])
from bench import event_time_ms

lib = _native.library()
dev = "cuda"
torch.manual_seed(0)
SHAPES = (("qo", 4096, 4096), ("kv", 1024, 4096), ("gateup", 14336, 4096), ("down", 4096, 14336))
t = lambda fn: event_time_ms(lambda r: fn(r), iters=8, reps=8) * 1e3  # noqa: E731
for T in [int(a) for a in sys.argv[1:] if a.isdigit()] or [17, 32, 64, 128, 129, 256, 300, 384, 512]:
    print(f"== T = {T}")
    for name, n, k in SHAPES:
        copies = max(2, int(6e8 // (n * k)))  # rotate over several weights: the codes come from HBM, not from the Infinity Cache
        xs = [torch.randn(T, k, device=dev, dtype=torch.bfloat16) for _ in range(2)]
        codes = [torch.randint(-128, 128, (n, k), device=dev, dtype=torch.int8) for _ in range(copies)]
        s8 = torch.rand(n, device=dev) * 1e-3 + 1e-4
        plan = int(lib.ffq_linear_wq_split(T, n, k, 0))
        us = t(lambda r: ops.linear_wq(xs[r % 2], codes[r % copies], s8, None))
        previous = lib.ffq_force_generic_kernels(1)
        try:
            tiles = t(lambda r: ops.linear_wq(xs[r % 2], codes[r % copies], s8, None))
        finally:
            lib.ffq_force_generic_kernels(previous)
        w4 = [torch.randint(-8, 8, (n, k), device=dev, dtype=torch.int8) for _ in range(2)]
        packed = [ops.pack_int4(w, block=128) for w in w4]
        s4 = torch.rand(n * (k // 128), device=dev) * 1e-2 + 1e-3
        us4 = t(lambda r: ops.linear_wq(xs[r % 2], packed[r % 2], s4, None, group=128, pack_block=128))
        wd = [ops.dequantize_by_tile(c, s8, (1, k), None, torch.bfloat16) for c in codes[:2]]
        vendor = t(lambda r: torch.nn.functional.linear(xs[r % 2], wd[r % 2]))
        a2v = t(lambda r: torch.nn.functional.linear(xs[r % 2], ops.dequantize_by_tile(codes[r % copies], s8, (1, k), None, torch.bfloat16)))
        f = 2.0 * T * n * k
        print(f"{name:7s} N={n:5d} K={k:5d} plan S={plan}: {us:6.1f}us {f / us / 1e6:5.0f}TF | 256-row tiles {tiles:6.1f}us | vendor GEMM alone {vendor:6.1f}us | A2+vendor {a2v:6.1f}us | "
              f"W4 g128 packed {us4:6.1f}us | vs vendor {vendor / us:4.2f}x | vs A2+vendor {a2v / us:4.2f}x", flush=True)
        del codes, wd, w4, packed
