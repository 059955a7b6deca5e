"""Split-K sweep of the weight-only GEMM on the Llama-3-8B shapes at T tokens: every forced split next to the library's plan,
the vendor's bf16 GEMM on the dequantized weight and A2 + that GEMM (what the kernel replaces). Tunes wq_split()'s cost model.
usage: [FFQ_LIB=...] python tools/wq_split_sweep.py [T ...]"""
import pathlib, sys
import torch
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from fastforward_amd import ops, _native
import os
if os.environ.get("FFQ_LIB"):
    from fastforward_amd._cabi import FFQLibrary
    _native._LIB = FFQLibrary(os.environ["FFQ_LIB"])
from bench import event_time_ms

lib = _native.library()
dev = "cuda"
torch.manual_seed(0)
for T in [int(a) for a in sys.argv[1:] if a.isdigit()] or [2048]:
    print(f"== T = {T}")
    for name, n, k in (("qo", 4096, 4096), ("kv", 1024, 4096), ("gateup", 14336, 4096), ("down", 4096, 14336)):
        x = torch.randn(T, k, device=dev, dtype=torch.bfloat16)
        w = (torch.randn(n, k, device=dev) * 0.02).to(torch.bfloat16)
        lo, hi = ops.minmax_by_tile(w, (1, k))
        s8, _ = ops.parameters_for_range(lo, hi, 8, True, False, want_offset=False)
        codes8 = ops.quantize_by_tile(w, s8, (1, k), 8, torch.int8)
        w8 = ops.dequantize_by_tile(codes8, s8, (1, k), None, torch.bfloat16)
        f = 2.0 * T * n * k
        plan = int(lib.ffq_linear_wq_split(T, n, k, 0))
        row = []
        for split in (1, 2, 3, 4, 6, 8, 12, 16, 24, 32):
            if split > (k // 64) // 4 or split > 16:
                continue
            try:  # (a form declines a split it cannot run: the 256-row tiles need all units of a tile resident at once)
                ms = event_time_ms(lambda r: ops.linear_wq(x, codes8, s8, None, two_pass=False, split=split), iters=5, reps=4)
            except Exception:  # noqa: BLE001
                continue
            row.append(f"S={split}{'*' if split == plan else ''} {ms * 1e3:.1f}us {f / ms / 1e9:.0f}TF")
        ms = event_time_ms(lambda r: ops.linear_wq(x, codes8, s8, None), iters=5, reps=4)
        row.append(f"default {ms * 1e3:.1f}us {f / ms / 1e9:.0f}TF")
        previous = lib.ffq_force_generic_kernels(1)  # the 256-row tiles with their own plan (what every launch above 128 rows took up to round 5)
        try:
            ms = event_time_ms(lambda r: ops.linear_wq(x, codes8, s8, None), iters=5, reps=4)
        finally:
            lib.ffq_force_generic_kernels(previous)
        row.append(f"256-row tiles {ms * 1e3:.1f}us {f / ms / 1e9:.0f}TF")
        ms = event_time_ms(lambda r: torch.nn.functional.linear(x, w8), iters=5, reps=4)
        row.append(f"vendor {ms * 1e3:.1f}us {f / ms / 1e9:.0f}TF")
        ms = event_time_ms(lambda r: torch.nn.functional.linear(x, ops.dequantize_by_tile(codes8, s8, (1, k), None, torch.bfloat16)), iters=5, reps=4)
        row.append(f"A2+vendor {ms * 1e3:.1f}us {f / ms / 1e9:.0f}TF")
        print(f"{name:7s} N={n:5d} K={k:5d} plan {plan}: " + " | ".join(row), flush=True)
