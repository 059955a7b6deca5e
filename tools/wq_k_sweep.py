"""The weight-only linear (int8 containers, per-channel scales) on N = 4096 output columns as a function of the contraction depth K,
next to the vendor's bf16 GEMM on the dequantized weight: microseconds per 64-deep super-step and tile round. What the down_proj gap
(K = 14336) depends on: the depth itself, the row stride's residue, or the operand footprint.
usage: [FFQ_LIB=...] python tools/wq_k_sweep.py [T ...]"""
import os, pathlib, sys
import torch
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from fastforward_amd import ops, _native
if os.environ.get("FFQ_LIB"):
    from fastforward_amd._cabi import FFQLibrary
    _native._LIB = FFQLibrary(os.environ["FFQ_LIB"])
from bench import event_time_ms

dev = "cuda"
torch.manual_seed(0)
N = 4096
t = lambda fn: event_time_ms(lambda r: fn(r), iters=6, reps=6) * 1e3  # noqa: E731
for T in [int(a) for a in sys.argv[1:] if a.isdigit()] or [16384, 4096]:
    print(f"== T = {T}, N = {N}")
    rounds = -(-(T // 256) * (N // 256) // 256)
    for k in [int(v) for v in os.environ.get("KS", "2048 4096 6144 7168 8192 10240 12288 14336 14464 16384 20480 28672").split()]:
        xs = [torch.randn(T, k, device=dev, dtype=torch.bfloat16) for _ in range(2)]
        codes = [torch.randint(-128, 128, (N, k), device=dev, dtype=torch.int8) for _ in range(3)]
        s8 = torch.rand(N, device=dev) * 1e-3 + 1e-4
        us = t(lambda r: ops.linear_wq(xs[r % 2], codes[r % 3], s8, None))
        wd = [ops.dequantize_by_tile(c, s8, (1, k), None, torch.bfloat16) for c in codes[:2]]
        vendor = t(lambda r: torch.nn.functional.linear(xs[r % 2], wd[r % 2]))
        steps = k // 64 * rounds
        f = 2.0 * T * N * k
        print(f"K={k:6d}: ours {us:8.1f}us {f / us / 1e6:5.0f}TF {us / steps * 1e3:6.0f} ns/step | vendor {vendor:8.1f}us {f / vendor / 1e6:5.0f}TF "
              f"{vendor / steps * 1e3:6.0f} ns/step | ours/vendor {us / vendor:5.3f}", flush=True)
        del xs, codes, wd
