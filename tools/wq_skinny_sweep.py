"""The skinny form of the weight-only GEMM (csrc/ffq_wskinny.hip, M <= 128) on the Llama-3-8B shapes at T tokens: the library's
plan and forced K splits, the 256-row-tile kernel (ffq_force_generic_kernels), the vendor's bf16 GEMM on the dequantized weight and
A2 + that GEMM (what the kernel replaces), and the packed-nibble (W4 group-128) form. The codes are the algorithmic bytes of a
GEMV-shaped call: GB/s = code bytes / time, against the 8 TB/s HBM peak.
usage: [FFQ_LIB=...] python tools/wq_skinny_sweep.py [T ...]"""
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


def t(fn):
    return event_time_ms(lambda r: fn(r), iters=8, reps=8) * 1e3


for T in [int(a) for a in sys.argv[1:] if a.isdigit()] or [1, 16, 64, 128]:
    print(f"== T = {T}")
    for name, n, k in SHAPES:
        # rotate over several weights so that the codes come from HBM, not from the 256 MiB Infinity Cache
        copies = max(2, int(6e8 // (n * k)))
        xs = [torch.randn(T, k, device=dev, dtype=torch.bfloat16) for _ in range(2)]
        codes = [torch.randint(-128, 128, (n, k), device=dev, dtype=torch.int8) for _ in range(copies)]
        s8 = torch.rand(n, device=dev) * 1e-3 + 1e-4
        plan = int(lib.ffq_linear_wq_split(T, n, k, 0))
        row = []
        for split in sorted({1, 2, 4, 8, 16, plan}):
            if split > k // 256:
                continue
            us = t(lambda r: ops.linear_wq(xs[r % 2], codes[r % copies], s8, None, split=split))
            row.append(f"S={split}{'*' if split == plan else ''} {us:.1f}us")
        us = t(lambda r: ops.linear_wq(xs[r % 2], codes[r % copies], s8, None))
        if os.environ.get("SWEEP_QUICK"):
            print(f"{name:7s} N={n:5d} K={k:5d}: default {us:.1f}us", flush=True)
            del codes
            continue
        head = f"default {us:.1f}us = {n * k / us / 1e3:.0f} GB/s = {n * k / us / 8e6:.3f} of 8 TB/s"
        prev = lib.ffq_force_generic_kernels(1)
        try:
            us_t = t(lambda r: ops.linear_wq(xs[r % 2], codes[r % copies], s8, None))
        finally:
            lib.ffq_force_generic_kernels(prev)
        wd = [ops.dequantize_by_tile(c, s8, (1, k), None, torch.bfloat16) for c in codes[:max(2, copies // 2)]]
        us_v = t(lambda r: torch.nn.functional.linear(xs[r % 2], wd[r % len(wd)]))
        us_a = t(lambda r: torch.nn.functional.linear(xs[r % 2], ops.dequantize_by_tile(codes[r % copies], s8, (1, k), None, torch.bfloat16)))
        del wd
        w4 = [torch.randint(-8, 8, (n, k), device=dev, dtype=torch.int8) for _ in range(2)]
        packed = [ops.pack_int4(w, block=128) for w in w4] * max(1, copies)
        s4 = torch.rand(n * (k // 128), device=dev) * 1e-2 + 1e-3
        us_4 = t(lambda r: ops.linear_wq(xs[r % 2], packed[r % len(packed)], s4, None, group=128, pack_block=128))
        print(f"{name:7s} N={n:5d} K={k:5d}: {head} | 256-row tiles {us_t:.1f}us | vendor GEMM alone {us_v:.1f}us | A2+vendor {us_a:.1f}us | "
              f"W4 g128 packed {us_4:.1f}us = {n * k / 2 / us_4 / 1e3:.0f} GB/s || " + " ".join(row), flush=True)
        del codes, packed, w4
