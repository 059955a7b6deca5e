"""Time residual add + RMSNorm + quantize ([14336, 4096] bf16, the hbm_kernels row, and the forward's [16384, 4096]); FFQ_LIB selects a variant build."""
import os, pathlib, sys, torch
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from fastforward_amd import _native, ops
if os.environ.get("FFQ_LIB"):
    from fastforward_amd._cabi import FFQLibrary
    _native._LIB = FFQLibrary(os.environ["FFQ_LIB"])
from bench import event_time_ms
dev = "cuda"
for rows in (14336, 16384):
    ws = [(torch.randn(rows, 4096, device=dev) * 0.5).to(torch.bfloat16) for _ in range(4)]
    gamma = torch.ones(4096, device=dev, dtype=torch.bfloat16)
    s1, o1 = torch.tensor([0.03], device=dev), torch.tensor([3.0], device=dev)
    for name, fn in (("sum + codes (7 B/elem)", lambda r: ops.add_rmsnorm_quantize(ws[r % 4], ws[(r + 1) % 4], gamma, 1e-5, [(s1, o1)])),
                     ("in-place sum + codes (7 B/elem)", lambda r: ops.add_rmsnorm_quantize(ws[r % 4], ws[(r + 1) % 4], gamma, 1e-5, [(s1, o1)], sum_inplace=True)),
                     ("sum + norm (8 B/elem)", lambda r: ops.add_rmsnorm_quantize(ws[r % 4], ws[(r + 1) % 4], gamma, 1e-5, (), want_norm=True))):
        bpe = 8 if "norm" in name else 7
        ms = min(event_time_ms(fn, iters=10, reps=5) for _ in range(3))
        print(f"[{rows}, 4096] {name:34s} {ms * 1e3:7.2f} us = {rows * 4096 * bpe / ms / 1e6:6.0f} GB/s = {rows * 4096 * bpe / ms / 8e9:.3f} of 8 TB/s", flush=True)
