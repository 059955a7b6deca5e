import sys, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from fastforward_amd import ops, _native
import os
if os.environ.get("FFQ_LIB"):
    from fastforward_amd._cabi import FFQLibrary
    _native._LIB = FFQLibrary(os.environ["FFQ_LIB"])
from bench import event_time_ms
dev = "cuda"
ws = [(torch.randn(14336, 4096, device=dev) * 0.02).to(torch.bfloat16) for _ in range(6)]
ms = event_time_ms(lambda r: ops.minmax_by_tile(ws[r % 6], (1, 4096)), iters=10, reps=12)
n = 14336 * 4096
print(f"per-channel [14336,4096] bf16: {ms*1e3:.2f} us  {n*2/ms/1e6:.0f} GB/s  {n*2/ms/1e6/8000:.3f}")
xs = [torch.randn(8, 2048, 14336, device=dev, dtype=torch.bfloat16) for _ in range(3)]
ms = event_time_ms(lambda r: ops.minmax_by_tile(xs[r % 3], xs[0].shape), iters=10, reps=6)
n = xs[0].numel()
print(f"per-tensor [8,2048,14336] bf16: {ms*1e3:.2f} us  {n*2/ms/1e6:.0f} GB/s  {n*2/ms/1e6/8000:.3f}")
xs = [torch.randn(8, 2048, 4096, device=dev, dtype=torch.bfloat16) for _ in range(6)]
ms = event_time_ms(lambda r: ops.minmax_by_tile(xs[r % 6], xs[0].shape), iters=10, reps=12)
n = xs[0].numel()
print(f"per-tensor [8,2048,4096] bf16: {ms*1e3:.2f} us  {n*2/ms/1e6:.0f} GB/s  {n*2/ms/1e6/8000:.3f}")
g = [(torch.randn(14336, 4096, device=dev) * 0.02).to(torch.bfloat16) for _ in range(6)]
ms = event_time_ms(lambda r: ops.minmax_by_tile(g[r % 6], (1, 128)), iters=10, reps=12)
n = 14336 * 4096
print(f"group-128 [14336,4096] bf16: {ms*1e3:.2f} us  {n*2/ms/1e6:.0f} GB/s  {n*2/ms/1e6/8000:.3f}")
