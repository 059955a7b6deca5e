"""Time ops.quantize_rows_batch over all 32 decoder layers' weights of Llama-3-8B (6.98 G elements, 3 B/elem).
usage: [FFQ_LIB=...] python tools/wbatch_time.py"""
import os, pathlib, sys
import torch
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from fastforward_amd import ops, _native
if os.environ.get("FFQ_LIB"):
    from fastforward_amd._cabi import FFQLibrary
    _native._LIB = FFQLibrary(os.environ["FFQ_LIB"])
from bench import event_time_ms
dev = "cuda"
shapes = [(4096, 4096), (1024, 4096), (1024, 4096), (4096, 4096), (14336, 4096), (14336, 4096), (4096, 14336)]
layers = [[(torch.randn(s, device=dev) * 0.02).to(torch.bfloat16) for s in shapes] for _ in range(32)]
scales = [torch.rand(s[0], device=dev) * 1e-3 + 1e-4 for s in shapes]
def run(r):
    for ws in layers:
        ops.quantize_rows_batch(ws, scales, [None] * 7, 8)
ms = event_time_ms(run, iters=3, reps=4)
elems = 32 * sum(a * b for a, b in shapes)
print(f"{os.environ.get('FFQ_LIB', 'shipped')}: {ms:.3f} ms  {elems * 3 / ms / 1e6:.0f} GB/s  {elems * 3 / ms / 1e6 / 8000:.4f} of 8 TB/s")
