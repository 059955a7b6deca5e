"""A1 bf16 -> int8 timing on the headline shapes (hipGraph-replayed): python tools/q_time.py"""
import pathlib, sys
import torch
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from fastforward_amd import ops
from bench import event_time_ms
dev = "cuda"
for name, shape, tile in (("weights [14336,4096] per-channel", (14336, 4096), (1, 4096)), ("activations [8,2048,4096] per-tensor", (8, 2048, 4096), (8, 2048, 4096)),
                          ("weights [4096,14336] per-channel", (4096, 14336), (1, 14336)), ("group-128 [14336,4096]", (14336, 4096), (1, 128))):
    n = 1
    for d in shape: n *= d
    ws = [(torch.randn(shape, device=dev) * 0.02).to(torch.bfloat16) for _ in range(6)]
    nt = 1
    for s_, t_ in zip(shape, tile): nt *= s_ // t_
    scale = torch.rand(nt, device=dev) * 0.001 + 0.0005
    offset = torch.rand(nt, device=dev) * 4 - 2
    for off in (None, offset):
        ms = event_time_ms(lambda r: ops.quantize_by_tile(ws[r % 6], scale, tile, 8, torch.int8, off), iters=10, reps=12)
        print(f"{name:42s} offset={off is not None!s:5s} {ms*1e3:7.1f} us  {n*3/ms/1e6:7.1f} GB/s")
