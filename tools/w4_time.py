"""Time the fused W4 group-128 quantize+pack and unpack+dequantize on [14336, 4096] bf16 (the hbm_kernels rows); FFQ_LIB selects a variant build."""
import os, pathlib, sys, torch
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from fastforward_amd import _native, ops
if os.environ.get("FFQ_LIB"):
    from fastforward_amd._cabi import FFQLibrary
    _native._LIB = FFQLibrary(os.environ["FFQ_LIB"])
from bench import event_time_ms
dev, shape = "cuda", (14336, 4096)
ws = [(torch.randn(shape, device=dev) * 0.02).to(torch.bfloat16) for _ in range(4)]
g4 = torch.ones(shape[0] * shape[1] // 128, device=dev) * 0.01
o4 = torch.round(torch.randn(shape[0] * shape[1] // 128, device=dev))
n = shape[0] * shape[1]
for name, off in (("no offset", None), ("offset", o4)):
    ms = min(event_time_ms(lambda r: ops.quantize_pack_int4(ws[r % 4], g4, (1, 128), off, block=128), iters=10, reps=5) for _ in range(3))
    print(f"quantize+pack {name:10s} {ms * 1e3:7.2f} us = {n * 2.5 / ms / 1e6:6.0f} GB/s = {n * 2.5 / ms / 8e9:.3f} of 8 TB/s", flush=True)
p = ops.quantize_pack_int4(ws[0], g4, (1, 128), None, block=128)
ms = min(event_time_ms(lambda r: ops.unpack_dequantize_int4(p, g4, shape, (1, 128), None, block=128), iters=10, reps=5) for _ in range(3))
print(f"unpack+dequantize        {ms * 1e3:7.2f} us = {n * 2.5 / ms / 1e6:6.0f} GB/s = {n * 2.5 / ms / 8e9:.3f} of 8 TB/s", flush=True)
