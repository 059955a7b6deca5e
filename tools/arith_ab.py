"""A/B of two builds of the library on one box: A1 (per-channel / per-tensor / group-128), the fused W4 quantize+pack and the
RMSNorm producer. usage: python tools/arith_ab.py [other_lib.so]"""
import os, pathlib, sys, torch
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from fastforward_amd import _native, ops
from fastforward_amd._cabi import FFQLibrary
from bench import event_time_ms
dev = "cuda"
libs = {"current": _native.library()}
for path in sys.argv[1:]:
    libs[pathlib.Path(path).stem.replace("libffq_", "")] = FFQLibrary(path)
shape = (14336, 4096)
n = shape[0] * shape[1]
ws = [(torch.randn(shape, device=dev) * 0.02).to(torch.bfloat16) for _ in range(6)]
scale = torch.rand(shape[0], device=dev) * 0.001 + 0.0005
g4 = torch.rand(n // 128, device=dev) * 0.002 + 0.002
s1, o1 = torch.tensor([0.03], device=dev), torch.tensor([3.0], device=dev)
gamma = torch.ones(shape[1], device=dev, dtype=torch.bfloat16)
packed = [ops.quantize_pack_int4(w, g4, (1, 128), None, block=128) for w in ws]
cases = {
    "A1 per-channel bf16->int8 (3 B/elem)": (3, lambda r: ops.quantize_by_tile(ws[r % 6], scale, (1, shape[1]), 8, torch.int8)),
    "A1 per-channel bf16->bf16 (4 B/elem)": (4, lambda r: ops.quantize_by_tile(ws[r % 6], scale, (1, shape[1]), 8, torch.bfloat16)),
    "A1 per-tensor bf16->int8 (3 B/elem)": (3, lambda r: ops.quantize_by_tile(ws[r % 6], s1, shape, 8, torch.int8, o1)),
    "W4 group-128 quantize+pack (2.5 B/elem)": (2.5, lambda r: ops.quantize_pack_int4(ws[r % 6], g4, (1, 128), None, block=128)),
    "W4 group-128 unpack+dequantize (2.5 B/elem)": (2.5, lambda r: ops.unpack_dequantize_int4(packed[r % 6], g4, shape, (1, 128), None, block=128)),
    "add+RMSNorm+quantize (7 B/elem)": (7, lambda r: ops.add_rmsnorm_quantize(ws[r % 6], ws[(r + 1) % 6], gamma, 1e-5, [(s1, o1)])),
    "SiLU*up+quantize (5 B/elem)": (5, lambda r: ops.silu_mul_quantize(ws[r % 6], ws[(r + 1) % 6], [(s1, o1)])),
    "backward per-channel (6 B/elem)": (6, lambda r: ops.quantize_by_tile_backward(ws[r % 6], ws[(r + 1) % 6], scale, (1, shape[1]), 8.0, None)),
}
for rep in range(2):
    for name, (bpe, fn) in cases.items():
        row = []
        for tag, lib in libs.items():
            _native._LIB = lib
            ms = event_time_ms(fn, iters=10, reps=12)
            row.append(f"{tag} {ms*1e3:6.1f} us {n*bpe/ms/1e6:6.0f} GB/s")
        print(f"{name:42s} " + "   ".join(row), flush=True)
