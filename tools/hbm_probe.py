"""Launch each HBM-bound hot-path kernel of bench.py's `hbm_kernels` table a few times (for rocprofv3 --pmc passes)."""
import pathlib
import sys

import torch

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from fastforward_amd import ops  # noqa: E402

dev = "cuda"
shape = (14336, 4096)
ws = [(torch.randn(shape, device=dev) * 0.02).to(torch.bfloat16) for _ in range(4)]
scale = torch.rand(shape[0], device=dev) * 0.001 + 0.0005
tile = (1, shape[1])
s1, o1 = torch.tensor([0.03], device=dev), torch.tensor([3.0], device=dev)
gamma = torch.ones(shape[1], device=dev, dtype=torch.bfloat16)
g4 = torch.ones(shape[0] * shape[1] // 128, device=dev) * 0.01
acts = [torch.randn(8, 2048, 4096, device=dev, dtype=torch.bfloat16) for _ in range(2)]
for r in range(3):
    ops.quantize_dynamic_by_tile(acts[r % 2], (1, 1, 4096), 8, False, True, torch.int8)  # A3 per token: one launch
    w = ws[r % 4]
    q = ops.quantize_by_tile(w, scale, tile, 8, torch.int8)
    ops.quantize_by_tile(w, scale, tile, 8, torch.bfloat16)
    ops.dequantize_by_tile(q, scale, tile, None, torch.bfloat16)
    ops.minmax_by_tile(w, tile)
    ops.add_rmsnorm_quantize(w, ws[(r + 1) % 4], gamma, 1e-5, [(s1, o1)])
    ops.silu_mul_quantize(w, ws[(r + 1) % 4], [(s1, o1)])
    p = ops.quantize_pack_int4(w, g4, (1, 128), None, block=128)
    ops.unpack_dequantize_int4(p, g4, shape, (1, 128), None, block=128)
    ops.quantize_by_tile_backward(w, ws[(r + 1) % 4], scale, tile, 8.0, None)
torch.cuda.synchronize()
print("done")
