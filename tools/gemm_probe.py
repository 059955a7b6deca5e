"""Run one W8A8 GEMM shape a few times (for rocprofv3 --pmc passes)."""
import sys, pathlib
import torch
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from fastforward_amd import ops

m, n, k = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (16384, 14336, 4096)))
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
dev = "cuda"
xq = torch.randint(-128, 128, (m, k), device=dev, dtype=torch.int8)
wq = torch.randint(-128, 128, (n, k), device=dev, dtype=torch.int8)
sx, ox = torch.tensor([0.02], device=dev), torch.tensor([4.0], device=dev)
sw = torch.rand(n, device=dev) * 0.001 + 0.0005
for _ in range(reps):
    y = ops.linear_w8a8(xq, wq, sx, ox, sw, None, out_dtype=torch.bfloat16)
torch.cuda.synchronize()
print("done", y.shape)
