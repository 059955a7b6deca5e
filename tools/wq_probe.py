"""Run one weight-only GEMM shape a few times (for rocprofv3 passes). FFQ_LIB selects an experiment build."""
import os, pathlib, sys
import torch
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from fastforward_amd import ops, _native
if os.environ.get("FFQ_LIB"):
    from fastforward_amd._cabi import FFQLibrary
    _native._LIB = FFQLibrary(os.environ["FFQ_LIB"])
m, n, k = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (16384, 14336, 4096)))
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
x = torch.randn(m, k, device="cuda", dtype=torch.bfloat16)
codes = (torch.randn(n, k, device="cuda") * 40).round().clamp(-128, 127).to(torch.int8)
s = torch.rand(n, device="cuda") * 1e-3 + 5e-4
w = ops.dequantize_by_tile(codes, s, (1, k), None, torch.bfloat16)
for _ in range(reps):
    y = ops.linear_wq(x, codes, s, None)
    z = torch.nn.functional.linear(x, w)
torch.cuda.synchronize()
print("done", y.shape)
