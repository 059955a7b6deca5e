"""Run the vendor library's int8 GEMM (torch._int_mm -> hipBLASLt) on one shape a few times (for rocprofv3 --pmc passes)."""
import sys
import torch

m, n, k = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (16384, 14336, 4096)))
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
xq = torch.randint(-128, 128, (m, k), device="cuda", dtype=torch.int8)
wt = torch.randint(-128, 128, (n, k), device="cuda", dtype=torch.int8).t()
for _ in range(reps):
    y = torch._int_mm(xq, wt)
torch.cuda.synchronize()
print("done", y.shape)
