"""Launches for a counter pass: the plain bf16-image weight-only GEMM (gate, up) and the one-launch MLP form, T tokens."""
import pathlib, sys
import torch
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from fastforward_amd import ops
T = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
n, k = 14336, 4096
torch.manual_seed(0)
x = torch.randn(T, k, device="cuda", dtype=torch.bfloat16)
pair = []
for i in range(2):
    w = (torch.randn(n, k, device="cuda") * 0.02).to(torch.bfloat16)
    lo, hi = ops.minmax_by_tile(w, (1, k))
    s8, _ = ops.parameters_for_range(lo, hi, 8, True, False, want_offset=False)
    pair.append((ops.quantize_by_tile(w, s8, (1, k), 8, torch.int8), s8))
(gc, gs), (uc, us) = pair
for _ in range(4):
    ops.linear_wq(x, gc, gs, None)
    ops.linear_wq(x, uc, us, None)
    ops.mlp_gate_up_wq(x, gc, uc, gs, None, us, None)
torch.cuda.synchronize()
