"""Three ffq_attention launches (codes only) on the Llama-3-8B shape — the target of tools/pmc_attn.sh."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fastforward_amd import ops
b, s, hq, hk, d = 8, 2048, 32, 8, 128
torch.manual_seed(0)
q = torch.randn(b, s, hq * d, device="cuda").to(torch.bfloat16)
k = torch.randn(b, s, hk * d, device="cuda").to(torch.bfloat16)
v = torch.randn(b, s, hk * d, device="cuda").to(torch.bfloat16)
sc, of = torch.tensor([0.03], device="cuda"), torch.tensor([-3.0], device="cuda")
for _ in range(3):
    ops.attention(q, k, v, d, quantizer=(sc, of), want_context=False)
torch.cuda.synchronize()
