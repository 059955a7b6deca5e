"""ffq_attention on the Llama-3-8B attention shape (B=8, S=2048, 32 q heads, 8 kv heads, D=128, causal):
correctness against float64 (and torch SDPA beside it) on small shapes, then time vs torch SDPA."""
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fastforward_amd import ops  # noqa: E402

dev = "cuda"
torch.manual_seed(0)


def ref64(q, k, v, d, causal=True):
    b, s, _ = q.shape
    h, hk = q.shape[2] // d, k.shape[2] // d
    qs = q.double().view(b, s, h, d).transpose(1, 2)
    ks = k.double().view(b, s, hk, d).transpose(1, 2).repeat_interleave(h // hk, dim=1)
    vs = v.double().view(b, s, hk, d).transpose(1, 2).repeat_interleave(h // hk, dim=1)
    w = qs @ ks.transpose(2, 3) * d**-0.5
    if causal:
        w = w.masked_fill(torch.ones(s, s, dtype=torch.bool, device=q.device).triu(1), float("-inf"))
    return (torch.softmax(w, -1) @ vs).transpose(1, 2).reshape(b, s, -1)


for (b, s, h, hk, causal) in [(1, 64, 1, 1, True), (1, 128, 2, 1, True), (2, 320, 4, 2, True), (1, 512, 8, 2, True), (1, 256, 4, 4, False), (2, 1024, 8, 2, True)]:
    d = 128
    q = torch.randn(b, s, h * d, device=dev).to(torch.bfloat16)
    k = torch.randn(b, s, hk * d, device=dev).to(torch.bfloat16)
    v = torch.randn(b, s, hk * d, device=dev).to(torch.bfloat16)
    sc, of = torch.tensor([0.03], device=dev), torch.tensor([-3.0], device=dev)
    ctx, codes = ops.attention(q, k, v, d, causal=causal, quantizer=(sc, of))
    torch.cuda.synchronize()
    want = ref64(q, k, v, d, causal)
    err = float((ctx.double() - want).abs().max())
    sd = F.scaled_dot_product_attention(q.view(b, s, h, d).transpose(1, 2), k.view(b, s, hk, d).transpose(1, 2), v.view(b, s, hk, d).transpose(1, 2),
                                        is_causal=causal, enable_gqa=h != hk).transpose(1, 2).reshape(b, s, -1)
    err_sdpa = float((sd.double() - want).abs().max())
    a1 = ops.quantize_by_tile(ctx, sc, ctx.shape, 8, torch.int8, of)
    print(f"B{b} S{s} H{h}/{hk} causal={causal}: max err vs f64 {err:.3e} (torch sdpa {err_sdpa:.3e}); codes == A1(ctx): {bool(torch.equal(a1, codes))}", flush=True)

b, s, hq, hk, d = 8, 2048, 32, 8, 128
q = torch.randn(b, s, hq * d, device=dev).to(torch.bfloat16)
k = torch.randn(b, s, hk * d, device=dev).to(torch.bfloat16)
v = torch.randn(b, s, hk * d, device=dev).to(torch.bfloat16)
sc, of = torch.tensor([0.03], device=dev), torch.tensor([-3.0], device=dev)


def t(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


flops = 4 * b * hq * s * s * d / 2
ctx, codes = ops.attention(q, k, v, d, quantizer=(sc, of))
sd = F.scaled_dot_product_attention(q.view(b, s, hq, d).transpose(1, 2), k.view(b, s, hk, d).transpose(1, 2), v.view(b, s, hk, d).transpose(1, 2),
                                    is_causal=True, enable_gqa=True).transpose(1, 2).reshape(b, s, -1)
print("full shape: max |ffq - sdpa|", float((ctx.float() - sd.float()).abs().max()), "nan:", bool(torch.isnan(ctx.float()).any()))
for name, fn in {
    "ffq_attention ctx + codes": lambda: ops.attention(q, k, v, d, quantizer=(sc, of)),
    "ffq_attention codes only": lambda: ops.attention(q, k, v, d, quantizer=(sc, of), want_context=False),
    "ffq_attention ctx only": lambda: ops.attention(q, k, v, d),
    "torch sdpa (transposed views, gqa)": lambda: F.scaled_dot_product_attention(q.view(b, s, hq, d).transpose(1, 2), k.view(b, s, hk, d).transpose(1, 2), v.view(b, s, hk, d).transpose(1, 2), is_causal=True, enable_gqa=True),
}.items():
    ms = t(fn)
    print(f"{name:40s} {ms:.3f} ms  {flops / ms / 1e9:.0f} TFLOP/s", flush=True)
