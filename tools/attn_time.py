"""Time ffq_attention (+ the o_proj input quantizer) on the Llama-3-8B shape; FFQ_LIB=<variant .so> selects an experiment build."""
import os, pathlib, sys, torch
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from fastforward_amd import _native, ops
if os.environ.get("FFQ_LIB"):
    from fastforward_amd._cabi import FFQLibrary
    _native._LIB = FFQLibrary(os.environ["FFQ_LIB"])
from bench import event_time_ms
dev, b, s, h, hk, d = "cuda", 8, 2048, 32, 8, 128
qs = [torch.randn(b, s, h * d, device=dev).to(torch.bfloat16) for _ in range(3)]
ks = [torch.randn(b, s, hk * d, device=dev).to(torch.bfloat16) for _ in range(3)]
vs = [torch.randn(b, s, hk * d, device=dev).to(torch.bfloat16) for _ in range(3)]
sc, of = torch.tensor([0.03], device=dev), torch.tensor([-3.0], device=dev)
flops = 4 * b * h * s * s * d / 2
for rep in range(3):
    ms = event_time_ms(lambda r: ops.attention(qs[r % 3], ks[r % 3], vs[r % 3], d, causal=True, quantizer=(sc, of), want_context=False), iters=10, reps=8)
    print(f"attention B={b} S={s} H={h}/{hk}: {ms:.4f} ms  {flops / ms / 1e9:.1f} TFLOP/s")
