"""Race hunt: the same launches many times, every result compared bit for bit with the first (int8 GEMM plain / gate+up mode,
weight-code GEMM two-pass / one-pass / MLP) at Llama-3-8B shapes and at a short-K many-tiles shape. usage: python tools/gemm_stress.py [reps]"""
import pathlib, sys
import torch
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from fastforward_amd import ops
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(1)
bad = 0
for (m, n, k) in ((16384, 4096, 4096), (16384, 14336, 4096), (16384, 4096, 14336), (8192, 8192, 256), (4096, 6144, 384)):
    xq = torch.randint(-128, 128, (m, k), device=dev, dtype=torch.int8, generator=g)
    wq = torch.randint(-128, 128, (n, k), device=dev, dtype=torch.int8, generator=g)
    uq = torch.randint(-128, 128, (n, k), device=dev, dtype=torch.int8, generator=g)
    sx, ox = torch.tensor([0.02], device=dev), torch.tensor([3.0], device=dev)
    sw = torch.rand(n, device=dev, generator=g) * 1e-3 + 1e-4
    so, oo = torch.tensor([0.05], device=dev), torch.tensor([-2.0], device=dev)
    xb = torch.randn(m, k, device=dev, generator=g).to(torch.bfloat16)
    fns = {
        "w8a8": lambda: ops.linear_w8a8(xq, wq, sx, ox, sw, None, out_dtype=torch.bfloat16),
        "w8a8 mlp": lambda: ops.mlp_gate_up_w8a8(xq, wq, uq, sx, ox, sw, sw, so, oo, 8),
        "wq two-pass": lambda: ops.linear_wq(xb, wq, sw, None, two_pass=True),
        "wq one-pass": lambda: ops.linear_wq(xb, wq, sw, None, two_pass=False),
        "wq mlp": lambda: ops.mlp_gate_up_wq(xb, wq, uq, sw, None, sw, None),
    }
    for name, fn in fns.items():
        first = fn()
        if first is None:
            continue
        diff = 0
        for _ in range(reps):
            diff += int(not torch.equal(fn(), first))
        bad += diff
        print(f"M={m} N={n} K={k} {name:12s}: {diff} of {reps} repeats differ", flush=True)
print("STRESS", "OK" if bad == 0 else "FAILED")
