"""Race hunt: the same launches many times, every result compared bit for bit with the first (int8 GEMM plain / gate+up mode,
weight-code GEMM two-pass (one-wave-per-SIMD kernel on whole tiles) / one-pass / MLP) at Llama-3-8B shapes and at a short-K
many-tiles shape, then the split-K forms at 2048 tokens (every admissible split, q/k/v in one launch, MLP mode), where the units
of a tile exchange partial sums through write-through slabs and counters. usage: python tools/gemm_stress.py [reps]"""
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
# split-K at 2048 tokens
t, k = 2048, 4096
x = torch.randn(t, k, device=dev, generator=g).to(torch.bfloat16)
ws = {n: torch.randint(-128, 128, (n, k), device=dev, dtype=torch.int8, generator=g) for n in (4096, 1024, 14336)}
ss = {n: torch.rand(n, device=dev, generator=g) * 1e-3 + 1e-4 for n in ws}
wd = torch.randint(-128, 128, (4096, 14336), device=dev, dtype=torch.int8, generator=g)
xd = torch.randn(t, 14336, device=dev, generator=g).to(torch.bfloat16)
cus = torch.cuda.get_device_properties(0).multi_processor_count
forms = {}
for n in (4096, 1024):
    for split in (2, 4, 8, 16):
        if (t // 256) * (n // 256) * split <= cus:
            forms[f"N={n} split {split}"] = (lambda n=n, split=split: ops.linear_wq(x, ws[n], ss[n], None, two_pass=False, split=split))
forms["N=4096 plan"] = lambda: ops.linear_wq(x, ws[4096], ss[4096], None)
forms["down plan"] = lambda: ops.linear_wq(xd, wd, ss[4096], None)
forms["q/k/v one launch"] = lambda: torch.cat(ops.linear_wq_multi(x, [ws[4096], ws[1024], ws[1024]], [ss[4096], ss[1024], ss[1024]], [None] * 3), dim=1)
forms["mlp (3 rounds + split tail)"] = lambda: ops.mlp_gate_up_wq(x, ws[14336], ws[14336], ss[14336], None, ss[14336], None)
for name, fn in forms.items():
    first = fn()
    diff = 0
    for _ in range(reps):
        diff += int(not torch.equal(fn(), first))
    bad += diff
    print(f"T={t} {name:28s}: {diff} of {reps} repeats differ", flush=True)
print("STRESS", "OK" if bad == 0 else "FAILED")
