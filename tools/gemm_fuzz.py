"""Random-shape fuzz of the int8 GEMM family against the exact accumulator (float64 matmul of the codes) and the restated fp32
epilogue: plain bf16 output, the residual epilogue where covered, MLP mode against silu_mul_quantize of the two plain outputs.
usage: python tools/gemm_fuzz.py [cases] [seed]"""
import pathlib, random, sys, torch
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from fastforward_amd import ops
dev = "cuda"
cases, seed = (int(sys.argv[1]) if len(sys.argv) > 1 else 40), (int(sys.argv[2]) if len(sys.argv) > 2 else 0)
rng = random.Random(seed)
bad = 0
for case in range(cases):
    m = rng.choice([rng.randint(1, 300), rng.randint(129, 6000), 256 * rng.randint(1, 40), rng.randint(4000, 20000)])
    n = rng.choice([8 * rng.randint(1, 40), 64 * rng.randint(2, 64), 128 * rng.randint(1, 40), rng.randint(1, 3000)])
    k = rng.choice([16 * rng.randint(1, 40), 64 * rng.randint(1, 64), 128 * rng.randint(2, 48)])
    g = torch.Generator(device=dev).manual_seed(seed * 1000 + case)
    xq = torch.randint(-128, 128, (m, k), device=dev, dtype=torch.int8, generator=g)
    wq = torch.randint(-128, 128, (n, k), device=dev, dtype=torch.int8, generator=g)
    sx, ox = torch.tensor([0.013], device=dev), (torch.tensor([float(rng.randint(-20, 20))], device=dev) if rng.random() < 0.8 else None)
    sw = torch.rand(n, device=dev, generator=g) * 1e-3 + 1e-4
    acc = (xq.double() @ wq.double().T).round()
    rsw = wq.sum(dim=1, dtype=torch.int64).float()
    v = acc.float() + (torch.round(ox) * rsw[None, :] if ox is not None else 0.0)
    want = ((sx * sw)[None, :] * v).to(torch.bfloat16)
    got = ops.linear_w8a8(xq, wq, sx, ox, sw, None, out_dtype=torch.bfloat16)
    ok = torch.equal(got, want)
    tags = ["plain"]
    res = (torch.randn(m, n, device=dev, generator=g) * 2).to(torch.bfloat16)
    fused = ops.linear_w8a8_residual(xq, wq, sx, ox, sw, res)
    if fused is not None:
        tags.append("residual")
        ok = ok and torch.equal(fused, res + want)
    if n % 128 == 0 and k % 64 == 0 and k >= 256:
        uq = torch.randint(-128, 128, (n, k), device=dev, dtype=torch.int8, generator=g)
        su = torch.rand(n, device=dev, generator=g) * 1e-3 + 1e-4
        so, oo = torch.tensor([0.02], device=dev), torch.tensor([float(rng.randint(-10, 10))], device=dev)
        codes = ops.mlp_gate_up_w8a8(xq, wq, uq, sx, ox, sw, su, so, oo, 8)
        if codes is not None:
            tags.append("mlp")
            up = ops.linear_w8a8(xq, uq, sx, ox, su, None, out_dtype=torch.bfloat16)
            _, (ref,) = ops.silu_mul_quantize(got, up, [(so, oo)], 8)
            ok = ok and torch.equal(codes, ref)
    bad += not ok
    print(f"{case:3d} M={m:6d} N={n:5d} K={k:5d} {'+'.join(tags):22s} {'ok' if ok else 'MISMATCH'}", flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
