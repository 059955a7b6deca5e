"""A/B of library builds on the weight-only GEMM (two-pass form: A2 + the bf16-image kernel) in ONE process, interleaved rounds:
usage: python tools/wq_ab.py T libA.so libB.so ...   ("-" = the shipped library). Prints ms per shape and build, the vendor's
bf16 GEMM beside them, and checks every build against the first bit for bit."""
import pathlib, sys
import torch
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from fastforward_amd import ops, _native
from fastforward_amd._cabi import FFQLibrary
from bench import event_time_ms

T = int(sys.argv[1])
shipped = _native.library()
libs = [(p, shipped if p == "-" else FFQLibrary(p)) for p in sys.argv[2:]]
dev = "cuda"
torch.manual_seed(0)
for name, n, k in (("qo", 4096, 4096), ("kv", 1024, 4096), ("gateup", 14336, 4096), ("down", 4096, 14336)):
    x = torch.randn(T, k, device=dev, dtype=torch.bfloat16)
    codes = torch.randint(-128, 128, (n, k), device=dev, dtype=torch.int8)
    s8 = torch.rand(n, device=dev) * 1e-3 + 1e-4
    wd = ops.dequantize_by_tile(codes, s8, (1, k), None, torch.bfloat16)
    best = {p: 1e9 for p, _ in libs}
    ref = None
    for rnd in range(3):
        for p, lib in libs:
            _native._LIB = lib
            out = ops.linear_wq(x, codes, s8, None, two_pass=True)
            if ref is None:
                ref = out.clone()
            assert torch.equal(out, ref), f"{p} differs from {libs[0][0]} on {name}"
            best[p] = min(best[p], event_time_ms(lambda r: ops.linear_wq(x, codes, s8, None, two_pass=True), iters=5, reps=3))
    _native._LIB = shipped
    v = min(event_time_ms(lambda r: torch.nn.functional.linear(x, ops.dequantize_by_tile(codes, s8, (1, k), None, torch.bfloat16)), iters=5, reps=3) for _ in range(3))
    f = 2.0 * T * n * k
    print(f"T={T} {name:7s}: " + " | ".join(f"{pathlib.Path(p).stem.replace('libffq_', '') if p != '-' else 'shipped'} {best[p]:.4f} ms {f / best[p] / 1e9:6.0f} TF" for p, _ in libs) + f" | A2 + vendor {v:.4f} ms {f / v / 1e9:6.0f} TF", flush=True)
