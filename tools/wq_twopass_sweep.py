"""Where the two-pass form of the weight-only linear (A2 of the whole weight into a bf16 image, then the one-wave-per-SIMD GEMM) overtakes
the one-pass form (codes converted inside the 8-wave GEMM, split-K of the partly filled round): Llama-3-8B shapes, token counts that are
multiples of 256 below the library's threshold of 4096, both forced through `two_pass=`; A2 + the vendor's GEMM beside them.
usage: python tools/wq_twopass_sweep.py [T ...]"""
import pathlib, sys
import torch
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from fastforward_amd import ops
from bench import event_time_ms

dev = "cuda"
torch.manual_seed(0)
t = lambda fn: event_time_ms(lambda r: fn(r), iters=6, reps=6) * 1e3  # noqa: E731
for T in [int(a) for a in sys.argv[1:] if a.isdigit()] or [768, 1024, 1536, 2048, 3072, 4096]:
    print(f"== T = {T}")
    x = {k: torch.randn(T, k, device=dev, dtype=torch.bfloat16) for k in (4096, 14336)}
    def weight(n, k):
        return torch.randint(-128, 128, (n, k), device=dev, dtype=torch.int8), torch.rand(n, device=dev) * 1e-3 + 1e-4
    for name, n, k in (("o", 4096, 4096), ("down", 4096, 14336)):
        w, s = weight(n, k)
        one = t(lambda r: ops.linear_wq(x[k], w, s, None, two_pass=False))
        two = t(lambda r: ops.linear_wq(x[k], w, s, None, two_pass=True))
        ven = t(lambda r: torch.nn.functional.linear(x[k], ops.dequantize_by_tile(w, s, (1, k), None, torch.bfloat16)))
        print(f"{name:6s} N={n:5d} K={k:5d} one-pass {one:7.1f}us | two-pass {two:7.1f}us | A2+vendor {ven:7.1f}us | two/one {two / one:5.3f}", flush=True)
    ws = [weight(n, 4096) for n in (4096, 1024, 1024)]
    one = t(lambda r: ops.linear_wq_multi(x[4096], [w for w, _ in ws], [s for _, s in ws], [None] * 3, two_pass=False))
    two = t(lambda r: ops.linear_wq_multi(x[4096], [w for w, _ in ws], [s for _, s in ws], [None] * 3, two_pass=True))
    ven = t(lambda r: [torch.nn.functional.linear(x[4096], ops.dequantize_by_tile(w, s, (1, 4096), None, torch.bfloat16)) for w, s in ws])
    print(f"q/k/v  N= 6144 K= 4096 one-pass {one:7.1f}us | two-pass {two:7.1f}us | A2+vendor {ven:7.1f}us | two/one {two / one:5.3f}", flush=True)
    (g, gs), (u, us) = weight(14336, 4096), weight(14336, 4096)
    one = t(lambda r: ops.mlp_gate_up_wq(x[4096], g, u, gs, None, us, None, two_pass=False))
    two = t(lambda r: ops.mlp_gate_up_wq(x[4096], g, u, gs, None, us, None, two_pass=True))
    ven = t(lambda r: ops.silu_mul_quantize(torch.nn.functional.linear(x[4096], ops.dequantize_by_tile(g, gs, (1, 4096), None, torch.bfloat16)),
                                            torch.nn.functional.linear(x[4096], ops.dequantize_by_tile(u, us, (1, 4096), None, torch.bfloat16)), (), want_product=True))
    print(f"mlp    N=28672 K= 4096 one-pass {one:7.1f}us | two-pass {two:7.1f}us | A2+vendor {ven:7.1f}us | two/one {two / one:5.3f}", flush=True)
