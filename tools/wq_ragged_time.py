"""Token counts that are not multiples of 256 on the one-wave-per-SIMD kernel (round 6: the activation pieces carry their row inside the
descriptor's range check) against the 8-wave kernel (ffq_force_generic_kernels) of the same library, two-pass form, Llama-3-8B shapes."""
import pathlib, sys
import torch
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from fastforward_amd import ops, _native
from bench import event_time_ms

lib = _native.library()
dev = "cuda"
torch.manual_seed(0)
t = lambda fn: event_time_ms(lambda r: fn(r), iters=6, reps=6) * 1e3  # noqa: E731
def both(fn):
    a = t(fn)
    previous = lib.ffq_force_generic_kernels(1)
    try:
        b = t(fn)
    finally:
        lib.ffq_force_generic_kernels(previous)
    return a, b
for T in (4300, 8191, 16000):
    x = {k: torch.randn(T, k, device=dev, dtype=torch.bfloat16) for k in (4096, 14336)}
    w = lambda n, k: (torch.randint(-128, 128, (n, k), device=dev, dtype=torch.int8), torch.rand(n, device=dev) * 1e-3 + 1e-4)  # noqa: E731
    for name, n, k in (("o", 4096, 4096), ("down", 4096, 14336)):
        c, s = w(n, k)
        a, b = both(lambda r: ops.linear_wq(x[k], c, s, None, two_pass=True, split=1))
        print(f"T={T:6d} {name:5s} one wave per SIMD {a:8.1f}us | 8-wave {b:8.1f}us | {b / a:5.3f}x", flush=True)
    (g, gs), (u, us) = w(14336, 4096), w(14336, 4096)
    a, b = both(lambda r: ops.mlp_gate_up_wq(x[4096], g, u, gs, None, us, None, two_pass=True, split=1))
    print(f"T={T:6d} mlp   one wave per SIMD {a:8.1f}us | 8-wave {b:8.1f}us | {b / a:5.3f}x", flush=True)
