import sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '.')
from fastforward_amd import ops, _native
from bench import event_time_ms
lib = _native.library()
t = lambda fn: event_time_ms(lambda r: fn(r), iters=6, reps=6) * 1e3
dev = "cuda"
w = lambda n, k: (torch.randint(-128, 128, (n, k), device=dev, dtype=torch.int8), torch.rand(n, device=dev) * 1e-3 + 1e-4)
for T in (1536, 2048, 3072, 4096, 4300, 5120, 6144, 8191):
    x = {k: torch.randn(T, k, device=dev, dtype=torch.bfloat16) for k in (4096, 14336)}
    for name, n, k in (("o", 4096, 4096), ("down", 4096, 14336), ("gateup1", 14336, 4096)):
        c, s = w(n, k)
        plan = int(lib.ffq_linear_wq_split(T, n, k, 0))
        a = t(lambda r: ops.linear_wq(x[k], c, s, None, two_pass=True))
        b = t(lambda r: ops.linear_wq(x[k], c, s, None, two_pass=True, split=1))
        print(f"T={T:5d} {name:8s} plan S={plan}: {a:8.1f}us | S=1 (one wave per SIMD): {b:8.1f}us | {a / b:5.3f}", flush=True)
    ws = [w(n, 4096) for n in (4096, 1024, 1024)]
    plan = int(lib.ffq_linear_wq_split(T, 6144, 4096, 0))
    a = t(lambda r: ops.linear_wq_multi(x[4096], [c for c, _ in ws], [s for _, s in ws], [None] * 3, two_pass=True))
    b = t(lambda r: ops.linear_wq_multi(x[4096], [c for c, _ in ws], [s for _, s in ws], [None] * 3, two_pass=True, split=1))
    print(f"T={T:5d} q/k/v    plan S={plan}: {a:8.1f}us | S=1 (one wave per SIMD): {b:8.1f}us | {a / b:5.3f}", flush=True)
