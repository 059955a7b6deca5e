"""ops.quantize_by_tile_unless_same on [16384, 4096] bf16 when the parameters are the earlier quantizer's (nothing to do) and when they
are not (A1), beside ops.quantize_by_tile; and ops.linear_w8a8_earlier beside ops.linear_w8a8 on the k / v shape. HIP events, us.
usage: python tools/siblings_time.py"""
import pathlib, sys
import torch
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from fastforward_amd import ops

DEV = "cuda"
x = (torch.randn(16384, 4096, device=DEV) * 2).to(torch.bfloat16)
t = lambda v: torch.tensor([v], device=DEV)  # noqa: E731
s, o, s2 = t(0.03), t(-2.6), t(0.04)


def timed(fn, n=50):
    for _ in range(5):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        a.record()
        for _ in range(n):
            fn()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / n * 1e3)
    return best


print(f"quantize_by_tile                          {timed(lambda: ops.quantize_by_tile(x, s, x.shape, 8, torch.int8, o)):7.1f} us")
print(f"quantize_by_tile_unless_same, same        {timed(lambda: ops.quantize_by_tile_unless_same(x, s, o, 8, s, o)):7.1f} us")
print(f"quantize_by_tile_unless_same, different   {timed(lambda: ops.quantize_by_tile_unless_same(x, s2, o, 8, s, o)):7.1f} us")
xq = ops.quantize_by_tile(x, s, x.shape, 8, torch.int8, o)
for n_out in (1024, 4096):
    wq = torch.randint(-128, 128, (n_out, 4096), device=DEV, dtype=torch.int8)
    sw, ow = torch.rand(n_out, device=DEV) * 1e-3 + 1e-4, torch.zeros(n_out, device=DEV)
    print(f"linear_w8a8 N={n_out} (weight offsets all zero) {timed(lambda: ops.linear_w8a8(xq, wq, s, o, sw, ow, None, out_dtype=torch.bfloat16)):7.1f} us")
    print(f"linear_w8a8_earlier N={n_out}                   {timed(lambda: ops.linear_w8a8_earlier(xq, (xq, s, o), wq, s, o, sw, ow, out_dtype=torch.bfloat16)):7.1f} us")
