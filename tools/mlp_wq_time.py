"""Time the one-launch weight-only MLP front half (ops.mlp_gate_up_wq) against its parts at T tokens.
usage: [FFQ_LIB=tools/_exp/libffq_x.so] python tools/mlp_wq_time.py [T]"""
import os, pathlib, sys
import torch
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from fastforward_amd import ops, _native
if os.environ.get("FFQ_LIB"):
    from fastforward_amd._cabi import FFQLibrary
    _native._LIB = FFQLibrary(os.environ["FFQ_LIB"])
from bench import event_time_ms

T = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
dev = "cuda"
torch.manual_seed(0)
n, k = 14336, 4096
x = torch.randn(T, k, device=dev, dtype=torch.bfloat16)
pair = []
for i in range(2):
    w = (torch.randn(n, k, device=dev) * 0.02).to(torch.bfloat16)
    lo, hi = ops.minmax_by_tile(w, (1, k))
    s8, _ = ops.parameters_for_range(lo, hi, 8, True, False, want_offset=False)
    pair.append((ops.quantize_by_tile(w, s8, (1, k), 8, torch.int8), s8))
(gc, gs), (uc, us) = pair
img = [ops.dequantize_by_tile(c, s, (1, k), None, torch.bfloat16) for c, s in pair]
variants = {
    "linear_wq x1 (two-pass)": lambda r: ops.linear_wq(x, gc, gs, None),
    "2 x linear_wq + silu_mul": lambda r: ops.silu_mul_quantize(ops.linear_wq(x, gc, gs, None), ops.linear_wq(x, uc, us, None), (), want_product=True),
    "mlp one-pass": lambda r: ops.mlp_gate_up_wq(x, gc, uc, gs, None, us, None, two_pass=False),
    "mlp two-pass": lambda r: ops.mlp_gate_up_wq(x, gc, uc, gs, None, us, None),
    "mlp two-pass, gate twice": lambda r: ops.mlp_gate_up_wq(x, gc, gc, gs, None, gs, None),
    "vendor x2 + silu_mul": lambda r: ops.silu_mul_quantize(torch.nn.functional.linear(x, img[0]), torch.nn.functional.linear(x, img[1]), (), want_product=True),
}
f = 2.0 * T * 2 * n * k
print(os.environ.get("FFQ_LIB", "shipped"), "T =", T)
for key, fn in variants.items():
    v = event_time_ms(fn, iters=5, reps=4)
    print(f"  {key:28s} {v:.4f} ms  {f / v / 1e9 * (0.5 if 'x1' in key else 1.0):6.0f} TF")
