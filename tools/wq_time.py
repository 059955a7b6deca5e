"""Time and check the weight-only linear (bf16 x weight codes) on the Llama-3-8B shapes at T tokens against what it
replaces: A2 into a bf16 tensor + the vendor's bf16 GEMM (hipGraph-replayed launches, HIP events).
usage: python tools/wq_time.py [T] [--check]"""
import pathlib, sys
import torch
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import os
from fastforward_amd import ops, _native
if os.environ.get("FFQ_LIB"):  # an experiments build (tools/build_experiments.sh)
    from fastforward_amd._cabi import FFQLibrary
    _native._LIB = FFQLibrary(os.environ["FFQ_LIB"])
from bench import event_time_ms

T = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 16384
check = "--check" in sys.argv
dev = "cuda"
torch.manual_seed(0)
tot, flops = {}, 0.0
for name, n, k, cnt in (("qo", 4096, 4096, 2), ("kv", 1024, 4096, 2), ("gateup", 14336, 4096, 2), ("down", 4096, 14336, 1)):
    x = torch.randn(T, k, device=dev, dtype=torch.bfloat16)
    w = (torch.randn(n, k, device=dev) * 0.02).to(torch.bfloat16)
    lo, hi = ops.minmax_by_tile(w, (1, k))
    s8, _ = ops.parameters_for_range(lo, hi, 8, True, False, want_offset=False)
    codes8 = ops.quantize_by_tile(w, s8, (1, k), 8, torch.int8)
    glo, ghi = ops.minmax_by_tile(w, (1, 128))
    s4, _ = ops.parameters_for_range(glo, ghi, 4, True, False, want_offset=False)
    codes4 = ops.quantize_by_tile(w, s4, (1, 128), 4, torch.int8)
    packed4 = ops.pack_int4(codes4, block=128)
    w8 = ops.dequantize_by_tile(codes8, s8, (1, k), None, torch.bfloat16)
    w4 = ops.dequantize_by_tile(codes4, s4, (1, 128), None, torch.bfloat16)
    variants = {
        "w8 one-pass": lambda r: ops.linear_wq(x, codes8, s8, None, two_pass=False),
        "w8 two-pass": lambda r: ops.linear_wq(x, codes8, s8, None, two_pass=True),
        "w4g128 one-pass": lambda r: ops.linear_wq(x, codes4, s4, None, group=128, two_pass=False),
        "w4g128 packed": lambda r: ops.linear_wq(x, packed4, s4, None, group=128, pack_block=128, two_pass=False),
        "w4g128 packed two-pass": lambda r: ops.linear_wq(x, packed4, s4, None, group=128, pack_block=128, two_pass=True),
        "vendor bf16": lambda r: torch.nn.functional.linear(x, w8),
        "A2 + vendor": lambda r: torch.nn.functional.linear(x, ops.dequantize_by_tile(codes8, s8, (1, k), None, torch.bfloat16)),
    }
    if check:
        ref8, ref4 = x.double() @ w8.double().T, x.double() @ w4.double().T
        for key, want in (("w8 one-pass", ref8), ("w8 two-pass", ref8), ("w4g128 one-pass", ref4), ("w4g128 packed", ref4), ("w4g128 packed two-pass", ref4)):
            got = variants[key](0).double()
            err = float(((got - want).abs() / (want.abs() * 2.0**-8 + 1e-3)).max())
            print(f"  check {name:7s} {key:24s} max |err| / (2^-8 |y| + 1e-3) = {err:.3f} {'ok' if err <= 1.0 else 'WRONG'}")
        assert torch.equal(variants["w8 one-pass"](0), variants["w8 two-pass"](0)), "one-pass and two-pass forms differ"
        assert torch.equal(variants["w4g128 one-pass"](0), variants["w4g128 packed"](0)) and torch.equal(variants["w4g128 packed"](0), variants["w4g128 packed two-pass"](0))
        del ref8, ref4
    f = 2.0 * T * n * k
    ms = {key: event_time_ms(fn, iters=5, reps=4) for key, fn in variants.items()}
    print(f"{name:7s} N={n:5d} K={k:5d} " + "  ".join(f"{key} {v:.4f} ms {f / v / 1e9:6.0f} TF |" for key, v in ms.items()))
    for key, v in ms.items():
        tot[key] = tot.get(key, 0.0) + cnt * v
    flops += cnt * f
# the MLP front half: two launches + SiLU * up against the one-launch form (ops.mlp_gate_up_wq), same weights
n, k = 14336, 4096
x = torch.randn(T, k, device=dev, dtype=torch.bfloat16)
pair = []
for i in range(2):
    w = (torch.randn(n, k, device=dev) * 0.02).to(torch.bfloat16)
    lo, hi = ops.minmax_by_tile(w, (1, k))
    s8, _ = ops.parameters_for_range(lo, hi, 8, True, False, want_offset=False)
    pair.append((ops.quantize_by_tile(w, s8, (1, k), 8, torch.int8), s8, w))
(gc, gs, gw), (uc, us, uw) = pair
del pair
mlp = {
    "2 x linear_wq + silu_mul": lambda r: ops.silu_mul_quantize(ops.linear_wq(x, gc, gs, None), ops.linear_wq(x, uc, us, None), (), want_product=True),
    "mlp_gate_up_wq one-pass": lambda r: ops.mlp_gate_up_wq(x, gc, uc, gs, None, us, None, two_pass=False),
    "mlp_gate_up_wq": lambda r: ops.mlp_gate_up_wq(x, gc, uc, gs, None, us, None),
    "2 x (A2 + vendor) + silu_mul": lambda r: ops.silu_mul_quantize(torch.nn.functional.linear(x, ops.dequantize_by_tile(gc, gs, (1, k), None, torch.bfloat16)),
                                                                  torch.nn.functional.linear(x, ops.dequantize_by_tile(uc, us, (1, k), None, torch.bfloat16)), (), want_product=True),
}
if check:
    assert torch.equal(mlp["mlp_gate_up_wq"](0), mlp["2 x linear_wq + silu_mul"](0)[0]) and torch.equal(mlp["mlp_gate_up_wq one-pass"](0), mlp["mlp_gate_up_wq"](0))
    print("  check mlp_gate_up_wq == silu_mul(linear_wq, linear_wq): ok")
f = 2.0 * T * 2 * n * k
print("gate+up+silu*up (w8): " + "  ".join(f"{key} {v:.4f} ms {f / v / 1e9:6.0f} TF |" for key, v in ((key, event_time_ms(fn, iters=5, reps=4)) for key, fn in mlp.items())))
print(f"layer mix at T = {T}: " + "  ".join(f"{key} {flops / v / 1e9:.0f} TFLOP/s ({v:.3f} ms) |" for key, v in tot.items()))
