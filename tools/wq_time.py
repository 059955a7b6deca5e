"""Time the weight-only linear (bf16 x int8 codes) on the Llama-3-8B shapes at T tokens against what it replaces:
A2 into a bf16 tensor + hipBLASLt's bf16 GEMM (hipGraph-replayed, HIP events)."""
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
tot = {"wq": 0.0, "wq_g128": 0.0, "blas": 0.0, "deq+blas": 0.0}
flops = 0.0
for name, n, k, cnt in (("qo", 4096, 4096, 2), ("kv", 1024, 4096, 2), ("gateup", 14336, 4096, 2), ("down", 4096, 14336, 1)):
    x = torch.randn(T, k, device=dev, dtype=torch.bfloat16)
    codes = (torch.randn(n, k, device=dev) * 40).round().clamp(-128, 127).to(torch.int8)
    codes4 = (torch.randn(n, k, device=dev) * 3).round().clamp(-8, 7).to(torch.int8)
    s = torch.rand(n, device=dev) * 1e-3 + 5e-4
    s4 = torch.rand(n * k // 128, device=dev) * 1e-2 + 5e-3
    w = ops.dequantize_by_tile(codes, s, (1, k), None, torch.bfloat16)
    ms = {
        "wq": event_time_ms(lambda r: ops.linear_wq(x, codes, s, None), iters=5, reps=4),
        "wq_g128": event_time_ms(lambda r: ops.linear_wq(x, codes4, s4, None, group=128), iters=5, reps=4),
        "blas": event_time_ms(lambda r: torch.nn.functional.linear(x, w), iters=5, reps=4),
        "deq+blas": event_time_ms(lambda r: torch.nn.functional.linear(x, ops.dequantize_by_tile(codes, s, (1, k), None, torch.bfloat16)), iters=5, reps=4),
    }
    f = 2.0 * T * n * k
    print(f"{name:7s} N={n:5d} K={k:5d} " + "  ".join(f"{key} {v:.4f} ms {f / v / 1e9:7.1f} TF" for key, v in ms.items()))
    for key, v in ms.items():
        tot[key] += cnt * v
    flops += cnt * f
print("layer mix: " + "  ".join(f"{key} {flops / v / 1e9:.1f} TFLOP/s ({v:.3f} ms)" for key, v in tot.items()))
