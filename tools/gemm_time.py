"""Time the W8A8 GEMM on the Llama-3-8B shapes at T tokens (hipGraph-replayed, HIP events)."""
import sys, pathlib, statistics
import torch
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from fastforward_amd import ops, _native
import os
if os.environ.get("FFQ_LIB"):  # experiment builds (tools/build_experiments.sh)
    from fastforward_amd._cabi import FFQLibrary
    _native._LIB = FFQLibrary(os.environ["FFQ_LIB"])
from bench import event_time_ms

T = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
dev = "cuda"
tot_ops = tot_ms = 0
SHAPES = (("qo", 4096, 4096, 2), ("kv", 1024, 4096, 2), ("gateup", 14336, 4096, 2), ("down", 4096, 14336, 1))
if os.environ.get("GT_MODEL") == "70b":
    SHAPES = (("qo", 8192, 8192, 2), ("kv", 1024, 8192, 2), ("gateup", 28672, 8192, 2), ("down", 8192, 28672, 1))
for name, n, k, cnt in SHAPES:
    xq = torch.randint(-128, 128, (T, k), device=dev, dtype=torch.int8)
    wq = torch.randint(-128, 128, (n, k), device=dev, dtype=torch.int8)
    if os.environ.get("GT_FILL") == "zero":
        xq.zero_(); wq.zero_()
    elif os.environ.get("GT_FILL") == "gauss":  # what real quantized weights / activations look like
        xq = (torch.randn(T, k, device=dev) * 20).round().clamp(-128, 127).to(torch.int8)
        wq = (torch.randn(n, k, device=dev) * 30).round().clamp(-128, 127).to(torch.int8)
    sx, ox = torch.tensor([0.02], device=dev), (None if os.environ.get("GT_NO_XOFF") else torch.tensor([4.0], device=dev))
    sw = torch.rand(n, device=dev) * 0.001 + 0.0005
    ms = event_time_ms(lambda r: ops.linear_w8a8(xq, wq, sx, ox, sw, None, out_dtype=torch.bfloat16), iters=5, reps=4)
    print(f"{name:7s} N={n:5d} K={k:5d} {ms:.4f} ms {2*T*n*k/ms/1e9:8.1f} TOP/s")
    tot_ops += cnt * 2 * T * n * k; tot_ms += cnt * ms
# gate+up with the SiLU*up + quantize epilogue (one launch); realistic magnitudes so that silu sees ordinary values
n, k = (28672, 8192) if os.environ.get("GT_MODEL") == "70b" else (14336, 4096)
xq = torch.randint(-128, 128, (T, k), device=dev, dtype=torch.int8)
gq = torch.randint(-128, 128, (n, k), device=dev, dtype=torch.int8)
uq = torch.randint(-128, 128, (n, k), device=dev, dtype=torch.int8)
sx, ox = torch.tensor([0.02], device=dev), torch.tensor([4.0], device=dev)
sg = torch.rand(n, device=dev) * 0.00002 + 0.00002
so, oo = torch.tensor([0.03], device=dev), torch.tensor([-3.0], device=dev)
ms = event_time_ms(lambda r: ops.mlp_gate_up_w8a8(xq, gq, uq, sx, ox, sg, sg, so, oo, 8), iters=5, reps=4)
print(f"mlp     N={2*n:5d} K={k:5d} {ms:.4f} ms {4*T*n*k/ms/1e9:8.1f} TOP/s")
print(f"layer mix: {tot_ops/tot_ms/1e9:.1f} TOP/s  ({tot_ms:.3f} ms per layer)")
