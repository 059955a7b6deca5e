"""Time the residual add in the GEMM epilogue against the plain GEMM + the RMSNorm kernel that adds (Llama-3-8B o_proj / down_proj)."""
import pathlib, sys, torch
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from fastforward_amd import ops
from bench import event_time_ms
dev, T = "cuda", 16384
s1, o1 = torch.tensor([0.03], device=dev), torch.tensor([3.0], device=dev)
gamma = torch.ones(4096, device=dev, dtype=torch.bfloat16)
for name, n, k in (("o_proj", 4096, 4096), ("down_proj", 4096, 14336)):
    xq = torch.randint(-128, 128, (T, k), device=dev, dtype=torch.int8)
    wq = torch.randint(-128, 128, (n, k), device=dev, dtype=torch.int8)
    sx, ox = torch.tensor([0.02], device=dev), torch.tensor([4.0], device=dev)
    sw = torch.rand(n, device=dev) * 0.001 + 0.0005
    hs = [torch.randn(T, n, device=dev).to(torch.bfloat16) for _ in range(3)]
    plain = event_time_ms(lambda r: ops.linear_w8a8(xq, wq, sx, ox, sw, None, out_dtype=torch.bfloat16), iters=5, reps=4)
    fused = event_time_ms(lambda r: ops.linear_w8a8_residual(xq, wq, sx, ox, sw, hs[r % 3], inplace=True), iters=5, reps=4)
    y = ops.linear_w8a8(xq, wq, sx, ox, sw, None, out_dtype=torch.bfloat16)
    norm_add = event_time_ms(lambda r: ops.add_rmsnorm_quantize(hs[r % 3], y, gamma, 1e-5, [(s1, o1)], sum_inplace=True), iters=5, reps=4)
    norm_only = event_time_ms(lambda r: ops.add_rmsnorm_quantize(hs[r % 3], None, gamma, 1e-5, [(s1, o1)]), iters=5, reps=4)
    print(f"{name:10s} gemm {plain*1e3:7.1f} us  gemm+residual {fused*1e3:7.1f} us  | rmsnorm with add {norm_add*1e3:6.1f} us  without {norm_only*1e3:6.1f} us"
          f"  | pair {1e3*(plain+norm_add):7.1f} -> {1e3*(fused+norm_only):7.1f} us")
