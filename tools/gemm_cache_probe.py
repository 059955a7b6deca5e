"""Does the int8 GEMM run faster when its operands were READ just before it (Infinity Cache), as the separate weight row-sum
pass does by accident? GEMM time alone (HIP events) after different preludes; operands are freshly written before every GEMM."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fastforward_amd import ops
dev = "cuda"
T = 16384
def run(name, N, K):
    torch.manual_seed(0)
    w = [(torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16) for _ in range(2)]
    scale = torch.rand(N, device=dev) * 0.001 + 0.0005
    a_src = [torch.randint(-128, 128, (T, K), dtype=torch.int8, device=dev) for _ in range(2)]
    a_buf = torch.empty_like(a_src[0])
    xs, xo = torch.tensor([0.02], device=dev), torch.tensor([4.0], device=dev)
    def once(variant, i):
        a_buf.copy_(a_src[i % 2])                       # the producer's write of the activation codes
        if variant == "fused sums":
            codes, rs = ops.quantize_rows_rowsum(w[i % 2], scale, None)
        else:
            codes, rs = ops.quantize_by_tile(w[i % 2], scale, (1, K), 8, torch.int8), None
        if variant == "touch A":
            _ = a_buf.view(torch.int32).sum()
        if variant == "touch A+B":
            _ = a_buf.view(torch.int32).sum(); _ = codes.view(torch.int32).sum()
            codes2, rs = codes, ops.quantize_rows_rowsum(w[i % 2], scale, None)[1]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.linear_w8a8(a_buf, codes, xs, xo, scale, None, w_rowsum=rs)
        e1.record()
        return e0, e1
    for variant in ("separate row sums (default)", "fused sums", "touch A", "touch A+B"):
        for i in range(3):
            once(variant, i)
        torch.cuda.synchronize()
        evs = [once(variant, i) for i in range(12)]
        torch.cuda.synchronize()
        ms = sorted(a.elapsed_time(b) for a, b in evs)[len(evs) // 2]
        print(f"{name:10s} N={N:5d} K={K:5d}  {variant:30s} GEMM (+ own row-sum launch if any) {ms*1e3:8.1f} us  {2.0*T*N*K/ms/1e9:7.0f} TOP/s", flush=True)
run("q/o_proj", 4096, 4096)
run("down_proj", 4096, 14336)
run("gate/up", 14336, 4096)
