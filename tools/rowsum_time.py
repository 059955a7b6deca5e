import sys, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from fastforward_amd import ops
from bench import event_time_ms
dev = "cuda"
for shape in ((14336, 4096), (4096, 14336), (4096, 4096), (1024, 4096)):
    ws = [(torch.randn(shape, device=dev) * 0.02).to(torch.bfloat16) for _ in range(6)]
    scale = torch.rand(shape[0], device=dev) * 0.001 + 0.0005
    pool = torch.zeros(shape[0], dtype=torch.int32, device=dev)
    n = shape[0] * shape[1]
    a = event_time_ms(lambda r: ops.quantize_by_tile(ws[r % 6], scale, (1, shape[1]), 8, torch.int8), iters=10, reps=12)
    b = event_time_ms(lambda r: ops.quantize_rows_rowsum(ws[r % 6], scale, None, 8.0, rowsum_out=pool), iters=10, reps=12)
    q = ops.quantize_by_tile(ws[0], scale, (1, shape[1]), 8, torch.int8)
    xq = torch.randint(-128, 128, (64, shape[1]), dtype=torch.int8, device=dev)
    print(f"{shape}: quantize_by_tile {a*1e3:.1f} us ({n*3/a/1e6:.0f} GB/s)   quantize_rows_rowsum {b*1e3:.1f} us ({n*3/b/1e6:.0f} GB/s)", flush=True)
