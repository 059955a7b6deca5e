"""down_proj behind the gate+up launch: does a read pass over its 235 MB of activation codes (just written with non-temporal stores by
the gate+up epilogue) pay for itself? HIP events around the down_proj launch (its weight row-sum launch included) and around the pair."""
import pathlib, statistics, sys
import torch
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from fastforward_amd import ops
dev = "cuda"
T, K, N = 16384, 4096, 14336
torch.manual_seed(0)
xq = torch.randint(-128, 128, (T, K), device=dev, dtype=torch.int8)
gq = torch.randint(-128, 128, (N, K), device=dev, dtype=torch.int8)
uq = torch.randint(-128, 128, (N, K), device=dev, dtype=torch.int8)
dq = torch.randint(-128, 128, (K, N), device=dev, dtype=torch.int8)
sx, ox = torch.tensor([0.02], device=dev), torch.tensor([4.0], device=dev)
sg = torch.rand(N, device=dev) * 0.00002 + 0.00002
sd = torch.rand(K, device=dev) * 0.001 + 0.0005
so, oo = torch.tensor([0.03], device=dev), torch.tensor([-3.0], device=dev)

def run(touch):
    ev = []
    for _ in range(8):
        codes = ops.mlp_gate_up_w8a8(xq, gq, uq, sx, ox, sg, sg, so, oo, 8)
        a, b, c = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        a.record()
        if touch:
            ops.minmax_by_tile(codes.view(torch.bfloat16), (T, N // 2))
        b.record()
        ops.linear_w8a8(codes, dq, so, oo, sd, None)
        c.record()
        ev.append((a, b, c))
    torch.cuda.synchronize()
    return statistics.median(b.elapsed_time(c) for a, b, c in ev[2:]) * 1e3, statistics.median(a.elapsed_time(c) for a, b, c in ev[2:]) * 1e3

for rep in range(3):
    d0, t0 = run(False); d1, t1 = run(True)
    print(f"down_proj alone {d0:7.1f} us (pair {t0:7.1f}) | behind a read pass over its input codes {d1:7.1f} us (touch + down {t1:7.1f})", flush=True)
