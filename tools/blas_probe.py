"""What the vendor GEMMs reach on the forward's shapes (reference points, not product paths): hipBLASLt bf16 (F.linear) and
int8 (torch._int_mm) at T = 16384, hipGraph-replayed HIP-event timing; run under rocprofv3 --kernel-trace to see kernel names."""
import pathlib, sys
import torch
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from bench import event_time_ms
T = 16384
for name, n, k in (("qo", 4096, 4096), ("gateup", 14336, 4096), ("down", 4096, 14336)):
    xq = torch.randint(-128, 128, (T, k), device="cuda", dtype=torch.int8)
    wq = torch.randint(-128, 128, (n, k), device="cuda", dtype=torch.int8)
    x = torch.randn(T, k, device="cuda", dtype=torch.bfloat16)
    w = torch.randn(n, k, device="cuda", dtype=torch.bfloat16)
    row = [name]
    try:
        wt = wq.t()  # [K, N] column-major view: _int_mm wants mat2 like this
        ms = event_time_ms(lambda r: torch._int_mm(xq, wt), iters=5, reps=4)
        row.append(f"int8 _int_mm {ms:.4f} ms {2*T*n*k/ms/1e9:.0f} TOP/s")
    except Exception as e:  # noqa: BLE001
        row.append(f"int8 _int_mm unavailable: {type(e).__name__}: {str(e)[:80]}")
    ms = event_time_ms(lambda r: torch.nn.functional.linear(x, w), iters=5, reps=4)
    row.append(f"bf16 linear {ms:.4f} ms {2*T*n*k/ms/1e9:.0f} TF")
    print("  ".join(row), flush=True)
