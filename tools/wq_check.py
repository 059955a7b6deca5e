"""Quick exactness check of a weight-only GEMM build (FFQ_LIB=...): identity activations -> the dequantized weight."""
import os, pathlib, sys
import torch
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from fastforward_amd import ops, _native
base = _native.library()
lib = base
if os.environ.get("FFQ_LIB"):
    from fastforward_amd._cabi import FFQLibrary
    lib = FFQLibrary(os.environ["FFQ_LIB"])
ok = True
for k, group, bits, offset in ((512, 512, 8, False), (512, 128, 4, True), (1024, 64, 4, False), (4096, 4096, 8, True)):
    g = torch.Generator().manual_seed(k + group)
    n = 320
    codes = torch.randint(-(2 ** (bits - 1)), 2 ** (bits - 1), (n, k), generator=g, dtype=torch.int8).cuda()
    scale = (torch.rand(n * (k // group), generator=g) * 0.02 + 0.002).cuda()
    off = (torch.round(torch.randn(n * (k // group), generator=g) * 3) + 0.25).cuda() if offset else None
    x = torch.eye(k, device="cuda", dtype=torch.bfloat16)
    _native._LIB = lib
    y = ops.linear_wq(x, codes, scale, off, group=group)
    xr = torch.randn(300, k, device="cuda", dtype=torch.bfloat16)
    yr = ops.linear_wq(xr, codes, scale, off, group=group)
    _native._LIB = base
    want = ops.dequantize_by_tile(codes, scale, (1, group), off, torch.bfloat16)
    ref = xr.double() @ want.double().t()
    e1 = torch.equal(y, want.t())
    e2 = bool(((yr.double() - ref).abs() <= 2.0**-8 * ref.abs() + 1e-5 * float(ref.abs().max())).all())
    ok &= e1 and e2
    print(k, group, bits, offset, "identity exact:", e1, "random within tolerance:", e2)
print("CHECK", "OK" if ok else "FAILED")
