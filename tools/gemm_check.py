"""Bit-exactness of a GEMM experiment build (FFQ_LIB=...) against the shipped library on the four Llama-3-8B shapes at
T = 16384 (plain bf16 launch with an activation offset, and the gate+up / SiLU / quantize launch)."""
import os, pathlib, sys
import torch
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from fastforward_amd import ops, _native
from fastforward_amd._cabi import FFQLibrary
base = _native.library()
other = FFQLibrary(os.environ["FFQ_LIB"])
T = 16384
ok = True
for n, k in ((4096, 4096), (1024, 4096), (14336, 4096), (4096, 14336)):
    xq = torch.randint(-128, 128, (T, k), device="cuda", dtype=torch.int8)
    wq = torch.randint(-128, 128, (n, k), device="cuda", dtype=torch.int8)
    sx, ox = torch.tensor([0.02], device="cuda"), torch.tensor([4.0], device="cuda")
    sw = torch.rand(n, device="cuda") * 1e-3 + 5e-4
    _native._LIB = base
    want = ops.linear_w8a8(xq, wq, sx, ox, sw, None, out_dtype=torch.bfloat16)
    _native._LIB = other
    got = ops.linear_w8a8(xq, wq, sx, ox, sw, None, out_dtype=torch.bfloat16)
    same = torch.equal(got, want)
    ok &= same
    print(n, k, "plain equal:", same)
    if n == 14336:
        uq = torch.randint(-128, 128, (n, k), device="cuda", dtype=torch.int8)
        so, oo = torch.tensor([0.004], device="cuda"), torch.tensor([-9.0], device="cuda")
        sg = torch.rand(n, device="cuda") * 1e-5 + 2e-5
        _native._LIB = base
        want = ops.mlp_gate_up_w8a8(xq, wq, uq, sx, ox, sg, sg, so, oo, 8)
        _native._LIB = other
        got = ops.mlp_gate_up_w8a8(xq, wq, uq, sx, ox, sg, sg, so, oo, 8)
        same = torch.equal(got, want)
        ok &= same
        print(n, k, "mlp mode equal:", same)
_native._LIB = base
print("CHECK", "OK" if ok else "FAILED")
