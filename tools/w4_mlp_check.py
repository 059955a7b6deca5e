"""Bit-equality of the gate + up + SiLU*up launch on the one-wave-per-SIMD kernel (whole tiles, two-pass form) with the 8-wave kernel
of the same build (ffq_force_generic_kernels) and with the composition of its parts; twice per shape (the K-loop runs across tiles).
usage: [FFQ_LIB=...] python tools/w4_mlp_check.py"""
import os, pathlib, sys
import torch
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from fastforward_amd import ops, _native
if os.environ.get("FFQ_LIB"):
    from fastforward_amd._cabi import FFQLibrary
    _native._LIB = FFQLibrary(os.environ["FFQ_LIB"])
lib = _native.library()
g = torch.Generator(device="cuda").manual_seed(33)
ok = True
for m, n, k, big in ((4096, 14336, 4096, False), (1024, 512, 256, True), (768, 384, 384, False), (2048, 1024, 640, True), (16384, 2048, 512, False), (4096, 1024, 8192, False)):
    x = torch.randn(m, k, device="cuda", generator=g).to(torch.bfloat16) * (30.0 if big else 1.0)  # (big: silu arguments outside the table's window too)
    gc = torch.randint(-128, 128, (n, k), device="cuda", dtype=torch.int8, generator=g)
    uc = torch.randint(-128, 128, (n, k), device="cuda", dtype=torch.int8, generator=g)
    gs = torch.rand(n, device="cuda", generator=g) * 1e-3 + 1e-4
    us = torch.rand(n, device="cuda", generator=g) * 1e-3 + 1e-4
    got = [ops.mlp_gate_up_wq(x, gc, uc, gs, None, us, None, two_pass=True, split=1) for _ in range(2)]
    previous = lib.ffq_force_generic_kernels(1)
    try:
        want = ops.mlp_gate_up_wq(x, gc, uc, gs, None, us, None, two_pass=True, split=1)
        parts = ops.silu_mul_quantize(ops.linear_wq(x, gc, gs, None, two_pass=True, split=1), ops.linear_wq(x, uc, us, None, two_pass=True, split=1), (), want_product=True)
    finally:
        lib.ffq_force_generic_kernels(previous)
    parts = parts[0]
    same = all(torch.equal(t.view(torch.int16), want.view(torch.int16)) for t in got)
    rotated = k >= 8192 and (k & (k - 1)) == 0  # the per-XCD contraction start: another summation order
    if rotated:
        same = torch.equal(got[0].view(torch.int16), got[1].view(torch.int16)) and bool(((got[0].double() - want.double()).abs() <= 2.0**-6 * want.double().abs() + 1e-3 * float(want.double().abs().max())).all())
    comp = torch.equal(want.view(torch.int16), parts.view(torch.int16))
    ok &= same and comp
    print(m, n, k, "4w == 8w:" if not rotated else "4w ~ 8w (rotated order):", same, "| 8w == parts:", comp, flush=True)
print("CHECK", "OK" if ok else "FAILED")
