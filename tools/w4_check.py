"""Bit-equality of the one-wave-per-SIMD bf16-image GEMM of a library build (FFQ_LIB=...) with the 8-wave kernel of the same build
(ffq_force_generic_kernels), on the shapes of tests/test_gemm_gpu.py::test_one_wave_per_simd_form_equals_the_eight_wave_kernel, twice
per shape (the K-loop runs across tile boundaries: a stale slot shows on the second tile)."""
import os, pathlib, sys
import torch
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from fastforward_amd import ops, _native
if os.environ.get("FFQ_LIB"):
    from fastforward_amd._cabi import FFQLibrary
    _native._LIB = FFQLibrary(os.environ["FFQ_LIB"])
lib = _native.library()
g = torch.Generator(device="cuda").manual_seed(21)
ok = True
for m, n, k in ((4096, 4096, 4096), (1024, 768, 256), (768, 256, 384), (2048, 1024, 640), (8192, 2048, 512), (16384, 1024, 256), (16384, 4096, 14336), (4096, 4096, 8192), (2048, 2048, 128 * 3 * 2)):
    x = torch.randn(m, k, device="cuda", generator=g).to(torch.bfloat16)
    w = torch.randint(-128, 128, (n, k), device="cuda", dtype=torch.int8, generator=g)
    s = torch.rand(n, device="cuda", generator=g) * 1e-3 + 1e-4
    got = [ops.linear_wq(x, w, s, None, two_pass=True, split=1) for _ in range(2)]
    previous = lib.ffq_force_generic_kernels(1)
    try:
        want = ops.linear_wq(x, w, s, None, two_pass=True, split=1)
    finally:
        lib.ffq_force_generic_kernels(previous)
    same = all(torch.equal(t, want) for t in got)
    if k >= 8192 and (k & (k - 1)) == 0:  # the per-XCD contraction start: another summation order, float64 tolerance instead
        ref = x.double() @ ops.dequantize_by_tile(w, s, (1, k), None, torch.bfloat16).double().t()
        same = all(bool(((t.double() - ref).abs() <= 2.0**-7 * ref.abs() + 1e-4 * float(ref.abs().max())).all()) for t in got) and torch.equal(got[0], got[1])
    ok &= same
    print(m, n, k, "equal" if same else "DIFFERENT", flush=True)
print("CHECK", "OK" if ok else "FAILED")
