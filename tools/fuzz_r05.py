"""Seeded random-shape checks of the round-5 kernels against their compositions on the same device (not a benchmark):
  skinny weight-only linear (both forms, forced splits, packed / int8, grouped, offsets, multi-matrix)  vs  float64 of the same operands
                                                                                                          and packed == int8 bits
  A3 with the symmetric one-sided fallback (guess / settle)                                              vs  the composed form
  estimator step + quantize in one pass                                                                  vs  the two calls
  gate/up while estimating (either / or) and the gated epilogue                                          vs  two linears + SiLU * up
  sibling quantizers the device compares (A1 unless same, the linear on the codes in force, gate/up too)  vs  every quantizer for itself
usage: python tools/fuzz_r05.py [seconds=120] [seed=0]"""
import pathlib, random, sys, time
import torch
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import fastforward_amd as ff
from fastforward_amd import _native, ops

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = random.Random(seed)
DEV = "cuda"
lib = _native.library()
counts = {}


def same(a, b):
    return torch.equal(a.isnan(), b.isnan()) and torch.equal(torch.where(a.isnan(), torch.zeros_like(a), a), torch.where(b.isnan(), torch.zeros_like(b), b))


def skinny():
    m = rng.choice([1, 2, 3, 5, 8, 15, 16, 17, 24, 31, 32, 33, 48, 64, 65, 100, 127, 128])
    k = 128 * rng.randint(2, 40)
    n = rng.choice([16, 48, 100, 128, 130, 256, 1000, 1024, 2048, 4096, 5000])
    g = torch.Generator(device=DEV).manual_seed(rng.randint(0, 1 << 30))
    x = torch.randint(-4, 5, (m, k), device=DEV, generator=g).to(torch.bfloat16)
    w4 = torch.randint(-8, 8, (n, k), device=DEV, dtype=torch.int8, generator=g)
    grouped = rng.random() < 0.6
    groups = k // 128 if grouped else 1
    s = torch.full((n * groups,), 0.25 if rng.random() < 0.5 else 0.5, device=DEV)
    o = torch.round(torch.randn(n * groups, device=DEV, generator=g) * 2) if rng.random() < 0.5 else None
    group = 128 if grouped else k
    wd = (w4.double().view(n, groups, k // groups) + (0 if o is None else o.double().view(n, groups, 1))).view(n, k) * s.double().view(n, groups, 1).expand(n, groups, k // groups).reshape(n, k)
    exact = (x.double() @ wd.t()).to(torch.bfloat16)
    plan = int(lib.ffq_linear_wq_split(m, n, k, 0))
    for split in sorted({0, 1, plan, rng.choice([2, 3, 4])}):
        if split > k // 256:
            continue
        got = ops.linear_wq(x, w4, s, o, group=group, split=split)
        assert got is not None and torch.equal(got, exact), ("skinny int8", m, n, k, grouped, o is not None, split)
    if grouped:
        packed = ops.pack_int4(w4, block=128)
        got = ops.linear_wq(x, packed, s, o, group=128, pack_block=128)
        assert torch.equal(got, exact), ("skinny packed", m, n, k)
    if rng.random() < 0.3 and not grouped and n % 256 == 0:
        ws = [w4, w4.flip(0).contiguous()]
        outs = ops.linear_wq_multi(x, ws, [s, s], [o, o])
        if outs is not None:
            assert torch.equal(outs[0], exact), ("skinny multi", m, n, k)


def wide():
    """Round 6: the weight-only linear above the skinny form's rows — 128-column tiles (17 .. 512 rows), 256-row tiles with split tails,
    the one-wave-per-SIMD kernel (whole tiles, two-pass) and its gate + up + SiLU*up mode — on operands whose sums are exact in fp32
    whatever the order (small integers, power-of-two scales): every form must give the SAME bits."""
    m = rng.choice([17, 40, 129, 200, 256, 300, 512, 513, 768, 1000, 1024, 1280, 2048, 4096])
    k = 128 * rng.randint(2, 40)
    n = rng.choice([128, 256, 384, 512, 1024, 2048, 4096])
    g = torch.Generator(device=DEV).manual_seed(rng.randint(0, 1 << 30))
    x = torch.randint(-4, 5, (m, k), device=DEV, generator=g).to(torch.bfloat16)
    w4 = torch.randint(-8, 8, (n, k), device=DEV, dtype=torch.int8, generator=g)
    grouped = rng.random() < 0.4
    groups = k // 128 if grouped else 1
    s = torch.full((n * groups,), 0.25 if rng.random() < 0.5 else 0.5, device=DEV)
    o = torch.round(torch.randn(n * groups, device=DEV, generator=g) * 2) if rng.random() < 0.3 else None
    group = 128 if grouped else k
    wd = (w4.double().view(n, groups, k // groups) + (0 if o is None else o.double().view(n, groups, 1))).view(n, k) * s.double().view(n, groups, 1).expand(n, groups, k // groups).reshape(n, k)
    exact = (x.double() @ wd.t()).to(torch.bfloat16)
    for two_pass in (None, False, True):
        got = ops.linear_wq(x, w4, s, o, group=group, two_pass=two_pass)
        assert got is not None and torch.equal(got, exact), ("wide int8", m, n, k, grouped, o is not None, two_pass)
    if grouped:
        got = ops.linear_wq(x, ops.pack_int4(w4, block=128), s, o, group=128, pack_block=128)
        assert torch.equal(got, exact), ("wide packed", m, n, k)
    if not grouped and rng.random() < 0.5:
        u4 = torch.randint(-8, 8, (n, k), device=DEV, dtype=torch.int8, generator=g)
        ud = (u4.double() + (0 if o is None else o.double().view(n, 1))) * s.double().view(n, 1)
        up = (x.double() @ ud.t()).to(torch.bfloat16)
        want = ops.silu_mul_quantize(exact, up, (), want_product=True)[0]
        for two_pass in (None, True):
            got = ops.mlp_gate_up_wq(x, w4, u4, s, o, s, o, two_pass=two_pass)
            assert got is not None and torch.equal(got.view(torch.int16), want.view(torch.int16)), ("wide mlp", m, n, k, two_pass)


def a3_symmetric():
    rows = rng.choice([1, 3, 33, 100, 257, 1000])
    cols = 16 * rng.randint(1, 600)
    tile = rng.choice([cols, 16, 32, 128]) if cols % 128 == 0 else cols
    if cols % tile:
        tile = cols
    dtype = rng.choice([torch.bfloat16, torch.float32])
    g = torch.Generator(device=DEV).manual_seed(rng.randint(0, 1 << 30))
    x = (torch.randn(rows, cols, device=DEV, generator=g) * 3).to(dtype)
    kind = rng.choice(["mixed", "abs", "abs_one_negative", "nan"])
    if kind != "mixed":
        x = x.abs()
    if kind == "abs_one_negative":
        x.view(-1)[rng.randrange(x.numel())] = -1.0
    if kind == "nan":
        x.view(-1)[rng.randrange(x.numel())] = float("nan")
    got = ops.quantize_dynamic_by_tile(x, (1, tile), 8, True, True, torch.int8)
    prev = lib.ffq_force_generic_kernels(1)
    try:
        want = ops.quantize_dynamic_by_tile(x, (1, tile), 8, True, True, torch.int8)
    finally:
        lib.ffq_force_generic_kernels(prev)
    for a, b in zip(got, want):
        assert same(a.float(), b.float()), ("a3 symmetric", rows, cols, tile, dtype, kind)


def running():
    rows = rng.choice([2, 33, 100, 512])
    cols = 16 * rng.randint(1, 900)
    dtype = rng.choice([torch.bfloat16, torch.float32])
    symmetric, one_sided = rng.choice([(True, True), (True, False), (False, True)])
    g = torch.Generator(device=DEV).manual_seed(rng.randint(0, 1 << 30))
    state = {}
    for route in ("fused", "two"):
        lo = torch.full((rows,), float("inf"), device=DEV, dtype=dtype)
        hi = torch.full((rows,), float("-inf"), device=DEV, dtype=dtype)
        status = torch.zeros(1, dtype=torch.int32, device=DEV)
        scale, offset = torch.ones(rows, device=DEV), torch.zeros(rows, device=DEV)
        gg = torch.Generator(device=DEV).manual_seed(7)
        outs = []
        for step in range(3):
            x = (torch.randn(rows, cols, device=DEV, generator=gg) * (step + 1)).to(dtype)
            if step < 2:
                x = x.abs()
            if route == "fused":
                codes = ops.running_minmax_quantize(x, (1, cols), lo, hi, status, 8, symmetric, one_sided, scale, offset, torch.int8)
                if codes is None:
                    return
            else:
                ops.running_minmax_step(x, (1, cols), lo, hi, status, 8, symmetric, one_sided, scale, offset)
                codes = ops.quantize_by_tile(x, scale, (1, cols), 8, torch.int8, offset)
            outs += [codes.clone(), scale.clone(), offset.clone(), lo.float().clone(), hi.float().clone(), status.clone()]
        state[route] = outs
    for a, b in zip(state["fused"], state["two"]):
        assert same(a.float(), b.float()), ("running", rows, cols, dtype, symmetric, one_sided)


def either_or():
    m = 256 * rng.randint(8, 12) - rng.choice([0, 0, 56])
    n = 128 * rng.randint(8, 20)
    k = 128 * rng.randint(2, 8)
    if ((m + 255) // 256) * ((n + 255) // 256) < 64:
        return
    g = torch.Generator(device=DEV).manual_seed(rng.randint(0, 1 << 30))
    x = (torch.randn(m, k, device=DEV, generator=g) * 2).to(torch.bfloat16)
    sg, og = torch.tensor([0.03], device=DEV), torch.tensor([float(rng.randint(-5, 5))], device=DEV)
    su, ou = (sg.clone(), og.clone()) if rng.random() < 0.6 else (torch.tensor([0.05], device=DEV), og.clone())
    xg = ops.quantize_by_tile(x, sg, x.shape, 8, torch.int8, og)
    xu = ops.quantize_by_tile(x, su, x.shape, 8, torch.int8, ou)
    wg = torch.randint(-128, 128, (n, k), device=DEV, dtype=torch.int8, generator=g)
    wu = torch.randint(-128, 128, (n, k), device=DEV, dtype=torch.int8, generator=g)
    swg = torch.rand(n, device=DEV, generator=g) * 1e-3 + 1e-4
    swu = torch.rand(n, device=DEV, generator=g) * 1e-3 + 1e-4
    owg = owu = None
    r = rng.random()
    if r < 0.3:
        owg, owu = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    elif r < 0.5:
        owg, owu = torch.zeros(n, device=DEV), torch.round(torch.randn(n, device=DEV, generator=g))
    gate = ops.linear_w8a8(xg, wg, sg, og, swg, owg, None, out_dtype=torch.bfloat16)
    up = ops.linear_w8a8(xu, wu, su, ou, swu, owu, None, out_dtype=torch.bfloat16)
    want = ops.silu_mul_quantize(gate, up, (), want_product=True)[0]
    lo, hi = ops.minmax_by_tile(want, want.shape)
    out = ops.mlp_gate_up_w8a8_estimating(xg, xu, wg, wu, (sg, og), (su, ou), (swg, owg), (swu, owu), want_extrema=True)
    if out is None:
        return
    assert torch.equal(out[0].view(torch.int16), want.view(torch.int16)), ("either/or", m, n, k)
    assert torch.equal(out[1].view(torch.int16), torch.cat([lo, hi]).view(torch.int16)), ("either/or extrema", m, n, k)
    if n % 64 == 0:
        gated = ops.linear_w8a8_gated(xu, wu, su, ou, swu, owu, gate, want_extrema=True)
        if gated is not None:
            assert torch.equal(gated[0].view(torch.int16), want.view(torch.int16)), ("gated", m, n, k)
            assert torch.equal(gated[1].view(torch.int16), torch.cat([lo, hi]).view(torch.int16)), ("gated extrema", m, n, k)


def siblings():
    m = 256 * rng.randint(2, 12) - rng.choice([0, 0, 56, 129])
    n = 128 * rng.randint(4, 20)
    k = 128 * rng.randint(2, 8)
    g = torch.Generator(device=DEV).manual_seed(rng.randint(0, 1 << 30))
    dtype = rng.choice([torch.bfloat16, torch.float16, torch.float32])
    x = (torch.randn(m, k, device=DEV, generator=g) * 2).to(dtype)
    t = lambda v: None if v is None else torch.tensor([v], device=DEV, dtype=torch.float32)  # noqa: E731
    es, eo = rng.choice([0.03, 0.011]), rng.choice([None, 0.0, -2.6, 3.5, 0.3])
    kind = rng.choice(["same", "same", "rounds_alike", "scale", "offset"])
    s, o = es, eo
    if kind == "rounds_alike":
        o = (0.0 if eo is None else round(eo)) + rng.choice([-0.4, 0.2, 0.45])
    if kind == "scale":
        s = es * 1.25
    if kind == "offset":
        o = (0.0 if eo is None else eo) + rng.choice([1.0, -2.0, 7.0])
    scale, offset, e_scale, e_offset = t(s), t(o), t(es), t(eo)
    first = ops.quantize_by_tile(x, e_scale, x.shape, 8, torch.int8, e_offset)
    own = ops.quantize_by_tile(x, scale, x.shape, 8, torch.int8, offset)
    maybe = torch.full(x.shape, 77, dtype=torch.int8, device=DEV)
    import ctypes
    p = lambda u: ctypes.c_void_p(None if u is None else u.data_ptr())  # noqa: E731
    lib.check(lib.ffq_quantize_by_tile_unless_same(p(x), ops._tag(x.dtype), p(scale), p(offset), x.numel(), 8.0, p(e_scale), p(e_offset), p(maybe),
                                                   torch.cuda.current_stream().cuda_stream))
    same_params = s == es and round(0.0 if o is None else o) == round(0.0 if eo is None else eo)  # (Python rounds half to even too)
    assert bool((maybe == 77).all()) if same_params else torch.equal(maybe, own), ("unless same", kind, m, k, dtype, s, o, es, eo)
    if same_params:
        assert torch.equal(own, first), ("same parameters, other codes", kind, s, o, es, eo)
    if dtype != torch.bfloat16:
        return
    wq = torch.randint(-128, 128, (n, k), device=DEV, dtype=torch.int8, generator=g)
    sw = torch.rand(n, device=DEV, generator=g) * 1e-3 + 1e-4
    ow = rng.choice([None, torch.zeros(n, device=DEV), torch.round(torch.randn(n, device=DEV, generator=g))])
    got = ops.linear_w8a8_earlier(maybe, (first, e_scale, e_offset), wq, scale, offset, sw, ow, out_dtype=torch.bfloat16)
    if got is None:
        assert not ops.linear_w8a8_takes_earlier(m, n, k)
        return
    want = ops.linear_w8a8(own, wq, scale, offset, sw, ow, None, out_dtype=torch.bfloat16)
    assert torch.equal(got.view(torch.int16), want.view(torch.int16)), ("earlier", kind, m, n, k)
    wg = torch.randint(-128, 128, (n, k), device=DEV, dtype=torch.int8, generator=g)
    swg = torch.rand(n, device=DEV, generator=g) * 1e-3 + 1e-4
    owg = None if ow is None else torch.zeros(n, device=DEV)
    e_off = e_offset if e_offset is not None else torch.zeros(1, device=DEV)
    gate = ops.linear_w8a8(first, wg, e_scale, e_off, swg, owg, None, out_dtype=torch.bfloat16)
    product = ops.silu_mul_quantize(gate, want, (), want_product=True)[0]
    out = ops.mlp_gate_up_w8a8_estimating(first, maybe, wg, wq, (e_scale, e_off), (scale, offset if offset is not None else torch.zeros(1, device=DEV)), (swg, owg), (sw, ow))
    if out is not None:
        assert torch.equal(out.view(torch.int16), product.view(torch.int16)), ("either/or on undecided codes", kind, m, n, k)


cases = [skinny, skinny, wide, wide, a3_symmetric, running, either_or, siblings]
t0 = time.time()
while time.time() - t0 < budget:
    fn = rng.choice(cases)
    fn()
    counts[fn.__name__] = counts.get(fn.__name__, 0) + 1
torch.cuda.synchronize()
for buf in ops._TICKETS.values():
    assert int(buf.abs().sum()) == 0, "ticket words not zero"
for words in ops._EXTREMA_WORDS.values():
    assert words.tolist() == [-1, 0, 0, 0], "extrema words not in their initial state"
print("ok", counts, f"seed {seed}", flush=True)
