"""Generate the golden fixtures in this directory by running the REFERENCE itself (CPU eager).

Run only in the build container, where /root/reference exists:

    mkdir -p /tmp/ffshim/optree && echo "from torch.utils._pytree import tree_map, tree_flatten, \
tree_unflatten, tree_leaves" > /tmp/ffshim/optree/__init__.py
    PYTHONPATH=/root/reference/src:/tmp/ffshim python tests/golden/gen_golden.py

(`optree` is the one eagerly imported dependency of the reference that is not installed here; the
two-line shim re-exports torch's own pytree functions and lives outside the repository. G17 additionally needs
/tmp/ffshim/ffstub.py — a meta-path finder that fabricates empty `gguf` / `onnx*` packages so that
`fastforward.export.stages.gguf` imports; the packing functions it then calls are the reference's own, pure torch.)

Nothing of the reference travels: the fixtures hold inputs (or the seed that regenerates them with
torch's CPU generator) and the outputs the reference produced. Files are torch.save'd dicts of
tensors and plain Python values, loadable with ``torch.load(..., weights_only=True)``.
"""

from __future__ import annotations

import itertools
import pathlib
import sys

import torch

HERE = pathlib.Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
from datagen import make_data  # noqa: E402

try:
    import fastforward as ff

    from fastforward.quantization import affine
    from fastforward.quantization.affine import parameters_for_range
    from fastforward.quantization.tiled_tensor import tiles_to_rows
except ImportError as e:  # pragma: no cover
    sys.exit(f"the reference is not importable ({e}); see the module docstring")


def gran_of(spec):
    kind = spec[0]
    if kind == "tensor":
        return ff.PerTensor()
    if kind == "channel":
        return ff.PerChannel(spec[1])
    if kind == "block":
        return ff.PerBlock(block_dims=spec[1], block_sizes=spec[2], per_channel_dims=spec[3])
    if kind == "tile":
        return ff.PerTile(tuple(spec[1]))
    raise ValueError(spec)


def tile_of(spec, shape):
    tile = gran_of(spec).tile_size(torch.Size(shape))
    return list(shape) if isinstance(tile, str) else list(tile)


def params_from_range(x, spec, num_bits, mode):
    """(scale, offset) the way LinearQuantizer derives them from the per-tile min/max of x."""
    tile = tile_of(spec, x.shape)
    rows = tiles_to_rows(x.float(), tile)
    lo, hi = rows.min(-1).values, rows.max(-1).values
    symmetric = mode != "asymmetric"
    scale, offset = parameters_for_range(lo, hi, num_bits, symmetric=symmetric, allow_one_sided=True)
    return scale, offset


def static_case(name, x, spec, num_bits, scale, offset, quantized_dtype):
    q = affine.quantize_per_granularity(x, scale, offset, gran_of(spec), num_bits, quantized_dtype)
    return {
        "name": name,
        "granularity": list(spec),
        "tile": tile_of(spec, x.shape),
        "shape": list(x.shape),
        "num_bits": num_bits,
        "scale": None if scale is None else scale.clone(),
        "offset": None if offset is None else offset.clone(),
        "quantized_dtype": str(quantized_dtype).removeprefix("torch.") if quantized_dtype else None,
        "codes": q.raw_data.clone(),
        "dequantized": q.dequantize().clone(),
    }


def g1_known_answers():
    """Known-answer vectors of the reference's own tests (tests/nn/test_linear_quantizer.py)."""
    out = {}
    # :20-72 — linspace(-8, 8, 17), scale 2, 2 bits, no offset
    data = torch.linspace(-8, 8, 17)
    params = affine.StaticAffineQuantParams(scale=torch.tensor(2.0), offset=None, granularity=ff.PerTensor(), num_bits=2)
    q = affine.AffineQuantizationFunction.quantize(data, params)
    out["symmetric_2bit"] = {
        "data": data, "scale": torch.tensor(2.0), "offset": None, "num_bits": 2,
        "codes": q.raw_data.clone(), "dequantized": q.dequantize().clone(),
        "expected_codes_in_reference_test": torch.tensor([-2.0] * 6 + [-1.0, 0.0, 0.0, 0.0] + [1.0] * 7),
    }
    # :189-219 — asymmetric per tensor, scale 2 offset 4
    data = torch.stack([torch.linspace(-2, 14, 17)] * 32)
    quantizer = ff.nn.LinearQuantizer(num_bits=2, symmetric=False)
    quantizer.quantization_range = (data.min(), data.max())
    with torch.no_grad():
        quantizer.scale.fill_(2.0)
        quantizer.offset.fill_(4.0)
    q = quantizer(data)
    out["asymmetric_2bit"] = {
        "data": data, "scale": torch.tensor([2.0]), "offset": torch.tensor([4.0]), "num_bits": 2,
        "codes": q.raw_data.clone(), "dequantized": q.dequantize().clone(),
        "expected_row_in_reference_test": torch.tensor([-2.0] * 8 + [-1.0, 0.0, 0.0, 0.0] + [1.0] * 5),
    }
    # :400-418 — one-sided data: 4-bit symmetric quantizer picks offset 8
    data = torch.rand(64, generator=torch.Generator().manual_seed(7))
    quantizer = ff.nn.LinearQuantizer(num_bits=4, symmetric=True, allow_one_sided=True)
    quantizer.quantization_range = (data.min(), data.max())
    q = quantizer(data)
    out["one_sided_4bit"] = {
        "data": data, "range_min": data.min(), "range_max": data.max(), "num_bits": 4,
        "scale": quantizer.scale.detach().clone(), "offset": quantizer.offset.detach().clone(),
        "codes": q.raw_data.clone(), "dequantized": q.dequantize().clone(),
    }
    return out


def g2_edges():
    """Ties, clamping edges, NaN, +-Inf, -0.0 with scale 1 and half-even rounding of the offset."""
    vals = [-2.5, -1.5, -0.5, 0.5, 1.5, 2.5, 126.5, 127.5, 128.5, -127.5, -128.5, -129.0,
            float("nan"), float("inf"), float("-inf"), -0.0, 0.0, 3.4e38, -3.4e38, 1e-45]
    cases = []
    for dtype, qdt, off in itertools.product(
        (torch.float32, torch.bfloat16), (None, torch.int8, torch.int32), (None, 0.5, 1.5, -2.5)
    ):
        x = torch.tensor(vals, dtype=dtype)
        offset = None if off is None else torch.tensor([off])
        name = f"edge_{str(dtype).removeprefix('torch.')}_{qdt}_{off}"
        cases.append(static_case(name, x, ("tensor",), 8, torch.tensor([1.0]), offset, qdt) | {"data": x})
    # zero and denormal scales (0/0 -> NaN, x/0 -> +-Inf -> clamp)
    x = torch.tensor([0.0, -0.0, 1.0, -1.0, 1e-40, -1e-40], dtype=torch.float32)
    for s in (0.0, 1e-42, 3.0e38):
        cases.append(static_case(f"edge_scale_{s}", x, ("tensor",), 8, torch.tensor([s]), None, None) | {"data": x})
    return cases


G3_GRANULARITIES = [
    ("tensor",),
    ("channel", (0,)),
    ("channel", (1,)),
    ("channel", (-1,)),
    ("channel", (0, 2)),
    ("block", (2,), (4,), (0,)),
    ("tile", (16, 8, 4)),
    ("tile", (1, 1, 1)),
]


def seeded_case(name, seed, shape, dtype, kind, spec, num_bits, mode, keep_dequantized=True):
    """A case whose input is regenerated from `seed`; codes stored as int8, dequantized in the data dtype."""
    x = make_data(seed, shape, dtype, kind)
    scale, offset = params_from_range(x, spec, num_bits, mode)
    c = static_case(name, x, spec, num_bits, scale, offset, None)
    # the int8 container must hold the same integers as the default (data dtype) container
    c8 = static_case(name, x, spec, num_bits, scale, offset, torch.int8)
    assert torch.equal(c["codes"].to(torch.int8), c8["codes"]) or torch.isnan(c["codes"].float()).any()
    assert torch.equal(c["dequantized"], c8["dequantized"]) or torch.isnan(c["dequantized"].float()).any()
    c["codes"] = c8["codes"]
    if not keep_dequantized:
        c["dequantized"] = None
    c |= {"seed": seed, "kind": kind, "dtype": str(dtype).removeprefix("torch."), "mode": mode}
    del c["quantized_dtype"]
    return c


def g3_sweeps():
    """Random sweeps on the reference's own test shape (32, 16, 8) and two 2-D shapes."""
    cases = []
    seed = 1000
    for spec, num_bits, dtype, mode in itertools.product(G3_GRANULARITIES, (2, 4, 8), (torch.float32, torch.bfloat16), ("symmetric", "asymmetric")):
        for kind in ("normal", "outlier") if num_bits == 8 else ("normal",):
            seed += 1
            name = f"sweep3d_{'_'.join(map(str, spec))}_{num_bits}b_{str(dtype).removeprefix('torch.')}_{mode}_{kind}"
            cases.append(seeded_case(name, seed, (32, 16, 8), dtype, kind, spec, num_bits, mode))
    for spec, dtype in itertools.product((("tensor",), ("channel", (0,))), (torch.float32, torch.bfloat16)):
        seed += 1  # all-positive data: the symmetric quantizer switches to the one-sided grid
        cases.append(seeded_case(f"sweep3d_onesided_{spec[0]}_{dtype}", seed, (32, 16, 8), dtype, "positive", spec, 4, "symmetric"))
    specs2d = [(("tensor",), 8, "asymmetric"), (("channel", (0,)), 8, "symmetric"), (("channel", (-1,)), 8, "symmetric"),
               (("block", (1,), (128,), (0,)), 4, "symmetric"), (("channel", (0,)), 4, "asymmetric")]
    for (spec, num_bits, mode), dtype in itertools.product(specs2d, (torch.float32, torch.bfloat16)):
        seed += 1
        cases.append(seeded_case(f"sweep2d_64x512_{'_'.join(map(str, spec))}_{num_bits}b_{dtype}", seed, (64, 512), dtype, "normal", spec, num_bits, mode))
    for spec, num_bits, mode in specs2d[:4]:  # one Llama-shaped slab, bf16, codes only
        seed += 1
        cases.append(seeded_case(f"sweep2d_128x4096_{'_'.join(map(str, spec))}_{num_bits}b", seed, (128, 4096), torch.bfloat16, "normal", spec, num_bits, mode, keep_dequantized=False))
    # config 1 of BASELINE.json: nn.Linear(1024, 1024) fp32 weight, 8-bit per-tensor
    seed += 1
    cases.append(seeded_case("cfg1_linear1024_weight", seed, (1024, 1024), torch.float32, "normal", ("tensor",), 8, "symmetric", keep_dequantized=False))
    return cases


def g3_dtype_sweep():
    """Mixed dtypes of data / scale / offset / container (the generic kernel's territory).

    Mirrors _quantize_per_element_impl, tests/quantization/test_tiled_affine.py:309-365, with the
    data 5 % away from rounding boundaries (gen_data, :283-306) — plus unconstrained random data so
    that the per-op roundings of half-precision parameters are actually exercised.
    """
    cases = []
    seed = 5000
    data_dtypes = (torch.float16, torch.bfloat16, torch.float32, torch.int32, torch.int16, torch.int8)
    scale_dtypes = (torch.float32, torch.float16, torch.bfloat16)
    offset_dtypes = (None, torch.float32, torch.float16, torch.bfloat16, torch.int32, torch.int8)
    out_dtypes = (torch.int32, torch.int16, torch.float32, torch.float16)
    for ddt, sdt, odt, qdt in itertools.product(data_dtypes, scale_dtypes, offset_dtypes, out_dtypes):
        seed += 1
        g = torch.Generator().manual_seed(seed)
        x = torch.randn(8, 6, 4, generator=g) * 3
        x = (x * 10).to(ddt) if not ddt.is_floating_point else x.to(ddt)
        spec = ("channel", (1,))
        scale = (torch.rand(6, generator=g) * 0.5 + 0.2).to(sdt)
        offset = None if odt is None else (torch.randn(6, generator=g) * 2 + 0.3).to(odt)
        try:
            c = static_case(f"dtypes_{ddt}_{sdt}_{odt}_{qdt}", x, spec, 4, scale, offset, qdt)
        except Exception as e:  # a combination the reference itself cannot run on CPU
            print(f"skip {ddt} {sdt} {odt} {qdt}: {type(e).__name__}: {str(e)[:60]}")
            continue
        cases.append(c | {"data": x})
    return cases


def g4_ranges():
    cases = []
    g = torch.Generator().manual_seed(42)
    for num_bits, symmetric, one_sided, kind, n in itertools.product((2, 4, 8), (True, False), (True, False), ("mixed", "positive", "degenerate", "zero"), (1, 7)):
        lo = torch.randn(n, generator=g) - 0.5
        hi = lo + torch.rand(n, generator=g) * 3
        if kind == "positive":
            lo = lo.abs()
            hi = lo + torch.rand(n, generator=g) * 3
        elif kind == "degenerate":
            hi = lo.clone()
        elif kind == "zero":
            lo = torch.zeros(n)
            hi = torch.zeros(n)
        scale, offset = parameters_for_range(lo, hi, num_bits, symmetric=symmetric, allow_one_sided=one_sided)
        cases.append({"min": lo, "max": hi, "num_bits": num_bits, "symmetric": symmetric, "allow_one_sided": one_sided,
                      "scale": scale.clone(), "offset": None if offset is None else offset.clone()})
    # tests/quantization/affine/test_range.py:17-40 — bf16 inputs give the same parameters as fp32
    lo16, hi16 = torch.tensor([-1.0], dtype=torch.bfloat16), torch.tensor([1.0], dtype=torch.bfloat16)
    for num_bits in (4, 8, 16):
        s, o = parameters_for_range(lo16, hi16, num_bits, symmetric=False, allow_one_sided=False)
        cases.append({"min": lo16, "max": hi16, "num_bits": num_bits, "symmetric": False, "allow_one_sided": False, "scale": s.clone(), "offset": o.clone()})
    return cases


def g5_minmax():
    """Five scaled batches as in tests/range_setting/test_minmax.py:44-61, plus bf16 and per-channel."""
    out = []
    for spec, dtype, symmetric in itertools.product((("tensor",), ("channel", (0,)), ("channel", (-1,))), (torch.float32, torch.bfloat16), (True, False)):
        g = torch.Generator().manual_seed(99)
        base = torch.randn(16, 24, generator=g)
        batches = [(base * (i + 1) / 3).to(dtype) for i in range(5)]
        quantizer = ff.nn.LinearQuantizer(4, symmetric=symmetric, granularity=gran_of(spec))
        model = torch.nn.ModuleList([quantizer])
        outputs = []
        with ff.estimate_ranges(model, ff.range_setting.running_minmax):
            for b in batches:
                outputs.append(quantizer(b).raw_data.clone())
        out.append({
            "granularity": list(spec), "symmetric": symmetric, "num_bits": 4, "batches": batches,
            "scale": quantizer.scale.detach().clone(),
            "offset": None if quantizer.offset is None else quantizer.offset.detach().clone(),
            "codes_per_step": outputs,
        })
    return out


def g6_linear():
    """QuantizedLinear W8A8 forward through the reference's fallback.linear."""
    torch.manual_seed(1234)
    cases = []
    for dtype, bias in ((torch.bfloat16, True), (torch.float32, False)):
        lin = torch.nn.Linear(256, 192, bias=bias).to(dtype)
        x = torch.randn(2, 48, 256).to(dtype)
        model = torch.nn.Sequential(lin)
        ff.quantize_model(model)
        lin.weight_quantizer = ff.nn.LinearQuantizer(8, granularity=ff.PerChannel(0))
        lin.input_quantizer = ff.nn.LinearQuantizer(8, symmetric=False)
        with ff.strict_quantization(False):
            with ff.estimate_ranges(model, ff.range_setting.running_minmax):
                model(x)
            y = model(x)
            xq = lin.input_quantizer(x)
            wq = lin.weight_quantizer(lin.weight)
        y64 = torch.nn.functional.linear(xq.dequantize().double(), wq.dequantize().double(), None if lin.bias is None else lin.bias.double())
        cases.append({
            "x": x, "weight": lin.weight.detach().clone(), "bias": None if lin.bias is None else lin.bias.detach().clone(),
            "x_scale": lin.input_quantizer.scale.detach().clone(), "x_offset": lin.input_quantizer.offset.detach().clone(),
            "w_scale": lin.weight_quantizer.scale.detach().clone(), "w_offset": lin.weight_quantizer.offset.detach().clone(),
            "x_codes": xq.raw_data.to(torch.int8), "w_codes": wq.raw_data.to(torch.int8),
            "y": y.detach().clone(), "y_float64": y64.float(),
        })
    return cases


def g7_tiny_llama():
    """End-to-end: the reference's Llama W8A8 recipe on a 2-layer fp32 model (CPU eager).

    Follows docs/examples/quick_start_quantize_llms.nb.py: quantize_model (:140), strict off (:145),
    weight quantizers LinearQuantizer(8, PerChannel()) on self_attn/* and mlp/* (:159-161),
    input quantizers LinearQuantizer(8, symmetric=False, PerTensor()) on every nn.Linear of the layers
    (:227-232), RunningMinMax calibration (:193,255). Stored: the weights (bf16-representable fp32),
    token ids, all 28 quantizers' (scale, offset), the int8 codes each input quantizer produced in the
    final forward, and the logits.
    """
    sys.path.insert(0, "/root/reference/docs/examples")
    from doc_helpers import quantized_llama  # noqa: F401  (registers the quantized Llama modules)
    from transformers import LlamaConfig, LlamaForCausalLM

    cfg = LlamaConfig(
        hidden_size=256, intermediate_size=896, num_hidden_layers=2, num_attention_heads=8, num_key_value_heads=2,
        vocab_size=512, max_position_embeddings=128, rms_norm_eps=1e-5, rope_theta=500000.0,
        attn_implementation="eager", tie_word_embeddings=False, attention_bias=False, mlp_bias=False,
    )
    torch.manual_seed(1240)
    model = LlamaForCausalLM(cfg).eval()
    with torch.no_grad():
        for p_ in model.parameters():
            if p_.dim() >= 2:
                p_.normal_(0.0, 0.08)
            p_.copy_(p_.to(torch.bfloat16).float())  # store as bf16, compute in fp32
    weights = {k: v.detach().to(torch.bfloat16) for k, v in model.state_dict().items() if "rotary" not in k}
    g = torch.Generator().manual_seed(1241)
    calib = [torch.randint(0, cfg.vocab_size, (4, 64), generator=g) for _ in range(2)]
    ids = torch.randint(0, cfg.vocab_size, (4, 64), generator=g)

    ff.quantize_model(model)
    ff.set_strict_quantization(False)
    w_quantizers = ff.find_quantizers(model, "**/layers/*/self_attn/*/[quantizer:parameter/weight]")
    w_quantizers |= ff.find_quantizers(model, "**/layers/*/mlp/*/[quantizer:parameter/weight]")
    w_quantizers.initialize(ff.nn.LinearQuantizer, num_bits=8, granularity=ff.PerChannel())
    a_quantizers = ff.find_quantizers(model, "**/layers/**/[cls:torch.nn.Linear]/[quantizer:activation/input]")
    a_quantizers.initialize(ff.nn.LinearQuantizer, num_bits=8, symmetric=False, granularity=ff.PerTensor())
    with torch.no_grad(), ff.estimate_ranges(model, ff.range_setting.running_minmax):
        for batch in calib:
            model(batch)
    named = dict(ff.nn.quantized_module.named_quantizers(model))
    assert len(named) == 28, len(named)
    codes = {}
    hooks = []
    for name, q in named.items():
        if name.endswith("input_quantizer"):
            hooks.append(q.register_forward_hook(lambda mod, inp, out, name=name: codes.__setitem__(name, out.raw_data.to(torch.int8).clone())))
    with torch.no_grad():
        logits = model(ids).logits
    for h in hooks:
        h.remove()
    ff.set_strict_quantization(True)
    return {
        "config": {"hidden_size": 256, "intermediate_size": 896, "num_layers": 2, "num_heads": 8, "num_kv_heads": 2,
                   "vocab_size": 512, "rope_theta": 500000.0, "rms_norm_eps": 1e-5},
        "weights": weights, "calibration_ids": calib, "ids": ids,
        "quantizers": {n: {"scale": q.scale.detach().clone(), "offset": None if q.offset is None else q.offset.detach().clone()} for n, q in named.items()},
        "input_codes": codes, "logits": logits.clone(),
    }


def g8_int4():
    """Group-128 4-bit codes and their GGUF Q4_0 nibble packing (block 32)."""
    torch.manual_seed(77)
    w = torch.randn(64, 256, dtype=torch.bfloat16)
    spec = ("block", (1,), (128,), (0,))
    scale, offset = params_from_range(w, spec, 4, "symmetric")
    q = affine.quantize_per_granularity(w, scale, offset, gran_of(spec), 4, torch.int8)
    out = {"weight": w, "scale": scale, "offset": offset, "codes": q.raw_data.clone(), "dequantized": q.dequantize().clone()}
    try:
        from fastforward.export.stages.gguf._packing import pack_q4_0_blocks

        codes32 = q.raw_data.reshape(-1, 32)
        packed = pack_q4_0_blocks(codes32, torch.ones(codes32.shape[0]))
        out["q4_0_nibbles_block32"] = packed[:, 2:].contiguous()  # drop the fp16 scale bytes
    except Exception as e:  # gguf adapter needs packages that are not installed here
        print("pack_q4_0_blocks not importable:", type(e).__name__, e)
        gq = (q.raw_data.reshape(-1, 2, 16).to(torch.int16) + 8).clamp(0, 15).to(torch.uint8)
        out["q4_0_nibbles_block32"] = gq[:, 0, :] | (gq[:, 1, :] << 4)  # the two lines of _packing.py:50-51
        out["note"] = "nibbles computed from the formula at export/stages/gguf/_packing.py:48-51"
    return out


def g9_dispatcher():
    """What predicate and kernel see on the two call paths of `linear`."""
    seen = []

    def predicate(*args, **kwargs):
        seen.append((len(args), sorted(kwargs)))
        return False

    x = affine.quantize_per_tensor(torch.randn(2, 8), 0.1, None, 8)
    w = affine.quantize_per_tensor(torch.randn(4, 8), 0.1, None, 8)
    with ff.strict_quantization(False), ff.dispatcher.register("linear", ff.dispatcher.Predicate(predicate), lambda *a, **k: None):
        ff.nn.functional.linear(x, w)
        torch.nn.functional.linear(x, w)
    return {"functional_then_torch": seen}


def g10_producers():
    """The three elementwise producers of the reference's quantized Llama helpers in bf16 (CPU eager), each
    followed by a static per-tensor 8-bit quantizer: residual add + QuantizedLlamaRMSNorm
    (quantized_llama/rms_norm.py:17-35, decoder.py:60-90), SiLU(gate) * up (mlp.py:30-40), and
    apply_rotary_pos_emb (rotary_embedding.py:14-62)."""
    sys.path.insert(0, "/root/reference/docs/examples")
    from doc_helpers.quantized_llama.rotary_embedding import apply_rotary_pos_emb
    from transformers import LlamaConfig
    from transformers.models.llama.modeling_llama import LlamaRMSNorm, LlamaRotaryEmbedding

    def quantize(t, bits=8):
        lo, hi = t.float().min(), t.float().max()
        scale, offset = parameters_for_range(lo, hi, bits, symmetric=False, allow_one_sided=True)
        q = affine.quantize_per_tensor(t, scale, offset, bits, torch.int8)
        return {"scale": scale.reshape(1).clone(), "offset": offset.reshape(1).clone(), "codes": q.raw_data.clone()}

    out = {}
    torch.manual_seed(1250)
    for name, (rows, cols) in {"rmsnorm_256": (48, 256), "rmsnorm_4096": (12, 4096), "rmsnorm_1040": (5, 1040)}.items():
        x = (torch.randn(rows, cols) * 1.7).to(torch.bfloat16)
        delta = (torch.randn(rows, cols) * 0.6).to(torch.bfloat16)
        norm = LlamaRMSNorm(cols, eps=1e-5).to(torch.bfloat16)
        with torch.no_grad():
            norm.weight.copy_((1.0 + 0.2 * torch.randn(cols)).to(torch.bfloat16))
            total = x + delta
            z = norm(total)
        out[name] = {"x": x, "delta": delta, "weight": norm.weight.detach().clone(), "eps": 1e-5, "sum": total, "normalised": z, "quantized": quantize(z)}
    gate = (torch.randn(24, 896) * 2.5).to(torch.bfloat16)
    up = torch.randn(24, 896).to(torch.bfloat16)
    gate[0, :8] = torch.tensor([0.0, -0.0, 20.0, -20.0, 88.0, -88.0, 1e-3, -1e-3]).to(torch.bfloat16)
    z = torch.nn.functional.silu(gate) * up
    out["silu_mul"] = {"gate": gate, "up": up, "product": z, "quantized": quantize(z)}
    cfg = LlamaConfig(hidden_size=256, num_attention_heads=8, num_key_value_heads=2, max_position_embeddings=128, rope_theta=500000.0)
    rotary = LlamaRotaryEmbedding(cfg)
    b, s_, d = 2, 48, 32
    q = torch.randn(b, s_, 8 * d).to(torch.bfloat16)
    k = torch.randn(b, s_, 2 * d).to(torch.bfloat16)
    cos, sin = rotary(q, torch.arange(s_)[None, :])  # [1, s, d] in q's dtype
    qe, ke = apply_rotary_pos_emb(q.view(b, s_, 8, d).transpose(1, 2), k.view(b, s_, 2, d).transpose(1, 2), cos, sin)
    out["rope"] = {"q": q, "k": k, "cos": cos[0].clone(), "sin": sin[0].clone(), "head_dim": d,
                   "q_rotated": qe.transpose(1, 2).reshape(b, s_, -1).clone(), "k_rotated": ke.transpose(1, 2).reshape(b, s_, -1).clone()}
    return out


def g11_backward():
    """quantize_by_tile_backward of the reference on seeded inputs: per-tensor, per-channel(0), group-128 and
    per-token tiles, symmetric (offset None) and asymmetric (fractional offsets), fp32 and bf16 data."""
    from fastforward.quantization import _quantizer_impl  # noqa: F401  (registers the ops)

    cases = []
    gen = torch.Generator().manual_seed(1260)
    for name, shape, tile, dtype, bits, with_offset in [
        ("tensor_f32", (32, 16, 8), (32, 16, 8), torch.float32, 4, True),
        ("tensor_f32_sym", (64, 256), (64, 256), torch.float32, 8, False),
        ("channel0_bf16", (48, 512), (1, 512), torch.bfloat16, 4, True),
        ("channel0_f32_sym", (16, 4096), (1, 4096), torch.float32, 3, False),
        ("group128_bf16", (24, 512), (1, 128), torch.bfloat16, 4, True),
        ("token_f32", (4, 6, 256), (1, 1, 256), torch.float32, 8, True),
        ("channel_last_f32", (40, 24), (40, 1), torch.float32, 4, True),
    ]:
        x = (torch.randn(shape, generator=gen) * 1.5).to(dtype)
        grad = torch.randn(shape, generator=gen).to(dtype)
        ntiles = 1
        for s_, t_ in zip(shape, tile):
            ntiles *= s_ // t_
        scale = torch.rand(ntiles, generator=gen) * 0.3 + 0.15
        offset = (torch.rand(ntiles, generator=gen) * 5 - 2.5) if with_offset else None
        dinput, dscale, doffset = torch.ops.fastforward.quantize_by_tile_backward(x, grad, scale, list(tile), float(bits), offset)
        cases.append({"name": name, "data": x, "grad": grad, "tile": list(tile), "num_bits": bits, "scale": scale, "offset": offset,
                      "dinput": dinput.clone(), "dscale": dscale.clone(), "doffset": doffset.clone()})
    return cases


def g12_mse_grid():
    """The reference's mse_grid estimator (range_setting/min_error.py) over 3 scaled batches: the search grid, the
    cumulative error of every candidate and the (scale, offset) the quantizer ends with."""
    from fastforward.range_setting.min_error import _MinAvgErrorGridEstimator, mse_grid

    cases = []
    for name, shape, spec, symmetric, negative, dtype, bits, ncand in [
        ("tensor_asym_f32", (64, 256), ("tensor",), False, True, torch.float32, 8, 25),
        ("tensor_sym_f32", (64, 256), ("tensor",), True, True, torch.float32, 4, 20),
        ("tensor_onesided_f32", (64, 256), ("tensor",), True, False, torch.float32, 4, 20),
        ("channel0_asym_f32", (24, 2048), ("channel", 0), False, True, torch.float32, 4, 16),
        ("channel0_sym_bf16", (24, 2048), ("channel", 0), True, True, torch.bfloat16, 4, 16),
        ("group128_asym_f32", (8, 512), ("block", (1,), (128,), (0,)), False, True, torch.float32, 4, 16),
        ("channel_last_sym_f32", (24, 16), ("channel", -1), True, True, torch.float32, 8, 9),
    ]:
        gen = torch.Generator().manual_seed(1270 + len(cases))
        batches = []
        for k in range(3):
            x = torch.randn(shape, generator=gen) * (1.0 + 0.25 * k)
            if not negative:
                x = x.abs()
            batches.append(x.to(dtype))
        quantizer = ff.nn.LinearQuantizer(bits, granularity=gran_of(spec), symmetric=symmetric)
        with ff.estimate_ranges(quantizer, mse_grid, num_candidates=ncand):
            estimator = next(o for o in quantizer._quantizer_overrides.values() if isinstance(o, _MinAvgErrorGridEstimator))
            for x in batches:
                quantizer(x)
            cases.append({
                "name": name, "granularity": list(spec), "symmetric": symmetric, "num_bits": bits, "num_candidates": ncand, "batches": batches,
                "min_threshold": estimator.min_threshold.clone(), "max_threshold": estimator.max_threshold.clone(),
                "cumulative_error": estimator.cumulative_error.clone(),
                "scale": quantizer.scale.detach().clone(), "offset": None if quantizer.offset is None else quantizer.offset.detach().clone(),
            })
    return cases


def g13_gptq():
    """The reference's gptq() (quantization/gptq.py) on small QuantizedLinear layers, CPU: initial weight, the layer's
    input activations, and the weight / scale / offset it ends with."""
    from fastforward.quantization.gptq import gptq

    cases = []
    for name, gran, symmetric, bits, out_f, in_f, block_size, actorder in [
        ("channel0_asym_4bit", ff.PerChannel(0), False, 4, 48, 160, 64, False),
        ("tensor_sym_4bit", ff.PerTensor(), True, 4, 40, 128, 128, False),
        ("channel0_sym_3bit_actorder", ff.PerChannel(0), True, 3, 32, 96, 32, True),
        ("group16_asym_4bit", ff.PerBlock(block_dims=1, block_sizes=16, per_channel_dims=0), False, 4, 24, 64, 32, False),
        ("channel1_sym_4bit", ff.PerChannel(1), True, 4, 24, 64, 32, False),
    ]:
        torch.manual_seed(1280 + len(cases))
        layer = torch.nn.Linear(in_f, out_f, bias=False)
        ff.quantize_model(layer)
        layer.weight_quantizer = ff.nn.LinearQuantizer(bits, granularity=gran, symmetric=symmetric)
        weight0 = layer.weight.detach().clone()
        acts = [torch.randn(2, 24, in_f) * (1.0 + 0.5 * torch.rand(in_f)) for _ in range(3)]
        dataset = [((a.clone(),), {}) for a in acts]  # calculate_hessian scales its fp32 input IN PLACE (gptq.py:312)
        with torch.no_grad(), ff.strict_quantization(False):
            gptq(layer, dataset, block_size=block_size, actorder=actorder)
        q = layer.weight_quantizer
        cases.append({"name": name, "granularity": name.split("_")[0], "symmetric": symmetric, "num_bits": bits, "block_size": block_size, "actorder": actorder,
                      "weight": weight0, "activations": acts, "result": layer.weight.detach().clone(),
                      "scale": q.scale.detach().clone(), "offset": None if q.offset is None else q.offset.detach().clone()})
    return cases


def g14_attention():
    """The attention of the reference's quantized Llama helper between the projections and o_proj
    (quantized_llama/attention.py:57-88) in bf16 on CPU, with the ops the helper itself calls — HF's repeat_kv,
    FFF.matmul, `* scaling`, the additive causal mask, FFF.softmax in fp32, FFF.matmul — followed by an 8-bit
    asymmetric per-tensor quantizer standing for o_proj's input quantizer (nn/linear.py:33)."""
    import fastforward.nn.functional as FFF
    from transformers.models.llama.modeling_llama import repeat_kv

    def quantize(t, bits=8):
        lo, hi = t.float().min(), t.float().max()
        scale, offset = parameters_for_range(lo, hi, bits, symmetric=False, allow_one_sided=True)
        q = affine.quantize_per_tensor(t, scale, offset, bits, torch.int8)
        return {"scale": scale.reshape(1).clone(), "offset": offset.reshape(1).clone(), "codes": q.raw_data.clone()}

    out = {}
    torch.manual_seed(1254)
    d = 128
    for name, (b, s_, heads, kv_heads, causal, spread) in {
        "causal_gqa_128": (2, 128, 4, 2, True, 1.0), "causal_mha_192": (1, 192, 2, 2, True, 2.0),
        "causal_gqa_320": (1, 320, 4, 1, True, 1.0), "full_gqa_64": (1, 64, 4, 2, False, 1.5),
    }.items():
        q = (torch.randn(b, s_, heads * d) * spread).to(torch.bfloat16)
        k = (torch.randn(b, s_, kv_heads * d) * spread).to(torch.bfloat16)
        v = torch.randn(b, s_, kv_heads * d).to(torch.bfloat16)
        with ff.strict_quantization(False), torch.no_grad():
            qs = q.view(b, s_, heads, d).transpose(1, 2)
            ks = repeat_kv(k.view(b, s_, kv_heads, d).transpose(1, 2), heads // kv_heads)
            vs = repeat_kv(v.view(b, s_, kv_heads, d).transpose(1, 2), heads // kv_heads)
            weights = FFF.matmul(qs, ks.transpose(2, 3)) * (d**-0.5)
            if causal:
                mask = torch.full((s_, s_), torch.finfo(torch.bfloat16).min, dtype=torch.bfloat16).triu(1)
                weights = weights + mask[None, None]
            weights = FFF.softmax(weights.to(torch.float32), dim=-1).to(qs.dtype)
            ctx = FFF.matmul(weights, vs).transpose(1, 2).contiguous().reshape(b, s_, -1)
        out[name] = {"q": q, "k": k, "v": v, "head_dim": d, "causal": causal, "context": ctx.clone(), "quantized": quantize(ctx)}
    return out


def g15_weight_only_linear():
    """QuantizedLinear with a weight quantizer only and a plain bf16 input: the non-quantized-input branch of the
    reference's fallback.linear (_gen/fallback.py:86-112, strict quantization off). W8 per output channel (BASELINE
    config 2), W4 in groups of 128 input channels (config 4: PerBlock(1, 128, 0)), W8 per tensor with an asymmetric
    (offset-carrying) quantizer, and a biased layer; M and N deliberately not multiples of the kernel's 256-wide tiles."""
    torch.manual_seed(4321)
    cases = []
    specs = [
        ("w8_per_channel", 8, ("channel", 0), True, False, (2, 100, 512), 320),
        ("w4_group128", 4, ("block", (1,), (128,), (0,)), True, False, (2, 100, 512), 320),
        ("w8_per_tensor_asymmetric_bias", 8, ("tensor",), False, True, (3, 37, 256), 200),
        ("w4_group64_asymmetric", 4, ("block", (1,), (64,), (0,)), False, False, (1, 130, 384), 256),
    ]
    for name, bits, spec, symmetric, bias, xshape, out_features in specs:
        lin = torch.nn.Linear(xshape[-1], out_features, bias=bias).to(torch.bfloat16)
        with torch.no_grad():
            lin.weight.mul_(3.0)
        x = torch.randn(*xshape).to(torch.bfloat16)
        model = torch.nn.Sequential(lin)
        ff.quantize_model(model)
        lin.weight_quantizer = ff.nn.LinearQuantizer(bits, granularity=gran_of(spec), symmetric=symmetric)
        with ff.strict_quantization(False):
            with ff.estimate_ranges(model, ff.range_setting.running_minmax):
                model(x)
            y = model(x)
            wq = lin.weight_quantizer(lin.weight)
        w_hat = wq.dequantize()
        y64 = torch.nn.functional.linear(x.double(), w_hat.double(), None if lin.bias is None else lin.bias.double())
        offset = lin.weight_quantizer.offset
        cases.append({
            "name": name, "num_bits": bits, "granularity": spec, "symmetric": symmetric,
            "x": x, "weight": lin.weight.detach().clone(), "bias": None if lin.bias is None else lin.bias.detach().clone(),
            "w_scale": lin.weight_quantizer.scale.detach().clone(), "w_offset": None if offset is None else offset.detach().clone(),
            "w_codes": wq.raw_data.to(torch.int8),
            "y": y.detach().clone(), "y_float64": y64.float(),
        })
    return cases


def g16_smoothed_minmax():
    """SmoothedMinMax trajectories (range_setting/minmax.py:67-92): exponential moving average of the per-batch extrema,
    the first batch adopted as is; five scaled batches, gamma in {1.0, 0.9, 0.3}, per tensor / per row / per column,
    fp32 and bf16 data, symmetric and asymmetric quantizers. Codes of every step and the final parameters."""
    out = []
    for spec, dtype, symmetric, gamma in itertools.product((("tensor",), ("channel", (0,)), ("channel", (-1,))), (torch.float32, torch.bfloat16), (True, False), (1.0, 0.9, 0.3)):
        g = torch.Generator().manual_seed(77)
        base = torch.randn(16, 24, generator=g)
        batches = [(base * (1.0 + 0.5 * ((i * 7) % 5)) + 0.1 * i).to(dtype) for i in range(5)]
        quantizer = ff.nn.LinearQuantizer(4, symmetric=symmetric, granularity=gran_of(spec))
        outputs, ranges = [], []
        with ff.estimate_ranges(torch.nn.ModuleList([quantizer]), ff.range_setting.smoothed_minmax, gamma=gamma):
            for b in batches:
                outputs.append(quantizer(b).raw_data.clone())
                ranges.append((quantizer.scale.detach().clone(), None if quantizer.offset is None else quantizer.offset.detach().clone()))
        out.append({
            "granularity": list(spec), "symmetric": symmetric, "num_bits": 4, "gamma": gamma, "batches": batches,
            "scale": quantizer.scale.detach().clone(),
            "offset": None if quantizer.offset is None else quantizer.offset.detach().clone(),
            "codes_per_step": outputs, "params_per_step": ranges,
        })
    return out


def g17_gguf_blocks():
    """The reference's GGUF block-32 packers themselves (export/stages/gguf/_packing.py:23-79, pure torch): Q4_0 and Q8_0
    records for seeded codes and scales, including codes outside the nibble range (clamped, :49), -128 (clipped to -127,
    :77) and scales that round in fp16. The package's __init__ pulls in `gguf` / `onnx`, absent here: a permissive stub
    importer (outside the repository, see the module docstring) lets the package import; the packers are untouched."""
    import ffstub  # noqa: F401  (outside the repository: /tmp/ffshim/ffstub.py)

    from fastforward.export.stages.gguf._packing import pack_q4_0_blocks, pack_q8_0_blocks

    g = torch.Generator().manual_seed(31)
    n = 96
    codes4 = torch.randint(-8, 8, (n, 32), generator=g, dtype=torch.int8)
    codes4[0, :4] = torch.tensor([-9, 8, 100, -100], dtype=torch.int8)  # out of range: clamped to [0, 15] after + 8
    codes8 = torch.randint(-128, 128, (n, 32), generator=g, dtype=torch.int8)
    codes8[1, :3] = torch.tensor([-128, 127, -127], dtype=torch.int8)
    scales = torch.rand(n, generator=g) * 0.05 + 1e-3
    scales[2] = 65504.0  # fp16 max
    scales[3] = 1e-8     # underflows to an fp16 subnormal / zero
    scales[4] = 0.1      # not representable in fp16: rounds
    return {"codes4": codes4, "codes8": codes8, "scales": scales,
            "q4_0": pack_q4_0_blocks(codes4, scales), "q8_0": pack_q8_0_blocks(codes8, scales)}


def g18_linear_large():
    """A W8A8 QuantizedLinear large enough for the 256 x 256-tile persistent GEMM (2048 tokens, N = 2048, K = 512: 64
    tiles): the reference's fallback.linear (dequantize, dequantize, F.linear in bf16). Inputs are regenerated from seeds
    (datagen.make_data); stored are the parameters, row sums of all codes (any flipped code shows) and every 32nd output
    row with its float64 recomputation."""
    x = make_data(181, (2, 1024, 512), torch.bfloat16, "normal")
    w = (make_data(182, (2048, 512), torch.float32, "normal") * 0.05).to(torch.bfloat16)
    lin = torch.nn.Linear(512, 2048, bias=False).to(torch.bfloat16)
    with torch.no_grad():
        lin.weight.copy_(w)
    model = torch.nn.Sequential(lin)
    ff.quantize_model(model)
    lin.weight_quantizer = ff.nn.LinearQuantizer(8, granularity=ff.PerChannel(0))
    lin.input_quantizer = ff.nn.LinearQuantizer(8, symmetric=False)
    with ff.strict_quantization(False):
        with ff.estimate_ranges(model, ff.range_setting.running_minmax):
            model(x)
        y = model(x)
        xq, wq = lin.input_quantizer(x), lin.weight_quantizer(lin.weight)
    rows = torch.arange(0, 2048, 32)
    y2 = y.detach().reshape(2048, 2048)
    y64 = torch.nn.functional.linear(xq.dequantize().double().reshape(2048, 512)[rows], wq.dequantize().double())
    return {
        "x_seed": 181, "w_seed": 182, "w_factor": 0.05, "x_shape": [2, 1024, 512], "w_shape": [2048, 512],
        "x_scale": lin.input_quantizer.scale.detach().clone(), "x_offset": lin.input_quantizer.offset.detach().clone(),
        "w_scale": lin.weight_quantizer.scale.detach().clone(),
        "x_code_row_sums": xq.raw_data.reshape(2048, 512).to(torch.int64).sum(1), "w_code_row_sums": wq.raw_data.to(torch.int64).sum(1),
        "x_code_abs_sum": int(xq.raw_data.to(torch.int64).abs().sum()), "w_code_abs_sum": int(wq.raw_data.to(torch.int64).abs().sum()),
        "rows": rows, "y_rows": y2[rows].clone(), "y_rows_float64": y64.float(),
    }


def main() -> None:
    torch.set_num_threads(8)
    if len(sys.argv) > 1:  # regenerate only the named fixtures, e.g. `gen_golden.py g10_producers`
        for name in sys.argv[1:]:
            torch.save(globals()[name](), HERE / f"{name}.pt")
            print(f"{name}.pt: {(HERE / f'{name}.pt').stat().st_size / 1024:.0f} KiB")
        return
    torch.save(g1_known_answers(), HERE / "g1_known_answers.pt")
    torch.save(g2_edges(), HERE / "g2_edges.pt")
    torch.save(g3_sweeps(), HERE / "g3_sweeps.pt")
    torch.save(g3_dtype_sweep(), HERE / "g3_dtype_sweep.pt")
    torch.save(g4_ranges(), HERE / "g4_ranges.pt")
    torch.save(g5_minmax(), HERE / "g5_minmax.pt")
    torch.save(g6_linear(), HERE / "g6_linear.pt")
    torch.save(g7_tiny_llama(), HERE / "g7_tiny_llama.pt")
    torch.save(g8_int4(), HERE / "g8_int4.pt")
    torch.save(g9_dispatcher(), HERE / "g9_dispatcher.pt")
    torch.save(g10_producers(), HERE / "g10_producers.pt")
    torch.save(g11_backward(), HERE / "g11_backward.pt")
    torch.save(g12_mse_grid(), HERE / "g12_mse_grid.pt")
    torch.save(g13_gptq(), HERE / "g13_gptq.pt")
    torch.save(g14_attention(), HERE / "g14_attention.pt")
    torch.save(g15_weight_only_linear(), HERE / "g15_weight_only_linear.pt")
    torch.save(g16_smoothed_minmax(), HERE / "g16_smoothed_minmax.pt")
    torch.save(g17_gguf_blocks(), HERE / "g17_gguf_blocks.pt")
    torch.save(g18_linear_large(), HERE / "g18_linear_large.pt")
    for f in sorted(HERE.glob("*.pt")):
        print(f"{f.name}: {f.stat().st_size / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
