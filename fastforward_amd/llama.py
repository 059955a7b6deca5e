"""Llama-3-shaped caller of the hot path: the harness behind bench.py and the multi-GPU calibration.

The reference defines "Llama W8A8" through its example helpers
(docs/examples/doc_helpers/quantized_llama/{attention,mlp,decoder,rms_norm}.py) and the quick-start
recipe (docs/examples/quick_start_quantize_llms.nb.py:140-232):

  * ``ff.quantize_model(model)`` turns every ``nn.Linear`` into a ``QuantizedLinear`` and gives the
    attention / MLP / decoder / norm modules identity ``QuantizerStub``s;
  * weight quantizers of the 7 linears per layer: ``LinearQuantizer(8, granularity=PerChannel())``;
  * input quantizers of the same 7 linears: ``LinearQuantizer(8, symmetric=False, PerTensor())``;
  * embedding, lm_head, RMSNorm, rotary, softmax stay float (``strict_quantization(False)``).

This module restates that structure with its own minimal decoder (no ``transformers`` dependency on
the GPU box): per decoder layer 7 weight quantizations (A1, per-channel, re-run every forward as in
reference nn/linear.py:34), 7 activation quantizations (A1, per-tensor asymmetric) and 7 quantized
linears (A6). Everything that is not a quantized linear runs as ordinary torch ops — it is the
traffic between hot-path calls, not the product.
"""

from __future__ import annotations

import contextlib
import dataclasses
import math
import types

import torch
import torch.nn.functional as F

import fastforward_amd as ff

from fastforward_amd.nn import QuantizedModule, QuantizerStub
from fastforward_amd.quantization.affine._memo import RECENT, sibling_quantizers


@dataclasses.dataclass(frozen=True)
class LlamaConfig:
    hidden_size: int = 4096
    intermediate_size: int = 14336
    num_layers: int = 32
    num_heads: int = 32
    num_kv_heads: int = 8
    vocab_size: int = 128256
    rope_theta: float = 500000.0
    rms_norm_eps: float = 1e-5
    attention: str = "sdpa"  # "sdpa": fused torch attention; "eager": the op sequence of the reference's helper

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_heads

    @classmethod
    def llama3_8b(cls) -> "LlamaConfig":
        return cls()

    @classmethod
    def llama3_70b(cls) -> "LlamaConfig":
        return cls(hidden_size=8192, intermediate_size=28672, num_layers=80, num_heads=64, num_kv_heads=8)

    @classmethod
    def tiny(cls) -> "LlamaConfig":
        """The 2-layer shape used by tests."""
        return cls(hidden_size=256, intermediate_size=896, num_layers=2, num_heads=8, num_kv_heads=2, vocab_size=512)

    def quantized_weight_elems(self) -> int:
        h, i, kv = self.hidden_size, self.intermediate_size, self.num_kv_heads * self.head_dim
        return self.num_layers * (2 * h * h + 2 * kv * h + 3 * h * i)

    def linear_flops_per_token(self) -> int:
        return 2 * self.quantized_weight_elems()


class LlamaRMSNorm(torch.nn.Module):
    def __init__(self, hidden_size: int, eps: float) -> None:
        super().__init__()
        self.weight = torch.nn.Parameter(torch.ones(hidden_size))
        self.variance_epsilon = eps

    def forward(self, hidden_states: torch.Tensor) -> torch.Tensor:
        dtype = hidden_states.dtype
        h = hidden_states.to(torch.float32)
        h = h * torch.rsqrt(h.pow(2).mean(-1, keepdim=True) + self.variance_epsilon)
        return self.weight * h.to(dtype)


def rotary_tables(seq_len: int, head_dim: int, theta: float, device: torch.device | str, dtype: torch.dtype) -> tuple[torch.Tensor, torch.Tensor]:
    inv_freq = 1.0 / (theta ** (torch.arange(0, head_dim, 2, device=device, dtype=torch.float32) / head_dim))
    freqs = torch.outer(torch.arange(seq_len, device=device, dtype=torch.float32), inv_freq)
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos().to(dtype), emb.sin().to(dtype)


def _rotate_half(x: torch.Tensor) -> torch.Tensor:
    half = x.shape[-1] // 2
    return torch.cat((-x[..., half:], x[..., :half]), dim=-1)


class LlamaAttention(torch.nn.Module):
    def __init__(self, config: LlamaConfig) -> None:
        super().__init__()
        self.config = config
        h, d = config.hidden_size, config.head_dim
        self.q_proj = torch.nn.Linear(h, config.num_heads * d, bias=False)
        self.k_proj = torch.nn.Linear(h, config.num_kv_heads * d, bias=False)
        self.v_proj = torch.nn.Linear(h, config.num_kv_heads * d, bias=False)
        self.o_proj = torch.nn.Linear(config.num_heads * d, h, bias=False)

    def forward(self, hidden_states: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor) -> torch.Tensor:
        b, s, _ = hidden_states.shape
        cfg = self.config
        q = self.q_proj(hidden_states).view(b, s, cfg.num_heads, cfg.head_dim).transpose(1, 2)
        k = self.k_proj(hidden_states).view(b, s, cfg.num_kv_heads, cfg.head_dim).transpose(1, 2)
        v = self.v_proj(hidden_states).view(b, s, cfg.num_kv_heads, cfg.head_dim).transpose(1, 2)
        q = q * cos + _rotate_half(q) * sin
        k = k * cos + _rotate_half(k) * sin
        if cfg.attention == "eager":
            attn = self._eager_attention(q, k, v)
        else:
            attn = F.scaled_dot_product_attention(q, k, v, is_causal=s > 1, enable_gqa=cfg.num_kv_heads != cfg.num_heads)
        return self.o_proj(attn.transpose(1, 2).reshape(b, s, -1))

    def _eager_attention(self, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
        """matmul -> scale -> additive causal mask -> fp32 softmax -> matmul, the order used by the
        reference's QuantizedLlamaAttention (docs/examples/doc_helpers/quantized_llama/attention.py:60-92)."""
        cfg = self.config
        groups = cfg.num_heads // cfg.num_kv_heads
        b, _, s, d = q.shape
        if groups > 1:  # repeat_kv
            k = k[:, :, None, :, :].expand(b, cfg.num_kv_heads, groups, s, d).reshape(b, cfg.num_heads, s, d)
            v = v[:, :, None, :, :].expand(b, cfg.num_kv_heads, groups, s, d).reshape(b, cfg.num_heads, s, d)
        weights = torch.matmul(q, k.transpose(2, 3)) * (d**-0.5)
        mask = torch.full((s, s), torch.finfo(q.dtype).min, dtype=q.dtype, device=q.device).triu(1)
        weights = weights + mask
        weights = F.softmax(weights.to(torch.float32), dim=-1).to(q.dtype)
        return torch.matmul(weights, v)


class LlamaMLP(torch.nn.Module):
    def __init__(self, config: LlamaConfig) -> None:
        super().__init__()
        self.gate_proj = torch.nn.Linear(config.hidden_size, config.intermediate_size, bias=False)
        self.up_proj = torch.nn.Linear(config.hidden_size, config.intermediate_size, bias=False)
        self.down_proj = torch.nn.Linear(config.intermediate_size, config.hidden_size, bias=False)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.down_proj(F.silu(self.gate_proj(x)) * self.up_proj(x))


class LlamaDecoderLayer(torch.nn.Module):
    def __init__(self, config: LlamaConfig) -> None:
        super().__init__()
        self.self_attn = LlamaAttention(config)
        self.mlp = LlamaMLP(config)
        self.input_layernorm = LlamaRMSNorm(config.hidden_size, config.rms_norm_eps)
        self.post_attention_layernorm = LlamaRMSNorm(config.hidden_size, config.rms_norm_eps)

    def forward(self, hidden_states: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor) -> torch.Tensor:
        hidden_states = hidden_states + self.self_attn(self.input_layernorm(hidden_states), cos, sin)
        return hidden_states + self.mlp(self.post_attention_layernorm(hidden_states))


class LlamaModel(torch.nn.Module):
    """Embedding -> decoder layers -> final norm -> lm_head (logits)."""

    def __init__(self, config: LlamaConfig) -> None:
        super().__init__()
        self.config = config
        self.embed_tokens = torch.nn.Embedding(config.vocab_size, config.hidden_size)
        self.layers = torch.nn.ModuleList(LlamaDecoderLayer(config) for _ in range(config.num_layers))
        self.norm = LlamaRMSNorm(config.hidden_size, config.rms_norm_eps)
        self.lm_head = torch.nn.Linear(config.hidden_size, config.vocab_size, bias=False)

    def forward(self, input_ids: torch.Tensor, logits: bool = True) -> torch.Tensor:
        hidden_states = self.embed_tokens(input_ids)
        cos, sin = rotary_tables(input_ids.shape[1], self.config.head_dim, self.config.rope_theta, hidden_states.device, hidden_states.dtype)
        for layer in self.layers:
            hidden_states = layer(hidden_states, cos, sin)
        hidden_states = self.norm(hidden_states)
        return self.lm_head(hidden_states) if logits else hidden_states


# ---- quantized counterparts: same stub slots as the reference's helpers ---------------------------
# The reference's helper modules run RMSNorm, SiLU * up, the rotary embedding and attention as eager ATen chains between their
# quantizer slots (docs/examples/doc_helpers/quantized_llama/). Where EVERY slot between two operations is still a stub (the
# Llama recipe sets none of them) those chains have exactly one meaning, and the modules below run them as the one-pass
# kernels of csrc/ffq_producers.hip / ffq_attention.hip: the module graph that ``ff.quantize_model`` builds is itself fast,
# no harness needed. Any quantizer installed in one of the slots, an active override, a CPU tensor, a non-bf16 dtype or a
# caller that wants gradients sends the module back to the eager chain. ``with llama.eager_modules():`` forces the eager
# chains (the reference-shaped arm of the tests and of bench.py).
_ONE_PASS_MODULES = True


# q's rotary embedding inside the attention launch (ops.attention(q_rope=...)) instead of a pass over the q projection; a switch for
# A/B measurements (tools/rope_ab.py) — both settings give the same bits
FUSE_Q_ROPE = True


@contextlib.contextmanager
def eager_modules(eager: bool = True):
    """Run the quantized Llama modules' forwards as the reference's eager ATen chains inside the block."""
    global _ONE_PASS_MODULES
    previous, _ONE_PASS_MODULES = _ONE_PASS_MODULES, not eager
    try:
        yield
    finally:
        _ONE_PASS_MODULES = previous


def _untouched(*quantizers: torch.nn.Module | None) -> bool:
    return all(q is None or (q.is_stub() and next(iter(q.overrides), None) is None) for q in quantizers)


def _hooked(*modules: torch.nn.Module | None) -> bool:
    """True if calling one of `modules` through ``nn.Module.__call__`` would run a user hook (its own or a global one). The
    one-pass paths below compute what a group of module calls would have computed without making those calls; hooks on the
    bypassed modules would silently stop firing — the reference's strict_quantization_for_module (strict_quantization.py:67-68),
    its ModuleIORecorder (export/_io_capture.py:78) and the usual GPTQ input capture hang exactly such hooks on linears and
    quantizers — so any hook sends the caller back to the module-by-module route."""
    from torch.nn.modules import module as nn_module

    for table in ("_global_forward_hooks", "_global_forward_pre_hooks", "_global_backward_hooks", "_global_backward_pre_hooks"):
        if getattr(nn_module, table, None):
            return True
    for m in modules:
        if m is None:
            continue
        if m._forward_hooks or m._forward_pre_hooks or m._backward_hooks or getattr(m, "_backward_pre_hooks", None):
            return True
    return False


def _codes_stay_with_the_linears(*linears: torch.nn.Module, module_calls: bool = False) -> bool:
    """Whether what the input quantizers of `linears` return is read by nobody but the GEMM of their linear — the condition for
    ``sibling_quantizers(undecided=True)``, under which a later sibling's codes may stay unwritten (quantization/affine/_memo.py):
    no hook on those quantizers (a forward hook is handed the codes), no override on them from outside this package (an override
    wraps the quantizer's forward and may look at what it returns; the range estimators of this package do not), and — where the
    linears run as module calls (`module_calls`: the module graph) — linears that are exactly ``QuantizedLinear``, whose forward
    hands the codes to ``ff.nn.functional.linear`` and to nothing else (that seam settles them for any kernel but this package's)."""
    from fastforward_amd.nn import QuantizedLinear

    for linear in linears:
        quantizer = getattr(linear, "input_quantizer", None)
        if quantizer is None or _hooked(quantizer):
            return False
        if any(not str(getattr(o, "__module__", "")).startswith("fastforward_amd.") for o in getattr(quantizer, "overrides", ())):
            return False
        if module_calls and (type(linear) is not QuantizedLinear or _hooked(linear)):
            return False
    return True


def _one_pass(*tensors: torch.Tensor) -> bool:
    from fastforward_amd import _native

    return (_ONE_PASS_MODULES and all(type(t) in (torch.Tensor, torch.nn.Parameter) and t.is_cuda and t.dtype == torch.bfloat16 for t in tensors)
            and not (torch.is_grad_enabled() and any(t.requires_grad for t in tensors)) and _native.is_available())


def _weight_only_gate_up(x: torch.Tensor, gate_proj: torch.nn.Module, up_proj: torch.nn.Module,
                         stored: tuple[tuple[torch.Tensor, int], tuple[torch.Tensor, int]] | None = None) -> torch.Tensor | None:
    """``silu(gate_proj(x)) * up_proj(x)`` of a WEIGHT-ONLY quantized MLP (plain bf16 `x`, quantized weights: BASELINE configs 2
    and 4) as one launch of the weight-code GEMM with the SiLU * up epilogue (ops.mlp_gate_up_wq) — the value
    ``silu_mul_quantize(gate_proj(x), up_proj(x))`` has when both projections go through the dispatcher's weight-only kernel,
    bit for bit. None whenever that is not the situation (any set activation quantizer, a bias, parameters the kernel does not
    cover, a hook on a bypassed module): the caller then runs the two module forwards.
    `stored`: (codes, packing block) of the two weights when the caller keeps them; else each weight quantizer runs here, as
    QuantizedLinear.forward would run it (reference nn/linear.py:34)."""
    from fastforward_amd import fused_linear
    from fastforward_amd.nn import QuantizedLinear

    kernels = fused_linear.KERNELS
    if not _one_pass(x) or not x.is_contiguous() or _hooked(gate_proj, up_proj):
        return None
    for lin in (gate_proj, up_proj):
        if (not isinstance(lin, QuantizedLinear) or lin.bias is not None or lin.weight_quantizer.is_stub()
                or not _untouched(lin.input_quantizer, lin.output_quantizer) or _hooked(lin.input_quantizer, lin.output_quantizer)):
            return None
    if not fused_linear._WEIGHT_ONLY_KERNEL:
        return None
    if stored is not None:
        gq, uq = gate_proj.weight_quantizer, up_proj.weight_quantizer
        group = kernels.weight_group(_Shaped(gate_proj.weight, gq))
        if group is None or group != kernels.weight_group(_Shaped(up_proj.weight, uq)) or stored[0][1] != stored[1][1]:
            return None
        return ff.ops.mlp_gate_up_wq(x, stored[0][0], stored[1][0], gq.scale, gq.offset, uq.scale, uq.offset, group=group, pack_block=stored[0][1])
    gw, uw = gate_proj.weight_quantizer(gate_proj.weight), up_proj.weight_quantizer(up_proj.weight)
    product = None
    if kernels.supported_weight_only(x, gw) and kernels.supported_weight_only(x, uw) and kernels.weight_group(gw) == kernels.weight_group(uw):
        (gs, go), (us, uo) = kernels._scale_offset(gw), kernels._scale_offset(uw)
        product = ff.ops.mlp_gate_up_wq(x, kernels._int8_codes(gw), kernels._int8_codes(uw), gs, go, us, uo, group=kernels.weight_group(gw))
    if product is None:  # the weights are quantized already: finish the two linears as QuantizedLinear.forward would
        gate = ff.nn.functional.linear(x, gw, None, output_quantizer=gate_proj.output_quantizer)
        up = ff.nn.functional.linear(x, uw, None, output_quantizer=up_proj.output_quantizer)
        if not (_one_pass(gate, up) and gate.shape == up.shape):
            return F.silu(gate) * up
        product = ff.ops.silu_mul_quantize(gate, up, (), want_product=True)[0]
    return product


def _weight_only_qkv(x: torch.Tensor, projections: tuple[torch.nn.Module, ...],
                     stored: tuple[tuple[torch.Tensor, int], ...] | None = None) -> list[torch.Tensor] | None:
    """``[q_proj(x), k_proj(x), v_proj(x)]`` of a WEIGHT-ONLY quantized attention block (plain bf16 `x`, quantized weights: BASELINE
    configs 2 and 4) as ONE launch of the weight-code GEMM over the concatenated output columns (ops.linear_wq_multi): each output
    is what the dispatcher's weight-only kernel gives for that projection (same operands; the K order of one tile walk). None
    whenever that is not the situation (a set activation quantizer, a bias, a hook on a bypassed module, parameters the kernel does
    not cover, differing granularities): the caller then runs the module forwards. `stored`: (codes, packing block) per weight
    when the caller keeps them; else each weight quantizer runs here, as QuantizedLinear.forward would run it (nn/linear.py:34)."""
    from fastforward_amd import fused_linear
    from fastforward_amd.nn import QuantizedLinear

    kernels = fused_linear.KERNELS
    if not fused_linear._WEIGHT_ONLY_KERNEL or not _one_pass(x) or not x.is_contiguous() or _hooked(*projections):
        return None
    for lin in projections:
        if (not isinstance(lin, QuantizedLinear) or lin.bias is not None or lin.weight_quantizer.is_stub()
                or not _untouched(lin.input_quantizer, lin.output_quantizer) or _hooked(lin.input_quantizer, lin.output_quantizer)):
            return None
    if stored is not None:
        groups = {kernels.weight_group(_Shaped(lin.weight, lin.weight_quantizer)) for lin in projections}
        if len(groups) != 1 or None in groups or len({block for _, block in stored}) != 1:
            return None
        quantizers = [lin.weight_quantizer for lin in projections]
        return ff.ops.linear_wq_multi(x, [codes for codes, _ in stored], [q.scale for q in quantizers], [q.offset for q in quantizers],
                                      group=groups.pop(), pack_block=stored[0][1], out_dtype=x.dtype)
    weights = [lin.weight_quantizer(lin.weight) for lin in projections]
    outs = None
    if all(kernels.supported_weight_only(x, w) for w in weights) and len({kernels.weight_group(w) for w in weights}) == 1:
        params = [kernels._scale_offset(w) for w in weights]
        if len({o is None for _, o in params}) == 1:
            outs = ff.ops.linear_wq_multi(x, [kernels._int8_codes(w) for w in weights], [s for s, _ in params], [o for _, o in params],
                                          group=kernels.weight_group(weights[0]), out_dtype=x.dtype)
    if outs is None:  # the weights are quantized already: finish the linears as QuantizedLinear.forward would
        outs = [ff.nn.functional.linear(x, w, None, output_quantizer=lin.output_quantizer) for lin, w in zip(projections, weights)]
    return outs


def _w8a8_gate_up_down_input(x: torch.Tensor, gate_proj: torch.nn.Module, up_proj: torch.nn.Module, down_proj: torch.nn.Module):
    """The input of ``down_proj``'s GEMM on a W8A8 model — ``down_proj.input_quantizer(silu(gate_proj(x)) * up_proj(x))`` — from ONE
    launch of the int8 GEMM's gate/up mode (ops.mlp_gate_up_w8a8: both projections, SiLU * up and A1 in its epilogue; equal to the
    module-by-module result bit for bit, tests/test_fullsize_gpu.py), when every quantizer involved is a plain static one:
    per-tensor 8-bit int8 input quantizers on gate / up that produced the SAME codes (the activation-code memo served the second
    from the first), per-output-channel symmetric int8 weights, a fusable input quantizer on down_proj
    (DispatcherKernels._requant's rule: initialised, per tensor, no override, fp32 parameters, not exporting).
    Returns (QuantizedTensor for down_proj's GEMM, None) on success, else (None, (gate, up)) with the two projections finished
    the ordinary way from the quantized operands already produced (nothing runs twice) — or (None, None) when the situation
    is not even close (the caller runs the module forwards)."""
    from fastforward_amd import fused_linear
    from fastforward_amd.nn import QuantizedLinear

    kernels = fused_linear.KERNELS
    if not _one_pass(x) or not x.is_contiguous() or _hooked(gate_proj, up_proj, down_proj, gate_proj.output_quantizer, up_proj.output_quantizer):
        return None, None
    for lin in (gate_proj, up_proj, down_proj):
        if not isinstance(lin, QuantizedLinear) or lin.bias is not None or lin.weight_quantizer.is_stub() or lin.input_quantizer.is_stub():
            return None, None
    if not _untouched(gate_proj.output_quantizer, up_proj.output_quantizer):
        return None, None
    fused = kernels._requant(down_proj.input_quantizer, x.dtype)
    if fused is None or fused["out_dtype"] != torch.int8:
        return None, None
    with sibling_quantizers():
        xg, xu = gate_proj.input_quantizer(x), up_proj.input_quantizer(x)
    gw, uw = gate_proj.weight_quantizer(gate_proj.weight), up_proj.weight_quantizer(up_proj.weight)
    codes = None
    if (kernels.supported_linear(xg, gw) and kernels.supported_linear(xu, uw) and kernels.row_mode(xg) == "tensor"
            and kernels.row_mode(gw) == "row" and kernels.row_mode(uw) == "row"
            and xg.raw_data.dtype == torch.int8 and gw.raw_data.dtype == torch.int8 and uw.raw_data.dtype == torch.int8
            and xu.raw_data.data_ptr() == xg.raw_data.data_ptr() and xu.raw_data.shape == xg.raw_data.shape and xu.raw_data.dtype == xg.raw_data.dtype):
        (xs, xo), (gs, go), (us, uo) = kernels._scale_offset(xg), kernels._scale_offset(gw), kernels._scale_offset(uw)
        if go is None and uo is None and kernels._deq_dtype(xg) == x.dtype:
            codes = ff.ops.mlp_gate_up_w8a8(xg.raw_data, gw.raw_data, uw.raw_data, xs, xo, gs, us, fused["out_scale"], fused["out_offset"], fused["out_num_bits"])
    if codes is not None:
        return kernels._wrap(None, codes, down_proj.input_quantizer, x.dtype), None
    gate = ff.nn.functional.linear(xg, gw, None, output_quantizer=gate_proj.output_quantizer)
    up = ff.nn.functional.linear(xu, uw, None, output_quantizer=up_proj.output_quantizer)
    return None, (gate, up)


def _w8a8_gate_up_product(x: torch.Tensor, gate_proj: torch.nn.Module, up_proj: torch.nn.Module, want_extrema: bool):
    """``silu(gate_proj(x)) * up_proj(x)`` of a W8A8 MLP whose quantizers may still be moving (range estimation) from ONE op
    (ops.mlp_gate_up_w8a8_estimating): every quantizer runs as itself first — input and weight quantizer of either linear, each
    with whatever override it carries (an estimator step) — then the op decides ON THE DEVICE whether the two input quantizers
    hold equal parameters (one launch: gate + up + SiLU * up) or not (two linears, the second with the gated epilogue). Returns
    ``(product, pair or None, None)`` — `pair` = [min, max] of the product when asked for — or ``(None, None, pre)`` with
    ``pre = ((xq_gate, wq_gate), (xq_up, wq_up))`` when the quantizers have run but the op does not apply (the caller finishes
    from them: nothing runs twice), or ``(None, None, None)`` when nothing was touched."""
    from fastforward_amd import fused_linear
    from fastforward_amd.nn import QuantizedLinear

    if not _one_pass(x) or not x.is_contiguous() or _hooked(gate_proj, up_proj, gate_proj.output_quantizer, up_proj.output_quantizer):
        return None, None, None
    for lin in (gate_proj, up_proj):
        if not isinstance(lin, QuantizedLinear) or lin.bias is not None or lin.weight_quantizer.is_stub() or lin.input_quantizer.is_stub():
            return None, None, None
    if not _untouched(gate_proj.output_quantizer, up_proj.output_quantizer):
        return None, None, None
    pre = tuple((lin.input_quantizer(x), lin.weight_quantizer(lin.weight)) for lin in (gate_proj, up_proj))
    params = []
    for xq, wq in pre:
        if not (isinstance(xq, ff.QuantizedTensor) and isinstance(wq, ff.QuantizedTensor)) or xq.raw_data.dtype != torch.int8 or wq.raw_data.dtype != torch.int8:
            return None, None, pre
        xp, wp = xq.quantization_context.quantization_params, wq.quantization_context.quantization_params
        if (xp.scale.numel() != 1 or fused_linear.KERNELS.row_mode(wq) != "row" or xp.num_bits > 8 or wp.num_bits > 8
                or fused_linear.KERNELS._deq_dtype(xq) != x.dtype):
            return None, None, pre
        w_offset = None if wp.offset is None or fused_linear.known_zero_offset(wp.offset) else wp.offset
        params.append(((xp.scale, xp.offset), (wp.scale, w_offset)))
    # up_proj's codes may be an undecided sibling's (``sibling_quantizers(undecided=True)``): the op reads gate_proj's codes wherever
    # the two input quantizers agree, which is exactly where those were left unwritten — provided gate_proj's ARE the earlier ones
    # gate_proj's own codes may be an undecided sibling's too (an OUTER ``sibling_quantizers(undecided=True)`` scope that had quantized `x`
    # before): they are handed to the op both as codes and as "the earlier codes", so they have to be written
    RECENT.settle(pre[0][0])
    earlier = RECENT.earlier_of(pre[1][0])
    if earlier is not None and not (earlier[0].data_ptr() == pre[0][0].raw_data.data_ptr() and earlier[1] is params[0][0][0] and earlier[2] is params[0][0][1]):
        RECENT.settle(pre[1][0])
    out = ff.ops.mlp_gate_up_w8a8_estimating(pre[0][0].raw_data, pre[1][0].raw_data, pre[0][1].raw_data, pre[1][1].raw_data,
                                             params[0][0], params[1][0], params[0][1], params[1][1], want_extrema=want_extrema)
    if out is None:
        return None, None, pre
    return (out[0], out[1], None) if want_extrema else (out, None, None)


def _estimating(quantizer: torch.nn.Module | None) -> bool:
    """The quantizer carries an override (a range estimator): its parameters are about to move with the data it is given."""
    return quantizer is not None and next(iter(getattr(quantizer, "overrides", ())), None) is not None


class QuantizedLlamaRMSNorm(QuantizedModule, LlamaRMSNorm):
    """Float under strict_quantization(False), like reference rms_norm.py:17-35."""

    def __init_quantization__(self) -> None:
        super().__init_quantization__()
        self.input_quantizer = QuantizerStub(input_quantizer=True)
        self.output_quantizer = QuantizerStub(output_quantizer=True)
        self.weight_quantizer = QuantizerStub(weight_quantizer=True)

    def _fusable(self, hidden_states: torch.Tensor) -> bool:
        width = hidden_states.shape[-1]
        return (_untouched(self.input_quantizer, self.output_quantizer, self.weight_quantizer) and _one_pass(hidden_states, self.weight)
                and width % 16 == 0 and width <= 8192)

    def forward(self, hidden_states: torch.Tensor) -> torch.Tensor:
        if self._fusable(hidden_states):
            return ff.ops.add_rmsnorm_quantize(hidden_states, None, self.weight, self.variance_epsilon, (), want_sum=False, want_norm=True)[1]
        with ff.strict_quantization(False):
            return self.output_quantizer(LlamaRMSNorm.forward(self, self.input_quantizer(hidden_states)))


class QuantizedLlamaAttention(QuantizedModule, LlamaAttention):
    """Reference attention.py:129-183 (the SDPA variant: one input stub, fused attention)."""

    def __init_quantization__(self) -> None:
        super().__init_quantization__()
        self.input_quantizer = QuantizerStub(input_quantizer=True)

    def forward(self, hidden_states: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor) -> torch.Tensor:
        hidden_states = self.input_quantizer(hidden_states)
        cfg = self.config
        if hidden_states.dim() == 3 and _one_pass(hidden_states) and attention_kernel_covers(cfg, hidden_states.shape[1], hidden_states.dtype):
            fused_qkv = _weight_only_qkv(hidden_states, (self.q_proj, self.k_proj, self.v_proj)) if type(hidden_states) is torch.Tensor else None
            if fused_qkv is not None:
                q, k, v = fused_qkv
            else:
                # equal input quantizers on the same hidden state share one A1 launch; quantizers whose parameters only the device
                # can compare (range estimation) leave it to the device whether the second and third launch run
                with sibling_quantizers(undecided=_codes_stay_with_the_linears(self.q_proj, self.k_proj, self.v_proj, module_calls=True)):
                    q, k, v = self.q_proj(hidden_states), self.k_proj(hidden_states), self.v_proj(hidden_states)
            if _one_pass(q, k, v, cos, sin) and cos.dim() == 2 and q.is_contiguous() and k.is_contiguous() and v.is_contiguous():
                # rotary embedding in place on the projections, then softmax(q k^T) v as one flash-style launch
                # (k in place; q as the attention launch loads it: no pass over the q projection)
                ff.ops.rope_(None if FUSE_Q_ROPE else q, k, cos, sin, cfg.head_dim)
                q_rope = (cos, sin) if FUSE_Q_ROPE else None
                causal = hidden_states.shape[1] > 1
                o_proj = self.o_proj
                fused = self._o_proj_input_in_epilogue(q.dtype)
                if fused is not None:
                    # o_proj's input quantizer (nn/linear.py:33) as the attention launch's epilogue: its codes == A1 of the
                    # bf16 context the launch would have written (tested), which never visits HBM
                    from fastforward_amd import fused_linear

                    _, codes = ff.ops.attention(q, k, v, cfg.head_dim, causal=causal, quantizer=(fused["out_scale"], fused["out_offset"]),
                                                num_bits=fused["out_num_bits"], want_context=False, q_rope=q_rope)
                    quantized = fused_linear.KERNELS._wrap(None, codes, o_proj.input_quantizer, q.dtype)
                    return ff.nn.functional.linear(quantized, o_proj.weight_quantizer(o_proj.weight), None, output_quantizer=o_proj.output_quantizer)
                ctx, _ = ff.ops.attention(q, k, v, cfg.head_dim, causal=causal, q_rope=q_rope)
                return o_proj(ctx)
            return self._attend(q, k, v, cos, sin)
        return LlamaAttention.forward(self, hidden_states, cos, sin)

    def _o_proj_input_in_epilogue(self, dtype: torch.dtype) -> dict | None:
        """Parameters of o_proj's input quantizer when the attention launch can apply it (a plain static per-tensor 8-bit int8
        LinearQuantizer without overrides: DispatcherKernels._requant's rule; o_proj an ordinary QuantizedLinear without bias)."""
        from fastforward_amd import fused_linear
        from fastforward_amd.nn import QuantizedLinear

        o_proj = self.o_proj
        if not isinstance(o_proj, QuantizedLinear) or o_proj.bias is not None or o_proj.weight_quantizer.is_stub() or o_proj.input_quantizer.is_stub():
            return None
        if _hooked(o_proj):  # the fused route calls neither o_proj nor its input quantizer (the latter: _requant's own check)
            return None
        fused = fused_linear.KERNELS._requant(o_proj.input_quantizer, dtype)
        return fused if fused is not None and fused["out_dtype"] == torch.int8 else None

    def _attend(self, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor) -> torch.Tensor:
        """The rest of LlamaAttention.forward for projections that are already computed."""
        cfg = self.config
        b, s = q.shape[0], q.shape[1]
        q = q.view(b, s, cfg.num_heads, cfg.head_dim).transpose(1, 2)
        k = k.view(b, s, cfg.num_kv_heads, cfg.head_dim).transpose(1, 2)
        v = v.view(b, s, cfg.num_kv_heads, cfg.head_dim).transpose(1, 2)
        q = q * cos + _rotate_half(q) * sin
        k = k * cos + _rotate_half(k) * sin
        if cfg.attention == "eager":
            attn = self._eager_attention(q, k, v)
        else:
            attn = F.scaled_dot_product_attention(q, k, v, is_causal=s > 1, enable_gqa=cfg.num_kv_heads != cfg.num_heads)
        return self.o_proj(attn.transpose(1, 2).reshape(b, s, -1))


class QuantizedLlamaMLP(QuantizedModule, LlamaMLP):
    """Reference mlp.py:19-40."""

    def __init_quantization__(self) -> None:
        super().__init_quantization__()
        self.input_quantizer = QuantizerStub(input_quantizer=True)
        self.gate_act_quantizer = QuantizerStub(input_quantizer=True)
        self.gated_up_proj_output_quantizer = QuantizerStub(output_quantizer=True)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        x = self.input_quantizer(x)
        if _untouched(self.gate_act_quantizer, self.gated_up_proj_output_quantizer):
            product = _weight_only_gate_up(x, self.gate_proj, self.up_proj) if type(x) is torch.Tensor else None
            if product is not None:
                return self.down_proj(product)
            quantized, parts = _w8a8_gate_up_down_input(x, self.gate_proj, self.up_proj, self.down_proj) if type(x) is torch.Tensor else (None, None)
            if quantized is not None:  # down_proj's GEMM on the codes its input quantizer would have produced
                down = self.down_proj
                return ff.nn.functional.linear(quantized, down.weight_quantizer(down.weight), None, output_quantizer=down.output_quantizer)
            if parts is None and type(x) is torch.Tensor and _estimating(getattr(self.down_proj, "input_quantizer", None)):
                # range estimation: the product from one op whatever the two input estimators hold (decided on the device), its
                # [min, max] handed to down_proj's input estimator
                with sibling_quantizers(undecided=_codes_stay_with_the_linears(self.gate_proj, self.up_proj, module_calls=True)):
                    product, pair, pre = _w8a8_gate_up_product(x, self.gate_proj, self.up_proj, want_extrema=True)
                    if product is not None:
                        RECENT.remember_extrema(product, pair)
                        return self.down_proj(product)
                    if pre is not None:  # the quantizers have run: finish the two linears from what they returned
                        parts = tuple(ff.nn.functional.linear(xq, wq, None, output_quantizer=lin.output_quantizer)
                                      for (xq, wq), lin in zip(pre, (self.gate_proj, self.up_proj)))
            if parts is None:
                with sibling_quantizers(undecided=_codes_stay_with_the_linears(self.gate_proj, self.up_proj, module_calls=True)):
                    parts = (self.gate_proj(x), self.up_proj(x))
            gate, up = parts
            if _one_pass(gate, up) and gate.shape == up.shape:
                return self.down_proj(ff.ops.silu_mul_quantize(gate, up, (), want_product=True)[0])  # silu(gate) * up, one pass
            return self.down_proj(F.silu(gate) * up)
        gated = self.gated_up_proj_output_quantizer(self.gate_act_quantizer(F.silu(self.gate_proj(x))) * self.up_proj(x))
        return self.down_proj(gated)


class QuantizedLlamaDecoderLayer(QuantizedModule, LlamaDecoderLayer):
    """Reference decoder.py:19-90."""

    def __init_quantization__(self) -> None:
        super().__init_quantization__()
        self.input_quantizer = QuantizerStub(input_quantizer=True)
        self.attn_res_act_quantizer = QuantizerStub(output_quantizer=True)
        self.mlp_res_act_quantizer = QuantizerStub(output_quantizer=True)

    def forward(self, hidden_states: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor) -> torch.Tensor:
        return self.forward_deferred(hidden_states, None, cos, sin, defer=False)[0]

    def accepts_pending(self, hidden_states: torch.Tensor) -> bool:
        """True when this layer can take (residual stream, a term not yet added to it): its entry slot is an untouched stub
        and its first RMSNorm runs as the one-pass kernel, which then performs the add."""
        norm = self.input_layernorm
        return (_untouched(self.input_quantizer) and isinstance(norm, QuantizedLlamaRMSNorm) and hidden_states.is_contiguous() and norm._fusable(hidden_states)
                and not _hooked(self, norm, self.input_quantizer))

    def forward_deferred(self, hidden_states: torch.Tensor, pending: torch.Tensor | None, cos: torch.Tensor, sin: torch.Tensor,
                         defer: bool) -> tuple[torch.Tensor, torch.Tensor | None]:
        """The layer on a residual stream whose last term may still be pending (`hidden_states + pending` is the value; the
        caller checked :meth:`accepts_pending`). Returns (stream, pending'): with `defer` and an untouched output slot the
        MLP's output is handed on un-added — the next layer's first RMSNorm launch (or the model's final one) adds it —
        else pending' is None. Between QuantizedLlamaModel and its layers only; ``forward`` is the ordinary entry."""
        if pending is not None:
            norm = self.input_layernorm
            hidden_states, normed, _ = ff.ops.add_rmsnorm_quantize(hidden_states, pending, norm.weight, norm.variance_epsilon, (), want_sum=True,
                                                                   want_norm=True, sum_inplace=True)
        else:
            hidden_states = self.input_quantizer(hidden_states)
            normed = self.input_layernorm(hidden_states)
        attn_out = self.self_attn(normed, cos, sin)
        norm = self.post_attention_layernorm
        if (_untouched(self.attn_res_act_quantizer) and isinstance(norm, QuantizedLlamaRMSNorm) and hidden_states.shape == attn_out.shape
                and hidden_states.is_contiguous() and attn_out.is_contiguous() and norm._fusable(hidden_states) and _one_pass(attn_out)
                and not _hooked(norm, self.attn_res_act_quantizer)):
            # the residual add and the RMSNorm behind it as one pass (both values are needed: the sum is the next residual)
            hidden_states, normed, _ = ff.ops.add_rmsnorm_quantize(hidden_states, attn_out, norm.weight, norm.variance_epsilon, (), want_sum=True, want_norm=True)
        else:
            hidden_states = self.attn_res_act_quantizer(hidden_states + attn_out)
            normed = norm(hidden_states)
        mlp_out = self.mlp(normed)
        # the stream handed on must itself be what the consumer's one-pass launch takes (a set attn_res_act_quantizer makes it a
        # QuantizedTensor; `consumer` was decided on this layer's INPUT stream)
        if (defer and _untouched(self.mlp_res_act_quantizer) and not _hooked(self.mlp_res_act_quantizer) and _one_pass(mlp_out, hidden_states)
                and mlp_out.shape == hidden_states.shape and mlp_out.is_contiguous() and hidden_states.is_contiguous()):
            return hidden_states, mlp_out
        return self.mlp_res_act_quantizer(hidden_states + mlp_out), None


class QuantizedLlamaModel(QuantizedModule, LlamaModel):
    def forward(self, input_ids: torch.Tensor, logits: bool = True) -> torch.Tensor:
        """LlamaModel.forward; between quantized decoder layers whose boundary slots are untouched stubs the MLP's residual add is
        left to the next layer's first RMSNorm launch (``forward_deferred``): one pass instead of an eager add plus a norm."""
        hidden_states = self.embed_tokens(input_ids)
        cos, sin = rotary_tables(input_ids.shape[1], self.config.head_dim, self.config.rope_theta, hidden_states.device, hidden_states.dtype)
        layers = list(self.layers)
        pending: torch.Tensor | None = None
        for i, layer in enumerate(layers):
            # a layer with a user hook is CALLED (hooks fire, pending added first); forward_deferred is the hook-free shortcut
            if isinstance(layer, QuantizedLlamaDecoderLayer) and (pending is None or layer.accepts_pending(hidden_states)) and not _hooked(layer):
                nxt = layers[i + 1] if i + 1 < len(layers) else None
                if nxt is None:
                    consumer = (isinstance(self.norm, QuantizedLlamaRMSNorm) and type(hidden_states) is torch.Tensor and hidden_states.is_contiguous()
                                and self.norm._fusable(hidden_states) and not _hooked(self.norm))
                else:
                    consumer = isinstance(nxt, QuantizedLlamaDecoderLayer) and type(hidden_states) is torch.Tensor and nxt.accepts_pending(hidden_states)
                hidden_states, pending = layer.forward_deferred(hidden_states, pending, cos, sin, defer=consumer)
            else:
                hidden_states = layer(hidden_states if pending is None else hidden_states + pending, cos, sin)
                pending = None
        if pending is not None:
            hidden_states = ff.ops.add_rmsnorm_quantize(hidden_states, pending, self.norm.weight, self.norm.variance_epsilon, (), want_sum=False, want_norm=True)[1]
        else:
            hidden_states = self.norm(hidden_states)
        return self.lm_head(hidden_states) if logits else hidden_states


class QuantizedEmbedding(QuantizedModule, torch.nn.Embedding):
    """Embedding stays float in the Llama recipe (not matched by the quantizer queries)."""


# ---- recipe ------------------------------------------------------------------------------------------
def load_hf_state_dict(model: LlamaModel, weights: dict[str, torch.Tensor]) -> None:
    """Load a Hugging Face ``LlamaForCausalLM`` state dict (keys ``model.layers.N...``, ``lm_head.weight``)."""
    own = model.state_dict()
    with torch.no_grad():
        for key, value in weights.items():
            name = key.removeprefix("model.")
            if name in own:
                own[name].copy_(value.to(own[name].dtype))
            elif "rotary" not in name:
                raise KeyError(key)


def decoder_linears(model: LlamaModel):
    """(name, QuantizedLinear) for the 7 linears of every decoder layer — what the reference's queries
    ``**/layers/*/self_attn/*`` and ``**/layers/*/mlp/*`` select; lm_head is not matched."""
    for li, layer in enumerate(model.layers):
        for group, names in (("self_attn", ("q_proj", "k_proj", "v_proj", "o_proj")), ("mlp", ("gate_proj", "up_proj", "down_proj"))):
            for name in names:
                yield f"layers.{li}.{group}.{name}", getattr(getattr(layer, group), name)


def build_model(config: LlamaConfig, device: torch.device | str, dtype: torch.dtype = torch.bfloat16, seed: int = 1234, std: float = 0.02) -> LlamaModel:
    """Random-init model with N(0, std^2) weights, created directly on `device`."""
    with torch.device(device):
        model = LlamaModel(config).to(dtype)
    gen = torch.Generator(device=device).manual_seed(seed)
    with torch.no_grad():
        for p in model.parameters():
            if p.dim() >= 2:
                p.normal_(0.0, std, generator=gen)
    return model.eval()


def quantize_llama(
    model: LlamaModel,
    w_bits: int | None = 8,
    a_bits: int | None = 8,
    quantized_dtype: torch.dtype | None = torch.int8,
    weight_granularity: ff.quantization.granularity.Granularity | None = None,
) -> LlamaModel:
    """``ff.quantize_model`` + the quick-start W{w_bits}A{a_bits} recipe. Quantizers are created on the
    model's device; ranges are uninitialised until a calibration pass (``ff.estimate_ranges``)."""
    device = next(model.parameters()).device
    ff.quantize_model(model)
    for _, linear in decoder_linears(model):
        if w_bits is not None:
            linear.weight_quantizer = ff.nn.LinearQuantizer(
                w_bits, granularity=weight_granularity or ff.PerChannel(0), quantized_dtype=quantized_dtype, device=device
            )
        if a_bits is not None:
            linear.input_quantizer = ff.nn.LinearQuantizer(
                a_bits, symmetric=False, granularity=ff.PerTensor(), quantized_dtype=quantized_dtype, device=device
            )
    return model


def calibrate(model: LlamaModel, batches, sync_free: bool = True, disable_quantization: bool = False, fused: bool = False) -> None:
    """RunningMinMax calibration over `batches` of token ids (reference quick-start :193,255). ``fused`` runs the
    producers between the quantizers as one-pass kernels (:class:`FusedProducersForward`)."""
    forward = FusedProducersForward(model) if fused else (lambda ids: model(ids, logits=False))
    with torch.no_grad(), ff.strict_quantization(False):
        with ff.estimate_ranges(model, ff.range_setting.running_minmax, sync_free=sync_free, disable_quantization=disable_quantization):
            for ids in batches:
                forward(ids)


def attention_kernel_covers(cfg: LlamaConfig, seq_len: int, dtype: torch.dtype) -> bool:
    """Shapes ``ops.attention`` (csrc/ffq_attention.hip) is built for; everything else takes torch's SDPA on the device."""
    return cfg.head_dim == 128 and seq_len % 64 == 0 and seq_len > 1 and dtype == torch.bfloat16 and cfg.attention == "sdpa"


def _sdpa(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, cfg: LlamaConfig, b: int, s: int) -> torch.Tensor:
    d = cfg.head_dim
    return F.scaled_dot_product_attention(
        q.view(b, s, cfg.num_heads, d).transpose(1, 2), k.view(b, s, cfg.num_kv_heads, d).transpose(1, 2),
        v.view(b, s, cfg.num_kv_heads, d).transpose(1, 2), is_causal=s > 1, enable_gqa=cfg.num_kv_heads != cfg.num_heads,
    ).transpose(1, 2).reshape(b, s, -1)


class FusedForward:
    """Inference forward of a calibrated W8A8 Llama with A1 fused into the kernels that produce the
    quantized linears' inputs.

    The module graph above follows the reference helpers op for op: every ``QuantizedLinear`` quantizes
    its own input (reference nn/linear.py:33) and RMSNorm / rotary / SiLU*up run as eager ATen chains
    between the hot-path calls. This runner executes the same computation with the producers of
    ``fastforward_amd/csrc/ffq_producers.hip``: residual add + RMSNorm + the q/k/v (or gate/up) input
    quantizers in one pass, SiLU(gate)*up + the down_proj input quantizer in one pass, rotary embedding
    in place on the projections. Weight quantizers still run on every call (reference nn/linear.py:34)
    unless ``cache_weight_codes`` is set, in which case codes are kept per ``(weight, scale, offset)``
    version (weights of a calibrated inference model do not change).

    It refuses anything it cannot reproduce exactly — active quantizer overrides (i.e. a running
    ``estimate_ranges``), non-stub activation quantizers between the linears, dynamic quantizers,
    per-channel activations, biases — instead of silently computing something else.
    """

    def __init__(self, model: LlamaModel, cache_weight_codes: bool = False, fuse_mlp: bool = True, fuse_attention: bool = True, fuse_rowsums: bool = False,
                 batch_weight_quantization: bool = True, batch_rowsums: bool = False, just_in_time_weights: bool = True, qkv_one_launch: bool = True) -> None:
        problems = self.unsupported(model)
        if problems:
            raise ff.exceptions.QuantizationError("FusedForward cannot run this model: " + "; ".join(problems[:4]))
        self.model = model
        self.cache_weight_codes = cache_weight_codes
        # gate_proj + up_proj + SiLU*up + the down_proj input quantizer in one launch (ops.mlp_gate_up_w8a8) where
        # both projections see the same activation codes and carry symmetric (zero-offset) weight quantizers
        self.fuse_mlp = fuse_mlp
        # attention + the o_proj input quantizer in one launch (ops.attention) where the kernel covers the shape
        self.fuse_attention = fuse_attention
        # weight codes and the row sums of the zero-point term in one pass (ops.quantize_rows_rowsum) where the kernel applies.
        # OFF by default: in isolation the one-pass kernel is 5 us per weight cheaper than A1 + the reduction launch
        # (tools/rowsum_time.py), but the Llama-3-8B forward measures 1.2 % SLOWER with it (tools/ab_forward.py, same box, 3
        # runs): the int8 GEMMs that follow run slower by more than the reduction cost (tools/gemm_cache_probe.py: a GEMM
        # preceded by more memory-bound work is faster — the chip is power-limited and the light pass lets it recover).
        self.fuse_rowsums = fuse_rowsums
        # the seven weights of a layer re-quantized by one launch instead of seven (ops.quantize_rows_batch)
        self.batch_weight_quantization = batch_weight_quantization
        # ... which can also leave the row sums of the codes (the int8 GEMM's zero-point term) beside them: no rowsum_i8_kernel
        # launch per linear (224 per forward, 2.1 ms of kernel time). OFF by default: the Llama-3-8B forward measures 2.9 % SLOWER
        # with it (125.3 vs 121.9 ms, two alternating runs of bench.py on one box in round 4) — the same direction as round 2's
        # one-pass codes + row sums per weight (`fuse_rowsums`): the int8 GEMMs behind a heavier quantization launch run slower by
        # more than the 224 small launches cost (the chip is power-limited in the GEMMs; tools/gemm_cache_probe.py)
        self.batch_rowsums = batch_rowsums
        # Round 5, what the two measurements above really show (rocprofv3 per kernel, profiles/r05_rowsum_ab.md): the quantization
        # launches are NOT heavier with the row sums (100.0 against 100.5 us) — the int8 GEMMs behind them are slower, the gate+up
        # launch by 11.6 % (1584 against 1420 us), because a rowsum_i8 launch right before a GEMM reads the weight codes and thereby
        # brings them into the 256 MiB Infinity Cache: the GEMM's L2 misses on the weight panels are then served from there instead
        # of HBM (the codes written by one launch for the whole layer are long evicted by the time down_proj runs). So the weights
        # are re-quantized JUST IN TIME, group by group, right before the GEMMs that read them — q/k/v as one launch that leaves
        # their row sums, o_proj and down_proj each by the one-pass codes + row sums kernel (codes hot, no reduction launch: 5 of
        # the 7 rowsum_i8 launches per layer gone) — while gate/up (117 MB of codes behind 235 MB of bf16 reads: more than the cache
        # keeps) stay one launch followed by their two rowsum_i8 launches, which double as the prefetch.
        self.just_in_time_weights = just_in_time_weights and batch_weight_quantization and not fuse_rowsums and not batch_rowsums
        # q_proj / k_proj / v_proj as ONE launch of the int8 GEMM on the code tensor their input quantizers share (round 6; ops.linear_w8a8_multi):
        # the just-in-time re-quantization of the three weights writes their codes and row sums side by side into one buffer; k / v (four
        # column tiles per row tile) no longer run alone, and a forward issues 64 launches fewer
        self.qkv_one_launch = qkv_one_launch and self.just_in_time_weights
        self._qkv_side: dict[int, tuple[torch.Tensor, torch.Tensor | None]] = {}  # layer -> (codes [Nq + Nk + Nv, K], row sums) of THIS forward
        self._qkv_scales: dict[int, tuple[tuple[int, ...], torch.Tensor]] = {}     # layer -> (scale versions, the three scale vectors as one)
        self._layer_rowsums: dict[int, torch.Tensor] = {}
        self._jit_single: set[int] = set()  # o_proj / down_proj of the layer in flight: one-pass codes + row sums right before their GEMM
        self._layer_codes: dict[int, torch.Tensor] = {}
        self._zero_offset: dict[int, tuple[int, bool]] = {}
        self._weight_cache: dict[int, tuple[tuple[int, int, int], tuple[torch.Tensor, torch.Tensor | None]]] = {}
        self._rowsum_rows = sum(linear.weight.shape[0] for _, linear in decoder_linears(model))
        self._rowsum_pool: torch.Tensor | None = None
        self._rowsum_used = 0
        # when set to a list, every int8 GEMM launch appends (output rows of weight processed, K, start event, end event)
        # — the fused gate/up launch counts both matrices: bench.py times the GEMM launches of a real forward with it
        self.linear_events: list[tuple[int, int, torch.cuda.Event, torch.cuda.Event]] | None = None
        # One host comparison per layer: consumers of the same tensor usually hold the same range. The table is keyed on the
        # version counters of the input quantizers' parameters and rebuilt when a range is set again after construction
        # (the range setter bumps them, nn/linear_quantizer.py::_write_parameters_for_range).
        self._fan: list[dict[str, tuple[list[tuple[torch.Tensor, torch.Tensor | None]], list[int]]]] = []
        self._fan_signature: tuple[int, ...] = ()
        self.refresh()

    def _input_signature(self) -> tuple[int, ...]:
        sig: list[int] = []
        for _, linear in decoder_linears(self.model):
            q = linear.input_quantizer
            sig += [id(q.scale), q.scale._version, -1 if q.offset is None else q.offset._version]
        return tuple(sig)

    def refresh(self) -> None:
        """Re-read everything this object derived from quantizer parameters on the host: which of q/k/v and gate/up share
        an input range, and which weight offsets are all zero. Called at construction and, automatically, by the next
        forward after any input-quantizer parameter changed (never inside a hipGraph capture: the reads synchronise)."""
        model = self.model
        self._fan = []
        for layer in model.layers:
            attn, mlp = layer.self_attn, layer.mlp
            self._fan.append({
                "qkv": self._distinct([attn.q_proj, attn.k_proj, attn.v_proj]),
                "gate_up": self._distinct([mlp.gate_proj, mlp.up_proj]),
            })
        for _, linear in decoder_linears(model):  # host reads happen here, never inside a (possibly graph-captured) forward
            self._symmetric_weights(linear)
        self._fan_signature = self._input_signature()

    @staticmethod
    def _params(linear: torch.nn.Module) -> tuple[torch.Tensor, torch.Tensor | None]:
        q = linear.input_quantizer
        return q.scale, q.offset

    @classmethod
    def _distinct(cls, linears: list[torch.nn.Module]) -> tuple[list[tuple[torch.Tensor, torch.Tensor | None]], list[int]]:
        """(distinct parameter pairs, index of each linear's pair)."""
        pairs: list[tuple[torch.Tensor, torch.Tensor | None]] = []
        index: list[int] = []
        for linear in linears:
            s, o = cls._params(linear)
            for i, (ps, po) in enumerate(pairs):
                if torch.equal(ps, s) and ((po is None and o is None) or (po is not None and o is not None and torch.equal(po, o))):
                    index.append(i)
                    break
            else:
                pairs.append((s, o))
                index.append(len(pairs) - 1)
        return pairs, index

    @staticmethod
    def unsupported(model: LlamaModel) -> list[str]:
        from fastforward_amd.nn import LinearQuantizer

        problems: list[str] = []
        cfg = model.config
        if cfg.hidden_size % 16 or cfg.hidden_size > 8192 or cfg.head_dim % 16 or cfg.intermediate_size % 16:
            problems.append("hidden size / head dim outside the fused kernels' range")
        if next(model.parameters()).dtype != torch.bfloat16:
            problems.append("fused producers are built for bf16 models")
        for name, q in ff.nn.named_quantizers(model, skip_stubs=False):
            if q is None:
                continue
            if next(iter(q.overrides), None) is not None:
                problems.append(f"{name}: an override is active (range estimation running?)")
            is_linear_slot = any(name.endswith(f"{p}.{slot}") for p in ("q_proj", "k_proj", "v_proj", "o_proj", "gate_proj", "up_proj", "down_proj") for slot in ("input_quantizer", "weight_quantizer")) and ".layers." in "." + name
            if not is_linear_slot:
                if not q.is_stub():
                    problems.append(f"{name}: only the decoder linears' input/weight quantizers may be set")
                continue
            if not isinstance(q, LinearQuantizer) or q.has_uninitialized_params:
                problems.append(f"{name}: needs an initialised LinearQuantizer")
                continue
            if q.num_bits > 8 or int(q.num_bits) != q.num_bits:
                problems.append(f"{name}: more than 8 bits")
            if name.endswith("input_quantizer") and (not q.per_tensor or q.scale.numel() != 1):
                problems.append(f"{name}: activation quantizers must be per-tensor")
            if name.endswith("weight_quantizer") and q.quantized_dtype is not torch.int8:
                problems.append(f"{name}: weight codes must be stored as int8")
        for name, linear in decoder_linears(model):
            if linear.bias is not None:
                problems.append(f"{name}: bias")
            wq = linear.weight_quantizer
            if isinstance(wq, LinearQuantizer) and _weight_row_mode(linear) is None:
                # the int8 GEMM reads N scales as one per OUTPUT row: PerChannel(1) on a square weight has N parameters too
                problems.append(f"{name}: weight granularity must be per tensor or per output channel (tile {tuple(wq.granularity.tile_size(linear.weight.shape))})")
        return problems

    def _symmetric_weights(self, linear: torch.nn.Module) -> bool:
        """True if the weight quantizer's offset is absent or all zero — one host read per offset version (the offset of
        a symmetric quantizer is a zero buffer unless its weights are one-sided)."""
        offset = linear.weight_quantizer.offset
        if offset is None:
            return True
        hit = self._zero_offset.get(id(linear))
        if hit is None or hit[0] != offset._version:
            hit = (offset._version, not bool(offset.any()))
            self._zero_offset[id(linear)] = hit
        return hit[1]

    def _rowsum_slice(self, rows: int, device: torch.device) -> torch.Tensor | None:
        """`rows` zeroed int32 entries of the per-forward pool (one fill for all the weights of a forward), or None."""
        pool, used = self._rowsum_pool, self._rowsum_used
        if pool is None or used + rows > pool.numel() or pool.device != device:
            return None
        self._rowsum_used = used + rows
        return pool[used:used + rows]

    def _quantize_weight(self, linear: torch.nn.Module) -> tuple[torch.Tensor, torch.Tensor | None]:
        """(int8 codes, int32 row sums of the codes or None) — A1 of the weight as the weight quantizer computes it
        (reference nn/linear.py:34). Per-output-channel quantizers on a bf16 weight take the one-pass kernel that leaves the
        row sums of the zero-point term beside the codes (ops.quantize_rows_rowsum); anything else goes through the
        quantizer's own forward and the GEMM entry point reduces the codes itself."""
        wq = linear.weight_quantizer
        rows = linear.weight.shape[0]
        ready = self._layer_codes.pop(id(linear), None)  # quantized with the rest of its layer in one launch (_quantize_layer)
        if ready is not None:
            return ready, self._layer_rowsums.pop(id(linear), None)
        if (self.fuse_rowsums or (self.just_in_time_weights and id(linear) in self._jit_single)) and _weight_row_mode(linear) == "row" and rows > 1 and (wq.offset is None or wq.offset.numel() == rows) and wq.num_bits <= 8:
            offset = None if self._symmetric_weights(linear) else wq.offset  # an all-zero offset buffer: same codes
            fused = ff.ops.quantize_rows_rowsum(linear.weight, wq.scale, offset, wq.num_bits, rowsum_out=self._rowsum_slice(rows, linear.weight.device))
            if fused is not None:
                return fused
        return wq(linear.weight).raw_data, None

    def _quantize_layer(self, layer: torch.nn.Module, group: tuple[torch.nn.Module, ...] | None = None, with_rowsums: bool | None = None) -> None:
        """A1 of the layer's weights that are re-quantized this forward (`group`: some of them, right before the GEMMs that read
        them — `just_in_time_weights`), as ONE launch (ops.quantize_rows_batch): seven launches otherwise, two of them (k_proj /
        v_proj) too short to reach the streaming rate on their own. Fills `_layer_codes`; weights the batched kernel does not cover
        (not per-channel bf16, odd sizes) and one-member groups are left to `_quantize_weight`."""
        if group is None:
            self._layer_codes, self._layer_rowsums = {}, {}
        if not self.batch_weight_quantization or self.fuse_rowsums:
            return
        attn, mlp = layer.self_attn, layer.mlp
        if group is None and self.just_in_time_weights:
            return  # group by group from __call__
        want_rowsums = self.batch_rowsums if with_rowsums is None else with_rowsums
        todo = []
        for linear in (group if group is not None else (attn.q_proj, attn.k_proj, attn.v_proj, attn.o_proj, mlp.gate_proj, mlp.up_proj, mlp.down_proj)):
            wq = linear.weight_quantizer
            if self.cache_weight_codes:
                key = (linear.weight._version, wq.scale._version, -1 if wq.offset is None else wq.offset._version)
                hit = self._weight_cache.get(id(linear))
                if hit is not None and hit[0] == key:
                    continue
            rows = linear.weight.shape[0]
            # per OUTPUT channel by its tiling, not by its parameter count (PerChannel(1) on a square weight also has `rows` scales)
            if _weight_row_mode(linear) == "row" and rows > 1 and (wq.offset is None or wq.offset.numel() == rows) and wq.num_bits == attn.q_proj.weight_quantizer.num_bits:
                todo.append(linear)
        if len(todo) < 2:
            return
        offsets = [None if self._symmetric_weights(l) else l.weight_quantizer.offset for l in todo]  # an all-zero offset buffer: same codes
        # q / k / v re-quantized together: their codes (and row sums) side by side in ONE buffer, so that the three linears can run as one
        # launch of the int8 GEMM (`_qkv_one_launch`); the per-linear entries are views of it
        together = (group is not None and len(todo) == 3 and tuple(todo) == (attn.q_proj, attn.k_proj, attn.v_proj) and self.qkv_one_launch
                    and len({l.weight.shape[1] for l in todo}) == 1 and all(l.weight.shape[0] % 256 == 0 for l in todo[:2]))
        sums, pool_slice = None, None
        if want_rowsums and all(l.weight.shape[1] % 1024 == 0 for l in todo):
            if together:
                pool_slice = self._rowsum_slice(sum(l.weight.shape[0] for l in todo), todo[0].weight.device)
                sums = None if pool_slice is None else list(torch.split(pool_slice, [l.weight.shape[0] for l in todo]))
            else:
                sums = [self._rowsum_slice(l.weight.shape[0], l.weight.device) for l in todo]
                sums = sums if all(t is not None for t in sums) else None
        codes_out, cat = None, None
        if together:
            cat = torch.empty((sum(l.weight.shape[0] for l in todo), todo[0].weight.shape[1]), dtype=torch.int8, device=todo[0].weight.device)
            codes_out = list(torch.split(cat, [l.weight.shape[0] for l in todo]))
        codes = ff.ops.quantize_rows_batch([l.weight for l in todo], [l.weight_quantizer.scale for l in todo], offsets, todo[0].weight_quantizer.num_bits,
                                           rowsums=sums, codes_out=codes_out)
        if codes is None and sums is not None:
            codes, sums = ff.ops.quantize_rows_batch([l.weight for l in todo], [l.weight_quantizer.scale for l in todo], offsets,
                                                     todo[0].weight_quantizer.num_bits, codes_out=codes_out), None
        if codes is not None and together:
            self._qkv_side[id(layer)] = (cat, pool_slice if sums is not None else None)
        if codes is not None:
            self._layer_codes.update({id(l): c for l, c in zip(todo, codes)})
            if sums is not None:
                self._layer_rowsums.update({id(l): t for l, t in zip(todo, sums)})

    def _weight(self, linear: torch.nn.Module) -> tuple[torch.Tensor, torch.Tensor | None, torch.Tensor, torch.Tensor | None]:
        """(int8 codes, row sums or None, scale, offset) of the linear's weight."""
        wq = linear.weight_quantizer
        if self.cache_weight_codes:
            key = (linear.weight._version, wq.scale._version, -1 if wq.offset is None else wq.offset._version)
            hit = self._weight_cache.get(id(linear))
            if hit is not None and hit[0] == key:
                return hit[1][0], hit[1][1], wq.scale, wq.offset
            produced = self._quantize_weight(linear)
            self._weight_cache[id(linear)] = (key, produced)
            return produced[0], produced[1], wq.scale, wq.offset
        codes, rowsum = self._quantize_weight(linear)
        return codes, rowsum, wq.scale, wq.offset

    def _qkv_one_launch(self, layer: torch.nn.Module, codes: Sequence[torch.Tensor], index: Sequence[int]) -> list[torch.Tensor] | None:
        """q_proj, k_proj and v_proj on the one code tensor their (equal) input quantizers share, as ONE launch of the int8 GEMM
        (ops.linear_w8a8_multi; reference nn/linear.py:32-39 three times): possible when this forward has just re-quantized the three
        weights into one buffer (`_quantize_layer` with the q / k / v group) and none of them carries a live offset. Bit for bit the three
        separate launches; None: the caller runs them."""
        side = self._qkv_side.pop(id(layer), None)
        attn = layer.self_attn
        linears = (attn.q_proj, attn.k_proj, attn.v_proj)
        if side is None or not all(self._symmetric_weights(l) for l in linears):
            return None
        w_codes, w_rowsum = side
        scales = [l.weight_quantizer.scale for l in linears]
        key = tuple(t._version for t in scales)
        hit = self._qkv_scales.get(id(layer))
        if hit is None or hit[0] != key:
            if w_codes.is_cuda and torch.cuda.is_current_stream_capturing():
                return None  # (the concatenated scales are made outside a capture: run the shape once before capturing)
            hit = (key, torch.cat([t.detach().reshape(-1).to(torch.float32) for t in scales]))
            self._qkv_scales[id(layer)] = hit
        x_scale, x_offset = self._params(attn.q_proj)
        rows = [l.weight.shape[0] for l in linears]
        if self.linear_events is not None:
            start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            start.record()
        outs = ff.ops.linear_w8a8_multi(codes[index[0]], w_codes, x_scale, x_offset, hit[1], rows, out_dtype=torch.bfloat16, w_rowsum=w_rowsum)
        if outs is None:
            # (the separate launches find their codes where `_quantize_layer` left them: views of the same buffer)
            return None
        for l in linears:  # consumed: the per-linear entries of this forward are not needed any more
            self._layer_codes.pop(id(l), None)
            self._layer_rowsums.pop(id(l), None)
        if self.linear_events is not None:
            end.record()
            self.linear_events.append((sum(rows), w_codes.shape[1], start, end))
        return outs

    def _linear(self, x_codes: torch.Tensor, linear: torch.nn.Module) -> torch.Tensor:
        """The int8 GEMM of one quantized linear (the RMSNorm kernel that follows adds its output to the residual stream)."""
        w_codes, w_rowsum, w_scale, w_offset = self._weight(linear)
        if w_offset is not None and self._symmetric_weights(linear):
            w_offset = None  # an all-zero offset buffer: same result without the GEMM's device-side offset check
        x_scale, x_offset = self._params(linear)
        if self.linear_events is None:
            return ff.ops.linear_w8a8(x_codes, w_codes, x_scale, x_offset, w_scale, w_offset, None, out_dtype=torch.bfloat16, w_rowsum=w_rowsum)
        start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start.record()
        out = ff.ops.linear_w8a8(x_codes, w_codes, x_scale, x_offset, w_scale, w_offset, None, out_dtype=torch.bfloat16, w_rowsum=w_rowsum)
        end.record()
        self.linear_events.append((w_codes.shape[0], w_codes.shape[1], start, end))
        return out

    @torch.no_grad()
    def __call__(self, input_ids: torch.Tensor, logits: bool = True) -> torch.Tensor:
        model, cfg = self.model, self.model.config
        b, s = input_ids.shape
        d = cfg.head_dim
        if self._input_signature() != self._fan_signature:  # a range was set again since the tables were built
            if input_ids.is_cuda and torch.cuda.is_current_stream_capturing():
                raise ff.exceptions.QuantizationError("quantizer ranges changed after FusedForward was built: call refresh() before capturing a graph")
            self.refresh()
        hidden = model.embed_tokens(input_ids)
        cos, sin = rotary_tables(s, d, cfg.rope_theta, hidden.device, hidden.dtype)
        pending: torch.Tensor | None = None  # down_proj's output: the next RMSNorm launch adds it to the residual stream
        if self.fuse_rowsums or self.just_in_time_weights or (self.batch_rowsums and self.batch_weight_quantization):
            # the weight row sums of this forward: one zero fill, slices handed out as the weights are quantized
            self._rowsum_pool = torch.zeros(self._rowsum_rows, dtype=torch.int32, device=hidden.device)
            self._rowsum_used = 0
        for layer, fan in zip(model.layers, self._fan):
            attn, mlp = layer.self_attn, layer.mlp
            self._quantize_layer(layer)
            pairs, index = fan["qkv"]
            bits = attn.q_proj.input_quantizer.num_bits
            hidden, _, codes = ff.ops.add_rmsnorm_quantize(
                hidden, pending, layer.input_layernorm.weight, layer.input_layernorm.variance_epsilon, pairs, bits, sum_inplace=pending is not None
            )
            if self.just_in_time_weights:
                self._jit_single = {id(attn.o_proj), id(mlp.down_proj)}
                self._quantize_layer(layer, (attn.q_proj, attn.k_proj, attn.v_proj), with_rowsums=True)
            qkv = self._qkv_one_launch(layer, codes, index) if index[0] == index[1] == index[2] else None
            if qkv is not None:
                q, k, v = qkv
            else:
                q = self._linear(codes[index[0]], attn.q_proj)
                k = self._linear(codes[index[1]], attn.k_proj)
                v = self._linear(codes[index[2]], attn.v_proj)
            o_in = attn.o_proj.input_quantizer
            if self.fuse_attention and attention_kernel_covers(cfg, s, q.dtype):
                # rotary embedding: k in place, q as the attention launch loads it (no pass over the q projection); softmax(q k^T) v
                # and o_proj's input quantizer in that one launch: the bf16 context never visits HBM
                ff.ops.rope_(None if FUSE_Q_ROPE else q, k, cos, sin, d)
                _, o_codes = ff.ops.attention(q, k, v, d, causal=s > 1, quantizer=(o_in.scale, o_in.offset), num_bits=o_in.num_bits, want_context=False,
                                              q_rope=(cos, sin) if FUSE_Q_ROPE else None)
            else:
                ff.ops.rope_(q, k, cos, sin, d)
                ctx = _sdpa(q, k, v, cfg, b, s)
                o_codes = ff.ops.quantize_by_tile(ctx, o_in.scale, ctx.shape, o_in.num_bits, torch.int8, o_in.offset)
            attn_out = self._linear(o_codes, attn.o_proj)
            pairs, index = fan["gate_up"]
            hidden, _, codes = ff.ops.add_rmsnorm_quantize(
                hidden, attn_out, layer.post_attention_layernorm.weight, layer.post_attention_layernorm.variance_epsilon,
                pairs, mlp.gate_proj.input_quantizer.num_bits, sum_inplace=True,
            )
            d_in = mlp.down_proj.input_quantizer
            d_codes = None
            if self.just_in_time_weights:
                self._quantize_layer(layer, (mlp.gate_proj, mlp.up_proj), with_rowsums=False)
            if self.fuse_mlp and index[0] == index[1] and self._symmetric_weights(mlp.gate_proj) and self._symmetric_weights(mlp.up_proj):
                g_codes, g_rowsum, g_scale, _ = self._weight(mlp.gate_proj)
                u_codes, u_rowsum, u_scale, _ = self._weight(mlp.up_proj)
                x_scale, x_offset = self._params(mlp.gate_proj)
                if self.linear_events is not None:
                    start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    start.record()
                d_codes = ff.ops.mlp_gate_up_w8a8(codes[index[0]], g_codes, u_codes, x_scale, x_offset, g_scale, u_scale, d_in.scale, d_in.offset, d_in.num_bits,
                                                  gate_rowsum=g_rowsum, up_rowsum=u_rowsum)
                if self.linear_events is not None and d_codes is not None:
                    end.record()
                    self.linear_events.append((2 * g_codes.shape[0], g_codes.shape[1], start, end))
            if d_codes is None:
                gate = self._linear(codes[index[0]], mlp.gate_proj)
                up = self._linear(codes[index[1]], mlp.up_proj)
                _, (d_codes,) = ff.ops.silu_mul_quantize(gate, up, [(d_in.scale, d_in.offset)], d_in.num_bits)
            pending = self._linear(d_codes, mlp.down_proj)
        _, normed, _ = ff.ops.add_rmsnorm_quantize(hidden, pending, model.norm.weight, model.norm.variance_epsilon, (), want_sum=False, want_norm=True)
        if not logits:
            return normed
        with ff.strict_quantization(False):  # lm_head stays float in the recipe (quick-start :145)
            return model.lm_head(normed)


class FusedProducersForward:
    """The forward for every case ``FusedForward`` refuses: producers fused, quantizers untouched.

    Range estimation needs every quantizer's own ``forward`` (that is where ``estimate_ranges`` installs its
    override: update the running min/max, set the range, quantize — reference range_setting/common.py:218-238), and a
    weight-only model (BASELINE configs 2 and 4) has no activation codes to fuse into anything, so nothing is fused
    INTO a quantizer here. What is fused is everything between the linears: residual add + RMSNorm, the rotary
    embedding, SiLU*up and attention run as the one-pass kernels of csrc/ffq_producers.hip / ffq_attention.hip instead
    of eager ATen chains. A linear whose input and weight quantizers both produce int8 codes takes the codes straight to
    the int8 GEMM; any other linear runs its module forward (``QuantizedLinear.forward``: quantizers, dispatcher,
    float fallback on the dequantized operands — reference nn/linear.py:32-39). Results equal the module graph up to the
    summation order inside RMSNorm and the attention launch's tolerance (see FusedForward).
    """

    def __init__(self, model: LlamaModel, weight_storage: str = "requantize") -> None:
        cfg = model.config
        if cfg.hidden_size % 16 or cfg.hidden_size > 8192 or cfg.head_dim % 16 or cfg.intermediate_size % 16:
            raise ff.exceptions.QuantizationError("hidden size / head dim outside the fused kernels' range")
        if next(model.parameters()).dtype != torch.bfloat16:
            raise ff.exceptions.QuantizationError("fused producers are built for bf16 models")
        for name, q in ff.nn.named_quantizers(model, skip_stubs=False):
            is_linear_slot = ".layers." in "." + name and any(f"{p}." in name for p in ("q_proj", "k_proj", "v_proj", "o_proj", "gate_proj", "up_proj", "down_proj"))
            if q is not None and not is_linear_slot and not q.is_stub():
                raise ff.exceptions.QuantizationError(f"{name}: only the decoder linears' quantizers may be set (the producers between them are fused)")
        if weight_storage not in ("requantize", "codes", "packed"):
            raise ValueError("weight_storage is 'requantize', 'codes' or 'packed'")
        self.model = model
        # How a WEIGHT-ONLY linear (stub input quantizer, static LinearQuantizer on the weight: BASELINE configs 2 / 4) gets its
        # weight codes. "requantize": the weight quantizer runs on every call (reference nn/linear.py:34). "codes" / "packed":
        # the codes are kept per (weight, scale, offset) version — int8 containers, or for <= 4-bit weights the packed nibbles
        # of export/stages/gguf/_packing.py:44-53 (0.5 B per weight: config 4's storage), which the GEMM consumes as they are
        # (ops.linear_wq(pack_block=...)); frozen inference weights do not change between forwards.
        self.weight_storage = weight_storage
        self._stored: dict[int, tuple[tuple[int, int, int], torch.Tensor, int]] = {}
        self._product_extrema: tuple[torch.Tensor, torch.Tensor] | None = None  # (SiLU * up product, its [min, max]) of the layer in flight
        self._want_product_extrema = False

    def _stored_weight(self, linear: torch.nn.Module) -> tuple[torch.Tensor, int] | None:
        """(codes or packed nibbles, packing block) of a weight-only linear's weight under ``weight_storage``, or None when
        the linear is not of that kind (then its module forward runs)."""
        from fastforward_amd.nn import LinearQuantizer

        wq = linear.weight_quantizer
        if (self.weight_storage == "requantize" or type(wq) is not LinearQuantizer or wq.has_uninitialized_params or next(iter(wq.overrides), None) is not None
                or ff.fused_linear.KERNELS.weight_group(_Shaped(linear.weight, wq)) is None):
            return None
        key = (linear.weight._version, wq.scale._version, -1 if wq.offset is None else wq.offset._version)
        hit = self._stored.get(id(linear))
        if hit is not None and hit[0] == key:
            return hit[1], hit[2]
        group = ff.fused_linear.KERNELS.weight_group(_Shaped(linear.weight, wq))
        block = 0
        if self.weight_storage == "packed" and wq.num_bits == 4 and group % 32 == 0 and (group & (group - 1)) == 0:
            block = min(group, 128)  # A1 (4 bits) + the nibble packing in one pass: the int8 codes never exist
            tile = wq.granularity.tile_size(linear.weight.shape)
            stored = ff.ops.quantize_pack_int4(linear.weight, wq.scale, linear.weight.shape if isinstance(tile, str) else tile, wq.offset, block=block)
        else:
            stored = wq(linear.weight).raw_data
        if stored.dtype not in (torch.int8, torch.uint8):
            return None
        self._stored[id(linear)] = (key, stored, block)
        return stored, block

    def _linear(self, x: torch.Tensor, linear: torch.nn.Module, gate: torch.Tensor | None = None, pre: tuple | None = None) -> torch.Tensor:
        """QuantizedLinear.forward; int8 codes on both sides go to the int8 GEMM directly, stored weight codes of a weight-only
        linear to the bf16 x weight-code GEMM. With `gate` (the bf16 result of the MLP's gate projection) the result is
        silu(gate) * linear(x) — formed in the int8 GEMM's epilogue where that launch covers the shapes. `pre`: what the linear's
        input and weight quantizers returned, when the caller has run them already."""
        if gate is not None:
            product = self._gated(x, linear, gate, pre)
            return product if product is not None else ff.ops.silu_mul_quantize(gate, self._linear(x, linear, pre=pre), (), want_product=True)[0]
        if linear.bias is not None or linear.weight_quantizer.is_stub() or not linear.output_quantizer.is_stub():
            return linear(x)
        if linear.input_quantizer.is_stub():
            stored = self._stored_weight(linear) if x.dtype == torch.bfloat16 else None
            if stored is not None:
                wq = linear.weight_quantizer
                group = ff.fused_linear.KERNELS.weight_group(_Shaped(linear.weight, wq))
                if ff.fused_linear._WEIGHT_ONLY_KERNEL:
                    out = ff.ops.linear_wq(x, stored[0], wq.scale, wq.offset, group=group, pack_block=stored[1], out_dtype=x.dtype)
                    if out is not None:
                        return out
                # outside the kernel's coverage (or inside ``weight_only_kernel(False)``, the A/B arm): the reference's own route
                # from the stored codes — A2 (or unpack + A2), float GEMM
                tile = (1, group)
                if stored[1]:
                    weight = ff.ops.unpack_dequantize_int4(stored[0], wq.scale, linear.weight.shape, tile, wq.offset, block=stored[1], output_dtype=x.dtype)
                else:
                    weight = ff.ops.dequantize_by_tile(stored[0], wq.scale, tile, wq.offset, x.dtype)
                return F.linear(x, weight)
            return linear(x)
        xq, wq = pre if pre is not None else (linear.input_quantizer(x), linear.weight_quantizer(linear.weight))
        # codes of a sibling quantizer that may not have been written (``sibling_quantizers(undecided=True)``): the GEMM takes the
        # earlier sibling's codes along and reads whichever are in force; every other reader gets them settled first
        earlier = RECENT.earlier_of(xq)
        if not (isinstance(xq, ff.QuantizedTensor) and isinstance(wq, ff.QuantizedTensor)) or xq.raw_data.dtype != torch.int8 or wq.raw_data.dtype != torch.int8:
            if earlier is not None:
                RECENT.settle(xq)
            return ff.nn.functional.linear(xq, wq, None, output_quantizer=linear.output_quantizer)  # e.g. disable_quantization=True
        xp, wp = xq.quantization_context.quantization_params, wq.quantization_context.quantization_params
        if xp.scale.numel() != 1 or ff.fused_linear.KERNELS.row_mode(wq) is None:
            if earlier is not None:
                RECENT.settle(xq)
            return ff.nn.functional.linear(xq, wq, None, output_quantizer=linear.output_quantizer)
        # an all-zero offset buffer of a symmetric weight quantizer: known from an earlier call once its version is stable, else
        # recognised by the GEMM on the device (no host read while a range estimator rewrites the parameters on every step)
        w_offset = None if wp.offset is None or ff.fused_linear.known_zero_offset(wp.offset) else wp.offset
        if earlier is not None:
            out = ff.ops.linear_w8a8_earlier(xq.raw_data, earlier, wq.raw_data, xp.scale, xp.offset, wp.scale, w_offset, out_dtype=torch.bfloat16)
            if out is not None:
                return out
            RECENT.settle(xq)
        return ff.ops.linear_w8a8(xq.raw_data, wq.raw_data, xp.scale, xp.offset, wp.scale, w_offset, None, out_dtype=torch.bfloat16)

    def _gated(self, x: torch.Tensor, linear: torch.nn.Module, gate: torch.Tensor, pre: tuple | None = None) -> torch.Tensor | None:
        """silu(gate) * linear(x) as ONE launch of the int8 GEMM (ops.linear_w8a8_gated), or None: the quantizers run exactly as
        in ``_linear`` (range estimation included); only what consumes their codes differs."""
        if (linear.bias is not None or linear.weight_quantizer.is_stub() or not linear.output_quantizer.is_stub() or linear.input_quantizer.is_stub()
                or gate.dtype != torch.bfloat16):
            return None
        xq, wq = pre if pre is not None else (linear.input_quantizer(x), linear.weight_quantizer(linear.weight))
        RECENT.settle(xq)  # (a no-op unless the codes are an undecided sibling's: this route reads them as they are)
        usable = (isinstance(xq, ff.QuantizedTensor) and isinstance(wq, ff.QuantizedTensor) and xq.raw_data.dtype == torch.int8 and wq.raw_data.dtype == torch.int8)
        if usable:
            xp, wp = xq.quantization_context.quantization_params, wq.quantization_context.quantization_params
            usable = xp.scale.numel() == 1 and ff.fused_linear.KERNELS.row_mode(wq) is not None
        product = None
        if usable:
            w_offset = None if wp.offset is None or ff.fused_linear.known_zero_offset(wp.offset) else wp.offset
            if self._want_product_extrema:  # [min, max] of the product rides along: down_proj's estimator step starts from the two numbers
                both = ff.ops.linear_w8a8_gated(xq.raw_data, wq.raw_data, xp.scale, xp.offset, wp.scale, w_offset, gate, want_extrema=True)
                if both is not None:
                    product, self._product_extrema = both[0], (both[0], both[1])
            else:
                product = ff.ops.linear_w8a8_gated(xq.raw_data, wq.raw_data, xp.scale, xp.offset, wp.scale, w_offset, gate)
        if product is None:  # the quantizers have run (an estimator step each): finish on the codes / tensors they returned
            if usable:
                up = ff.ops.linear_w8a8(xq.raw_data, wq.raw_data, xp.scale, xp.offset, wp.scale, w_offset, None, out_dtype=torch.bfloat16)
            else:
                up = ff.nn.functional.linear(xq, wq, None, output_quantizer=linear.output_quantizer)
            product = ff.ops.silu_mul_quantize(gate, up, (), want_product=True)[0]
        return product

    def _qkv(self, normed: torch.Tensor, attn: torch.nn.Module) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """The three projections of the attention block: one launch for weight-only linears the GEMM covers, else one by one."""
        projections = (attn.q_proj, attn.k_proj, attn.v_proj)
        if all(p.input_quantizer.is_stub() for p in projections):
            stored = None
            if self.weight_storage != "requantize":
                kept = tuple(self._stored_weight(p) for p in projections)
                stored = kept if all(k is not None for k in kept) else None
            if stored is not None or self.weight_storage == "requantize":
                outs = _weight_only_qkv(normed, projections, stored)
                if outs is not None:
                    return outs[0], outs[1], outs[2]
        # (a forward hook on an input quantizer would be handed codes the device may have left unwritten: then every quantizer quantizes)
        with sibling_quantizers(undecided=_codes_stay_with_the_linears(*projections)):
            return self._linear(normed, attn.q_proj), self._linear(normed, attn.k_proj), self._linear(normed, attn.v_proj)

    def _gate_up(self, normed: torch.Tensor, mlp: torch.nn.Module) -> torch.Tensor:
        """silu(gate_proj(x)) * up_proj(x): one launch for a weight-only MLP the GEMM covers, else the two linears + SiLU * up."""
        gate_proj, up_proj = mlp.gate_proj, mlp.up_proj
        if gate_proj.input_quantizer.is_stub() and up_proj.input_quantizer.is_stub():
            stored = None
            if self.weight_storage != "requantize":
                both = (self._stored_weight(gate_proj), self._stored_weight(up_proj))
                stored = both if both[0] is not None and both[1] is not None else None
            if stored is not None or self.weight_storage == "requantize":
                product = _weight_only_gate_up(normed, gate_proj, up_proj, stored)
                if product is not None:
                    return product
        with sibling_quantizers(undecided=_codes_stay_with_the_linears(gate_proj, up_proj)):
            if all(l.bias is None and not l.weight_quantizer.is_stub() and l.output_quantizer.is_stub() and not l.input_quantizer.is_stub()
                   for l in (gate_proj, up_proj)):
                # every quantizer first (each is its own estimator step during range estimation), then ONE op for what consumes their
                # codes: gate + up + SiLU * up as one launch where the two input quantizers turn out — on the device — to hold equal
                # parameters, else the two linears with the gated epilogue
                product, pair, pre = _w8a8_gate_up_product(normed, gate_proj, up_proj, want_extrema=self._want_product_extrema)
                if product is not None:
                    if pair is not None:
                        self._product_extrema = (product, pair)
                    return product
                if pre is not None:
                    gate = self._linear(normed, gate_proj, pre=pre[0])
                    return self._linear(normed, up_proj, gate=gate, pre=pre[1])
            gate = self._linear(normed, gate_proj)
            return self._linear(normed, up_proj, gate=gate)

    @torch.no_grad()
    def __call__(self, input_ids: torch.Tensor, logits: bool = False) -> torch.Tensor:
        model, cfg = self.model, self.model.config
        b, s = input_ids.shape
        d = cfg.head_dim
        with ff.strict_quantization(False):
            hidden = model.embed_tokens(input_ids)
            cos, sin = rotary_tables(s, d, cfg.rope_theta, hidden.device, hidden.dtype)
            pending: torch.Tensor | None = None
            for layer in model.layers:
                attn, mlp = layer.self_attn, layer.mlp
                ln1, ln2 = layer.input_layernorm, layer.post_attention_layernorm
                hidden, normed, _ = ff.ops.add_rmsnorm_quantize(hidden, pending, ln1.weight, ln1.variance_epsilon, (), want_norm=True, sum_inplace=pending is not None)
                q, k, v = self._qkv(normed, attn)
                if attention_kernel_covers(cfg, s, q.dtype):
                    ff.ops.rope_(None if FUSE_Q_ROPE else q, k, cos, sin, d)
                    ctx, _ = ff.ops.attention(q, k, v, d, causal=s > 1, q_rope=(cos, sin) if FUSE_Q_ROPE else None)
                else:
                    ff.ops.rope_(q, k, cos, sin, d)
                    ctx = _sdpa(q, k, v, cfg, b, s)
                attn_out = self._linear(ctx, attn.o_proj)
                hidden, normed, _ = ff.ops.add_rmsnorm_quantize(hidden, attn_out, ln2.weight, ln2.variance_epsilon, (), want_norm=True, sum_inplace=True)
                self._product_extrema = None
                # (only an estimator on down_proj's input quantizer has a use for the product's extrema)
                self._want_product_extrema = next(iter(getattr(mlp.down_proj.input_quantizer, "overrides", ())), None) is not None
                product = self._gate_up(normed, mlp)
                with sibling_quantizers():
                    if self._product_extrema is not None and self._product_extrema[0] is product:
                        RECENT.remember_extrema(product, self._product_extrema[1])
                    pending = self._linear(product, mlp.down_proj)
                self._product_extrema = None
            _, normed, _ = ff.ops.add_rmsnorm_quantize(hidden, pending, model.norm.weight, model.norm.variance_epsilon, (), want_sum=False, want_norm=True)
            return model.lm_head(normed) if logits else normed


FusedCalibrationForward = FusedProducersForward  # the name the calibration path was introduced under


def _weight_row_mode(linear: torch.nn.Module) -> str | None:
    """'tensor' / 'row' (one parameter pair per output channel) / None for the tiling of the linear's weight quantizer
    (DispatcherKernels.row_mode on the weight's shape and the quantizer's granularity)."""
    return ff.fused_linear.KERNELS.row_mode(_Shaped(linear.weight, linear.weight_quantizer))


class _Shaped:
    """Just enough of a quantized weight for ``DispatcherKernels.weight_group``: a shape and its quantizer's granularity."""

    def __init__(self, weight: torch.Tensor, quantizer: torch.nn.Module) -> None:
        self.shape = weight.shape
        self._dim = weight.dim()
        self.quantization_context = types.SimpleNamespace(quantization_params=types.SimpleNamespace(granularity=quantizer.granularity))

    def dim(self) -> int:
        return self._dim


def count_quantizers(model: LlamaModel) -> int:
    return sum(1 for _ in ff.nn.named_quantizers(model))


def approx_tokens_per_second(config: LlamaConfig, tokens: int, seconds: float) -> float:
    return tokens / seconds if seconds > 0 else math.inf
