"""The BASELINE.json configurations other than the headline bench line, on ONE MI355X.

    python tools/bench_configs.py [--configs 2,3,4,5] [--out profiles/rNN_configs.json]

  cfg2  Llama-3-8B W8 per-channel weight-only, bf16 activations: forward tokens/s (B=1 and B=8, S=2048),
        whole-model weight quantize GB/s (A1, 3 B/elem)
  cfg3  cfg2 + A8 per-tensor: RunningMinMax calibration over 512 sequences x 2048 (64 steps of 8):
        sequences/s, then the timed W8A8 forward (same as bench.py; repeated here for one table)
  cfg4  W4 PerBlock(128) weights: quantize+pack GB/s (2.5 B/elem), unpack+dequantize GB/s (2.5 B/elem),
        forward tokens/s (hand-written weight-code GEMM; packed-nibble storage; the vendor-GEMM arm for comparison)
  cfg5  Llama-3-70B shapes, cfg3 recipe, this GPU's share of the 512 sequences (64 on an 8-GPU node):
        calibration wall time; the all-reduce payload (what 8 ranks would exchange)

All inputs are synthetic (seeds as SURVEY §8(d)); every forward is the product path (HIP library through
the C ABI). Timings: wall clock around synchronised regions; forwards replayed from a hipGraph.
"""

from __future__ import annotations

import argparse
import json
import pathlib
import sys
import time

import torch

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

import fastforward_amd as ff  # noqa: E402

from fastforward_amd import distributed as ffd  # noqa: E402
from fastforward_amd import llama, ops  # noqa: E402

DEV = torch.device("cuda", 0)
HBM_PEAK_GBS = 8000.0


def timed_forward(fn, steps: int = 10, warmup: int = 3, graph: bool = True) -> float:
    """Seconds per call of `fn` (a forward over a resident batch)."""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    g = None
    if graph:
        g = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.cuda.graph(g, stream=side):
            fn()
        torch.cuda.current_stream().wait_stream(side)
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        g.replay() if g is not None else fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def ids(config, batch, seq, seed):
    gen = torch.Generator(device=DEV).manual_seed(seed)
    return torch.randint(0, config.vocab_size, (batch, seq), device=DEV, generator=gen)


def event_ms(fn, reps: int = 10) -> float:
    """Median ms per call: `reps` calls captured into a hipGraph and replayed (no host launch path in the timing)."""
    from bench import event_time_ms

    return event_time_ms(lambda r: fn(), iters=8, reps=reps)


def cfg2() -> dict:
    config = llama.LlamaConfig.llama3_8b()
    model = llama.build_model(config, DEV, torch.bfloat16, seed=1234 + 1)
    llama.quantize_llama(model, w_bits=8, a_bits=None, quantized_dtype=torch.int8)
    llama.calibrate(model, [ids(config, 1, 256, 1)])  # weight ranges only (no activation quantizers)
    out = {"config": "Llama-3-8B W8 per-channel weight-only, bf16 activations"}
    for b in (1, 8):
        batch = ids(config, b, 2048, 2 + b)

        def fwd():
            with torch.no_grad(), ff.strict_quantization(False):
                return model(batch)

        s = timed_forward(fwd)
        out[f"forward_B{b}_S2048_module_graph"] = {"ms": round(s * 1e3, 2), "tokens_per_s": round(b * 2048 / s, 1)}
        producers = llama.FusedProducersForward(model)
        s = timed_forward(lambda: producers(batch, logits=True))
        out[f"forward_B{b}_S2048"] = {"ms": round(s * 1e3, 2), "tokens_per_s": round(b * 2048 / s, 1),
                                      "forward": "FusedProducersForward (RMSNorm / rotary / SiLU*up / attention as one-pass kernels); decoder linears: weight quantizer every call (A1), then the hand-written bf16 x weight-code GEMM at every token count (q/k/v as one launch: ops.linear_wq_multi; gate+up+SiLU*up as one launch: ops.mlp_gate_up_wq; o_proj / down_proj: ops.linear_wq; below 4096 tokens the codes are converted inside the GEMM and the tiles of a partly filled round are split along K, from 4096 tokens on A2 runs once per call into a bf16 image) — no vendor GEMM on the quantized path; lm_head stays float"}
        with ff.fused_linear.weight_only_kernel(False):  # A/B arm: the reference's own path for such linears (A2 + F.linear = the vendor's bf16 GEMM)
            s = timed_forward(lambda: producers(batch, logits=True))
        out[f"forward_B{b}_S2048_vendor_gemm_arm"] = {"ms": round(s * 1e3, 2), "tokens_per_s": round(b * 2048 / s, 1),
                                                     "forward": "same, decoder linears through the float fallback (fallback.py:86-112: A2 + F.linear on the vendor's GEMM)"}
        stored = llama.FusedProducersForward(model, weight_storage="codes")
        s = timed_forward(lambda: stored(batch, logits=True))
        out[f"forward_B{b}_S2048_stored_codes"] = {"ms": round(s * 1e3, 2), "tokens_per_s": round(b * 2048 / s, 1),
                                                  "forward": "same as the first, int8 weight codes kept across steps (SURVEY 8(f) row 1): no per-step A1"}
    # whole-model weight quantization: 6.98 G elements, 3 B/elem
    linears = [l for _, l in llama.decoder_linears(model)]

    def quantize_all():
        for l in linears:
            l.weight_quantizer(l.weight)

    ms = event_ms(quantize_all, reps=5)
    elems = config.quantized_weight_elems()
    out["weight_quantize_all_linears"] = {"ms": round(ms, 3), "elements": elems, "GB_per_s": round(elems * 3 / ms / 1e6, 1),
                                          "frac_of_hbm_peak": round(elems * 3 / ms / 1e6 / HBM_PEAK_GBS, 4), "launches": len(linears),
                                          "what": "every weight quantizer's own forward: one A1 launch per linear"}
    layers = [[l for l in (layer.self_attn.q_proj, layer.self_attn.k_proj, layer.self_attn.v_proj, layer.self_attn.o_proj, layer.mlp.gate_proj,
                           layer.mlp.up_proj, layer.mlp.down_proj)] for layer in model.layers]

    def quantize_all_batched():
        for group in layers:
            ops.quantize_rows_batch([l.weight for l in group], [l.weight_quantizer.scale for l in group], [None] * len(group), 8)

    ms = event_ms(quantize_all_batched, reps=5)
    out["weight_quantize_all_linears_batched"] = {"ms": round(ms, 3), "elements": elems, "GB_per_s": round(elems * 3 / ms / 1e6, 1),
                                                  "frac_of_hbm_peak": round(elems * 3 / ms / 1e6 / HBM_PEAK_GBS, 4), "launches": len(layers),
                                                  "what": "ops.quantize_rows_batch: the seven weights of a decoder layer per launch (what llama.FusedForward runs each step)"}
    return out


def cfg3(total_sequences: int) -> dict:
    config = llama.LlamaConfig.llama3_8b()
    model = llama.build_model(config, DEV, torch.bfloat16, seed=1234 + 2)
    llama.quantize_llama(model, w_bits=8, a_bits=8, quantized_dtype=torch.int8)
    steps = total_sequences // 8
    gen = torch.Generator(device=DEV).manual_seed(77)
    batches = [torch.randint(0, config.vocab_size, (8, 2048), device=DEV, generator=gen) for _ in range(steps)]
    def reset() -> None:  # a timed run starts from uninitialised ranges, like a first calibration
        for _, quantizer in ff.nn.named_quantizers(model):
            quantizer.reset_parameters()

    # untimed pass over one batch through either forward: code objects load, the allocator grows (once per process; bench.py does the same)
    ffd.calibrate_sharded(model, batches[:1], disable_quantization=False)
    reset()
    ffd.calibrate_sharded(model, batches[:1], disable_quantization=False, fused=True)
    reset()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ffd.calibrate_sharded(model, batches[:8], disable_quantization=False)  # the reference-shaped module graph, 64 sequences
    torch.cuda.synchronize()
    s_mg = time.perf_counter() - t0
    reset()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    payload = ffd.calibrate_sharded(model, batches, disable_quantization=False, fused=True)
    torch.cuda.synchronize()
    s = time.perf_counter() - t0
    out = {"config": f"Llama-3-8B W8A8, RunningMinMax calibration over {steps * 8} sequences x 2048 tokens ({steps} steps of 8), quantize-while-calibrating (reference default)",
           "calibration": {"seconds": round(s, 2), "sequences_per_s": round(steps * 8 / s, 2), "tokens_per_s": round(steps * 8 * 2048 / s, 1),
                           "range_floats_for_allreduce": payload, "forward": "FusedProducersForward (quantizers' own forwards with their estimator overrides; fused producers in between)"},
           "calibration_module_graph_64_sequences": {"seconds": round(s_mg, 2), "sequences_per_s": round(64 / s_mg, 2)}}
    batch = batches[0]
    fused = llama.FusedForward(model)
    s = timed_forward(lambda: fused(batch))
    out["forward_B8_S2048_fused"] = {"ms": round(s * 1e3, 2), "tokens_per_s": round(8 * 2048 / s, 1)}
    cached = llama.FusedForward(model, cache_weight_codes=True)
    s = timed_forward(lambda: cached(batch))
    out["forward_B8_S2048_fused_cached_weight_codes"] = {"ms": round(s * 1e3, 2), "tokens_per_s": round(8 * 2048 / s, 1),
                                                         "note": "int8 weight codes kept across steps (SURVEY §8(f) row 1); not the headline"}

    def module_graph():
        with torch.no_grad(), ff.strict_quantization(False):
            return model(batch)

    s = timed_forward(module_graph)
    out["forward_B8_S2048_module_graph"] = {"ms": round(s * 1e3, 2), "tokens_per_s": round(8 * 2048 / s, 1)}
    return out


def cfg4() -> dict:
    config = llama.LlamaConfig.llama3_8b()
    out = {"config": "Llama-3-8B W4 PerBlock(block 128 along in, per output channel), packed nibbles (GGUF Q4_0 order)"}
    # kernels on the headline shape
    shape = (14336, 4096)
    n = shape[0] * shape[1]
    ws = [(torch.randn(shape, device=DEV) * 0.02).to(torch.bfloat16) for _ in range(6)]
    gran = ff.PerBlock(1, 128, 0)
    tile = gran.tile_size(torch.Size(shape))
    lo, hi = ops.minmax_by_tile(ws[0], tile)
    scale, offset = ops.parameters_for_range(lo, hi, 4, True, True)
    state = {"i": 0}

    def quant_pack():
        state["i"] += 1
        codes = ops.quantize_by_tile(ws[state["i"] % 6], scale, tile, 4, torch.int8, offset)
        return ops.pack_int4(codes, block=128)

    packed = [quant_pack() for _ in range(6)]

    def unpack_dequant():
        state["i"] += 1
        codes = ops.unpack_int4(packed[state["i"] % 6], shape, torch.int8, block=128)
        return ops.dequantize_by_tile(codes, scale, tile, offset, torch.bfloat16)

    def quant_pack_fused():
        state["i"] += 1
        return ops.quantize_pack_int4(ws[state["i"] % 6], scale, tile, offset, block=128)

    def unpack_dequant_fused():
        state["i"] += 1
        return ops.unpack_dequantize_int4(packed[state["i"] % 6], scale, shape, tile, offset, block=128)

    for name, fn, kern in (("quantize+pack fused", quant_pack_fused, "quantize_pack_int4_kernel"), ("unpack+dequantize fused", unpack_dequant_fused, "unpack_dequantize_int4_kernel")):
        ms = event_ms(fn)
        out[f"{name}_[14336,4096]"] = {"ms": round(ms, 4), "GB_per_s_algorithmic_2.5B_per_elem": round(n * 2.5 / ms / 1e6, 1),
                                       "frac_of_hbm_peak": round(n * 2.5 / ms / 1e6 / HBM_PEAK_GBS, 4), "kernels": kern + " (one launch)"}
    ms = event_ms(quant_pack)
    out["quantize_plus_pack_[14336,4096]"] = {"ms": round(ms, 4), "GB_per_s_algorithmic_2.5B_per_elem": round(n * 2.5 / ms / 1e6, 1),
                                              "frac_of_hbm_peak": round(n * 2.5 / ms / 1e6 / HBM_PEAK_GBS, 4), "kernels": "A1 (bf16 -> int8 codes) + A7 pack (two launches; codes make one HBM round trip)"}
    ms = event_ms(unpack_dequant)
    out["unpack_plus_dequantize_[14336,4096]"] = {"ms": round(ms, 4), "GB_per_s_algorithmic_2.5B_per_elem": round(n * 2.5 / ms / 1e6, 1),
                                                  "frac_of_hbm_peak": round(n * 2.5 / ms / 1e6 / HBM_PEAK_GBS, 4), "kernels": "A7 unpack + A2 (two launches)"}
    del ws, packed
    model = llama.build_model(config, DEV, torch.bfloat16, seed=1234 + 3)
    llama.quantize_llama(model, w_bits=4, a_bits=None, quantized_dtype=torch.int8, weight_granularity=gran)
    llama.calibrate(model, [ids(config, 1, 256, 1)])
    batch = ids(config, 8, 2048, 5)

    def fwd():
        with torch.no_grad(), ff.strict_quantization(False):
            return model(batch)

    s = timed_forward(fwd)
    out["forward_B8_S2048_module_graph"] = {"ms": round(s * 1e3, 2), "tokens_per_s": round(8 * 2048 / s, 1)}
    producers = llama.FusedProducersForward(model)
    s = timed_forward(lambda: producers(batch, logits=True))
    out["forward_B8_S2048"] = {"ms": round(s * 1e3, 2), "tokens_per_s": round(8 * 2048 / s, 1),
                               "forward": "FusedProducersForward; decoder linears: weight quantizer every call (A1, int8 container), then ops.linear_wq with group-128 parameters — no vendor GEMM on the quantized path"}
    with ff.fused_linear.weight_only_kernel(False):
        s = timed_forward(lambda: producers(batch, logits=True))
    out["forward_B8_S2048_vendor_gemm_arm"] = {"ms": round(s * 1e3, 2), "tokens_per_s": round(8 * 2048 / s, 1),
                                               "forward": "same, decoder linears through the float fallback (A2 + F.linear on the vendor's GEMM)"}
    packed = llama.FusedProducersForward(model, weight_storage="packed")
    s = timed_forward(lambda: packed(batch, logits=True))
    stored_bytes = sum(t[1].numel() * t[1].element_size() for t in packed._stored.values())
    out["forward_B8_S2048_packed_nibbles"] = {"ms": round(s * 1e3, 2), "tokens_per_s": round(8 * 2048 / s, 1), "stored_weight_bytes": stored_bytes,
                                              "bytes_per_weight": round(stored_bytes / config.quantized_weight_elems(), 3),
                                              "forward": "weights STORED as packed nibbles (quantize+pack once, 0.5 B per weight), consumed by ops.linear_wq(pack_block=128) as they are: unpack + A2 into the GEMM's bf16 image (two-pass form) or inside the GEMM"}
    return out


def cfg5(local_sequences: int) -> dict:
    config = llama.LlamaConfig.llama3_70b()
    t0 = time.perf_counter()
    model = llama.build_model(config, DEV, torch.bfloat16, seed=1234 + 4)
    llama.quantize_llama(model, w_bits=8, a_bits=8, quantized_dtype=torch.int8)
    torch.cuda.synchronize()
    build_s = time.perf_counter() - t0
    steps = local_sequences // 8
    gen = torch.Generator(device=DEV).manual_seed(78)
    batches = [torch.randint(0, config.vocab_size, (8, 2048), device=DEV, generator=gen) for _ in range(steps)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    payload = ffd.calibrate_sharded(model, batches, disable_quantization=True, fused=True)
    torch.cuda.synchronize()
    s = time.perf_counter() - t0
    fingerprint = ffd.ranges_fingerprint(model)
    out = {"config": f"Llama-3-70B shapes W8A8, RunningMinMax calibration of ONE rank's share: {steps * 8} sequences x 2048 ({steps} steps of 8), ranges on the un-quantized forward",
           "model_build_seconds": round(build_s, 1),
           "calibration": {"seconds": round(s, 2), "sequences_per_s_per_gpu": round(steps * 8 / s, 3), "tokens_per_s_per_gpu": round(steps * 8 * 2048 / s, 1)},
           "allreduce": {"floats": payload, "bytes": payload * 4, "collective": "1 x all_reduce(MIN) of [mins | -maxes | -inf flag] over RCCL (latency-bound)"},
           "quantizer_parameters": int(fingerprint.numel()), "all_finite": bool(torch.isfinite(fingerprint).all()),
           "hbm_GiB_allocated": round(torch.cuda.max_memory_allocated() / 2**30, 1)}
    batch = batches[0]
    fused = llama.FusedForward(model)
    s = timed_forward(lambda: fused(batch), steps=3, warmup=1)
    out["forward_B8_S2048_fused"] = {"ms": round(s * 1e3, 1), "tokens_per_s": round(8 * 2048 / s, 1)}
    return out


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="2,3,4,5")
    ap.add_argument("--calib-seqs", type=int, default=512, help="cfg3: calibration sequences (BASELINE: 512)")
    ap.add_argument("--calib-seqs-70b", type=int, default=64, help="cfg5: this GPU's share (512 / 8)")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    assert torch.cuda.is_available(), "needs the MI355X"
    torch.cuda.set_device(0)
    results = {"device": torch.cuda.get_device_name(0), "hbm_peak_GBs_used": HBM_PEAK_GBS}
    runners = {"2": cfg2, "3": lambda: cfg3(args.calib_seqs), "4": cfg4, "5": lambda: cfg5(args.calib_seqs_70b)}
    for key in args.configs.split(","):
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats()
        t0 = time.perf_counter()
        try:
            results[f"cfg{key}"] = runners[key]()
        except Exception as e:  # keep the other configurations' numbers
            results[f"cfg{key}"] = {"error": f"{type(e).__name__}: {e}"}
        results[f"cfg{key}"]["wall_seconds_total"] = round(time.perf_counter() - t0, 1)
        print(json.dumps({f"cfg{key}": results[f"cfg{key}"]}), flush=True)
    if args.out:
        pathlib.Path(args.out).parent.mkdir(parents=True, exist_ok=True)
        pathlib.Path(args.out).write_text(json.dumps(results, indent=1))


if __name__ == "__main__":
    main()
