"""bench.py — Llama-3-8B W8A8 quantized forward on N MI355X GPUs (one process per GPU).

    python bench.py --gpus 1 --steps 10 --warmup 3
    python bench.py --gpus N ...          # no launcher: starts N rank processes itself (torch.distributed.run children)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): tokens/sec of the Llama-3-8B W8A8 quantized forward, plus quant/dequant
GB/s against the HBM3E peak. Workload = BASELINE.json configs[2]'s forward: Llama-3-8B shapes,
synthetic N(0, 0.02^2) bf16 weights, 8-bit per-channel symmetric weight quantizers and 8-bit
per-tensor asymmetric input quantizers on the 7 linears of each of the 32 layers, ranges from a
RunningMinMax calibration (sharded over the ranks, ONE RCCL all-reduce of the activation ranges).

A "step" is one forward pass over one batch of B x S synthetic token ids resident in HBM. With the
reference's semantics every step re-quantizes all 6.98 G weight elements (nn/linear.py:34),
quantizes the 7 linear inputs per layer and runs the 224 quantized linears. The forward is captured
once into a hipGraph and replayed (no tracing compiler; the C ABI only enqueues kernels), so the
timed region holds no Python.

Multi-GPU: inference shards over the batch with no data-path collective (replicas; "weak"
scaling: per-GPU batch fixed). value = total tokens of all ranks / max-over-ranks time.

Output: ONE JSON line on rank 0. `roofline` describes the dominant kernel of the step (the int8
MFMA GEMM); `hbm_kernels` lists the HBM-bound hot-path kernels on the headline shape
[14336, 4096] bf16 per-channel with achieved GB/s vs 8 TB/s; `cpu_baseline` is the reference's
eager ATen chain (oracle/eager_chain.py, a port) timed on this box's host cores on a bounded sample.
"""

from __future__ import annotations

import argparse
import dataclasses
import json
import os
import pathlib
import statistics
import sys
import time

import torch

ROOT = pathlib.Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

import fastforward_amd as ff  # noqa: E402

from fastforward_amd import distributed as ffd  # noqa: E402
from fastforward_amd import llama, ops  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak (same guide)
INT8_PEAK_TOPS = 5000.0    # dense int8 MFMA peak = 2 x the 2.5 PF dense bf16 peak (same guide)


def event_time_ms(fn, iters: int, reps: int = 8) -> float:
    """Median duration of one call of `fn`: `reps` calls captured into a hipGraph on the stream the
    C ABI launches on, replayed `iters` times between HIP events."""
    for r in range(2):
        fn(r)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side), torch.cuda.graph(graph, stream=side):
        for r in range(reps):
            fn(r)
    torch.cuda.current_stream().wait_stream(side)
    graph.replay()
    torch.cuda.synchronize()
    # HBM-bound kernels measured cold read 10-20 % low on some boxes (memory / fabric clocks ramp with sustained load):
    # keep the graph replaying for ~30 ms before the timed iterations
    t_end = time.perf_counter() + 0.03
    while time.perf_counter() < t_end:
        graph.replay()
        torch.cuda.synchronize()
    samples = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        graph.replay()
        b.record()
        b.synchronize()
        samples.append(a.elapsed_time(b) / reps)
    return statistics.median(samples)


def _pmc_tables(stem: str):
    import glob

    reads = sorted(glob.glob(str(ROOT / "profiles" / f"*{stem}FETCH_SIZE.json")))
    writes = sorted(glob.glob(str(ROOT / "profiles" / f"*{stem}WRITE_SIZE.json")))
    if not reads or not writes:
        return None
    r, w = json.load(open(reads[-1])), json.load(open(writes[-1]))
    stamp = r.pop("__kernel_source_sha16__", None)
    w.pop("__kernel_source_sha16__", None)
    PMC_STAMPS[pathlib.Path(reads[-1]).name] = stamp
    return r, w, pathlib.Path(reads[-1]).name.split("_pmc_")[0]


PMC_STAMPS: dict[str, "str | None"] = {}  # profile file -> fingerprint of the kernel sources it was measured on (None: older profile)


def kernel_source_sha16() -> str:
    """Fingerprint of csrc/*.hip + *.h as tools/pmc_summary.py stamps it into the PMC profiles."""
    import hashlib

    root = ROOT / "fastforward_amd" / "csrc"
    h = hashlib.sha256()
    for f in sorted(list(root.glob("*.hip")) + list(root.glob("*.h"))):
        h.update(f.name.encode() + b"\0" + f.read_bytes())
    return h.hexdigest()[:16]


def pmc_traffic(*needles: str, stems: tuple[str, ...] = ("_pmc_", "_pmc_hbm_")) -> tuple[float | None, str | None]:
    """HBM bytes per launch (read + write) of the kernel whose name contains all `needles`, from the
    committed PMC passes (profiles/*_pmc_FETCH_SIZE.json / *_pmc_WRITE_SIZE.json, produced by
    tools/collect_profiles.sh: separate --pmc runs, FETCH_SIZE x2 on gfx950 as the microarch guide
    prescribes). bench.py cannot run rocprofv3 on itself, so this is a lookup, labelled with its source."""
    for stem in stems:
        tables = _pmc_tables(stem)
        if tables is None:
            continue
        r, w, prof = tables
        for name, row in r.items():
            if all(n in name for n in needles) and name in w:
                return float(row["hbm_read_bytes"] + w[name]["hbm_write_bytes"]), prof
    return None, None


def pmc_traffic_mix(needle: "str | tuple[str, ...]", keep=None) -> tuple[float | None, dict, str | None]:
    """Launch-count-weighted mean of the fabric bytes per launch over the kernel variants whose name contains `needle` (and that
    `keep(name)` accepts: the PMC pass profiles the whole bench command, calibration included, so the caller picks the variants
    of the TIMED forward), the per-variant figures, and the profile the numbers come from."""
    tables = _pmc_tables("_pmc_")
    if tables is None:
        return None, {}, None
    r, w, prof = tables
    total, launches, variants = 0.0, 0, {}
    needles = needle if isinstance(needle, tuple) else (needle,)
    for name, row in r.items():
        if any(n in name for n in needles) and name in w and (keep is None or keep(name)):
            b = float(row["hbm_read_bytes"] + w[name]["hbm_write_bytes"])
            variants[name] = {"launches": row["launches"], "bytes_per_launch": b}
            total += b * row["launches"]
            launches += row["launches"]
    return (total / launches if launches else None), variants, prof


def same_box_probe(device: torch.device, target_ms: float = 50.0) -> dict | None:
    """What THIS box's matrix pipes sustain on toggling operands, measured in this process AFTER the timed region (side measurement):
    v_mfma_i32_16x16x64_i8 (and the bf16 twin) issued back to back on pseudo-random operands, 8 waves per CU, no LDS / global traffic
    (tools/probes/ffq_probe.hip -> tools/probes/libffq_probe.so, built by __graft_entry__.build(); measurement code, not the product
    library). The boards of the pool settle at different clocks under the same 1400 W limit: `frac_of_same_box_probe` is the figure
    that can be compared across boxes, `frac` stays against the nominal 5 POP/s."""
    import ctypes

    so = ROOT / "tools" / "probes" / "libffq_probe.so"
    if not so.exists():
        return None
    lib = ctypes.CDLL(str(so))
    out = {}
    sink = torch.zeros(4, dtype=torch.int32, device=device)
    stream = torch.cuda.current_stream()
    blocks = 2 * torch.cuda.get_device_properties(device).multi_processor_count  # 8 waves per CU
    for name, fn, iters in (("int8_16x16x64", lib.ffq_probe_mfma_i8, 40000), ("bf16_16x16x32", lib.ffq_probe_mfma_bf16, 40000)):
        fn.restype = ctypes.c_int
        fn.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_double), ctypes.c_void_p]
        ops_per_launch = ctypes.c_double(0.0)
        launch = lambda: fn(iters, blocks, sink.data_ptr(), ctypes.byref(ops_per_launch), stream.cuda_stream)  # noqa: E731
        if launch() != 0:  # warm: clocks settle under load
            return None
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        launch()
        e1.record(stream)
        torch.cuda.synchronize()
        one = e0.elapsed_time(e1)
        reps = max(1, min(16, int(round(target_ms / max(one, 1e-3)))))
        e0.record(stream)
        for _ in range(reps):
            launch()
        e1.record(stream)
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        out[name] = {"T(FL)OP/s": round(ops_per_launch.value * reps / ms / 1e9, 1), "ms": round(ms, 2), "launches": reps}
    return out


def gemm_roofline(config: llama.LlamaConfig, tokens: int, device: torch.device, fused: "llama.FusedForward | None", batch: torch.Tensor) -> dict:
    """The dominant kernel of the step: the int8 MFMA GEMM. Algorithmic ops per launch = 2*T*N*K.

    `achieved` comes from the launches of a REAL forward (the code distributions the timed steps see): one eager
    forward with a HIP event pair around each int8 GEMM launch (128 per forward since round 6: q+k+v as one launch, o, down and the fused
    gate+up launch, which contracts both weight matrices and carries the SiLU*up + quantize epilogue), on the stream
    they are launched on.
    The event pair also covers the one-pass side kernel with the weight row sums (1 B/elem of the weight).
    `per_shape_uniform_random` repeats the measurement on uniform random int8 operands (hipGraph-replayed), the
    worst case for the chip's power management: the matrix pipe clocks lower on high-entropy data."""
    h, i, kv = config.hidden_size, config.intermediate_size, config.num_kv_heads * config.head_dim
    shapes = {"q/o_proj": (h, h, 2), "k/v_proj": (kv, h, 2), "gate/up_proj": (i, h, 2), "down_proj": (h, i, 1)}
    per_shape_random = {}
    for name, (n, k, count) in shapes.items():
        xq = torch.randint(-128, 128, (tokens, k), device=device, dtype=torch.int8)
        wq = torch.randint(-128, 128, (n, k), device=device, dtype=torch.int8)
        sx, ox = torch.tensor([0.02], device=device), torch.tensor([4.0], device=device)
        sw = torch.rand(n, device=device) * 0.001 + 0.0005
        ms = event_time_ms(lambda r: ops.linear_w8a8(xq, wq, sx, ox, sw, None, out_dtype=torch.bfloat16), iters=5, reps=4)
        per_shape_random[name] = {"N": n, "K": k, "ms": round(ms, 4), "TOP/s": round(2.0 * tokens * n * k / ms / 1e9, 1)}
        del xq, wq
    # reference point, not a product path: the vendor library's own int8 GEMM (torch._int_mm -> hipBLASLt, int32 output, no
    # scale / zero-point epilogue) on the same uniform random codes — what hand-tuned assembly reaches on this chip
    vendor = {}
    for name, (n, k, count) in shapes.items():
        try:
            xq = torch.randint(-128, 128, (tokens, k), device=device, dtype=torch.int8)
            wt = torch.randint(-128, 128, (n, k), device=device, dtype=torch.int8).t()
            ms = event_time_ms(lambda r: torch._int_mm(xq, wt), iters=5, reps=4)
            vendor[name] = {"ms": round(ms, 4), "TOP/s": round(2.0 * tokens * n * k / ms / 1e9, 1)}
            del xq, wt
        except Exception as e:  # noqa: BLE001  (not every build exposes an int8 GEMM)
            vendor[name] = {"unavailable": f"{type(e).__name__}: {str(e)[:80]}"}
    per_shape, total_ops, total_ms, launches, source = {}, 0.0, 0.0, 0, "uniform random operands (module-graph forward: no per-launch events)"
    if fused is not None:
        fused(batch)  # warm
        torch.cuda.synchronize()
        samples: dict[tuple[int, int], list[float]] = {}
        for _ in range(2):
            fused.linear_events = []
            fused(batch)
            torch.cuda.synchronize()
            for n, k, a, b in fused.linear_events:
                samples.setdefault((n, k), []).append(a.elapsed_time(b))
            fused.linear_events = None
        names = {(h, h): "q/o_proj", (kv, h): "k/v_proj", (i, h): "gate/up_proj", (h, i): "down_proj",
                 (2 * i, h): "gate_proj + up_proj + SiLU*up + quantize (one launch, both weight matrices)",
                 (h + 2 * kv, h): "q_proj + k_proj + v_proj (one launch, three weight matrices)"}
        for (n, k), times in samples.items():
            ms = statistics.mean(times)
            per_shape[names.get((n, k), f"{n}x{k}")] = {"N": n, "K": k, "ms": round(ms, 4), "TOP/s": round(2.0 * tokens * n * k / ms / 1e9, 1), "launches_per_forward": len(times) // 2}
            total_ops += len(times) * 2.0 * tokens * n * k
            total_ms += sum(times)
            launches += len(times)
        source = "HIP events around every quantized linear of two eager forwards of the benchmarked model (real codes)"
    else:
        for name, (n, k, count) in shapes.items():
            total_ops += count * 2.0 * tokens * n * k
            total_ms += count * per_shape_random[name]["ms"]
            launches += count
        per_shape = per_shape_random
    achieved = total_ops / total_ms / 1e9
    # the timed forward launches two instantiations: <bf16, REQUANT=false, MLP=false, WOFF=false> and the gate+up launch
    # <int8, true, true, false>; WOFF=true (weight offsets decided on the device) runs during CALIBRATION only — reported apart
    woff = lambda name: name.rstrip().rstrip(")").rstrip().endswith("true>") or ", true>" in name.split("(")[0][-8:]  # noqa: E731
    traffic, traffic_variants, prof = pmc_traffic_mix("w8a8_gemm256fq_kernel", keep=lambda name: not woff(name))
    traffic_calibration, calibration_variants, _ = pmc_traffic_mix("w8a8_gemm256fq_kernel", keep=woff)
    # algorithmic bytes of the same launch mix: int8 activation codes + int8 weight codes read once, output written once
    # (bf16 for the plain launches; int8 codes for the gate+up launch, which reads two weight matrices)
    # (shape, launches per layer) as a LIST: with kv == h (an MHA config such as --model tiny) dict keys would collide
    plain_launches = [((h, h), 2), ((kv, h), 2), ((h, i), 1)]
    if (h + 2 * kv, h) in {(v["N"], v["K"]) for v in per_shape.values()} and kv != h:  # q / k / v ran as one launch (llama.FusedForward.qkv_one_launch)
        plain_launches = [((h + 2 * kv, h), 1), ((h, h), 1), ((h, i), 1)]
    alg_bytes = sum(c * (tokens * k + n * k + tokens * n * 2) for (n, k), c in plain_launches) + (tokens * h + 2 * i * h + tokens * i)
    alg_launches = sum(c for _, c in plain_launches) + 1
    # what back-to-back MFMAs alone sustain on toggling operands (no memory traffic): tools/probes/mfma_power.hip, committed run
    ceiling = None
    probe = ROOT / "profiles" / "r01_mfma_power_probe.txt"
    if probe.exists():
        rates = [float(line.split("toggling")[1].split("TOP/s")[0]) for line in probe.read_text().splitlines() if line.startswith("16x16x64 random bytes")]
        ceiling = round(statistics.mean(rates), 1) if rates else None
    probe_now = same_box_probe(device)
    probe_tops = None if not probe_now else probe_now["int8_16x16x64"]["T(FL)OP/s"]
    return {
        "bound": "mfma",
        "same_box_probe_TOPs": probe_tops,
        "frac_of_same_box_probe": None if not probe_tops else round(achieved / probe_tops, 4),
        "same_box_probe": probe_now,
        "same_box_probe_note": "v_mfma_i32_16x16x64_i8 (and its bf16 twin) back to back on toggling pseudo-random operands, 8 waves per CU, no LDS / global traffic, ~50 ms "
                               "in this process after the timed region (tools/probes/ffq_probe.hip): what this board's matrix pipe sustains under its power limit; "
                               "frac_of_same_box_probe = achieved / that, comparable across the pool's boxes",
        "kernel": "w8a8_gemm256fq_kernel (v_mfma_i32_16x16x64_i8, 256x256 tiles, persistent one-block-per-CU tile loop, ping-pong wave groups, full-line LDS-DMA staging; plain and gate+up/SiLU-epilogue modes) + rowsum_i8_kernel",
        "achieved": round(achieved, 1),
        "peak": INT8_PEAK_TOPS,
        "unit": "TFLOP/s",
        "unit_note": "integer multiply-accumulates (TOP/s); dense int8 MFMA peak",
        "frac": round(achieved / INT8_PEAK_TOPS, 4),
        "mfma_only_ceiling_on_toggling_operands": ceiling,
        "mfma_only_ceiling_note": None if ceiling is None else "TOP/s of v_mfma_i32_16x16x64_i8 issued back to back on random operands, no LDS / global traffic "
                                  "(profiles/r01_mfma_power_probe.txt): the chip is power-limited on real data; frac stays against the nominal peak",
        "traffic": traffic,
        "traffic_lookup": {"kernel_source_sha16_now": kernel_source_sha16(), "profiles": dict(PMC_STAMPS),
                           "note": "traffic figures are lookups of committed rocprofv3 --pmc passes (a process cannot profile itself); a profile whose stamp differs from kernel_source_sha16_now was measured on older kernel sources"},
        "traffic_note": None if traffic is None else f"fabric (L2-miss) read+write bytes per launch from FETCH_SIZE x2 + WRITE_SIZE, launch-count-weighted mean over the two kernel variants the TIMED forward launches "
                        f"(plain bf16-out and the gate+up / SiLU / quantize launch; the calibration-only WOFF variant is listed apart); separate --pmc passes of profiles/{prof}_pmc_*.json. Infinity-Cache hits are counted, so this is L2->fabric traffic, an upper bound of HBM bytes",
        "traffic_per_variant": traffic_variants,
        "traffic_calibration_variant": {"bytes_per_launch": traffic_calibration, "variants": calibration_variants,
                                        "note": "WOFF=true instantiation (weight offsets decided on the device): calibration steps only, NOT part of `traffic`"},
        "algorithmic_bytes_per_launch": alg_bytes / alg_launches,
        "algorithmic_bytes_note": f"codes of x and W read once + output written once, mean over the same launch mix ({alg_launches - 1} plain launches + 1 gate+up launch per layer)",
        "avg_launch_ms": round(total_ms / launches, 4),
        "algorithmic_ops_per_launch": total_ops / launches,
        "measured_on": source,
        "per_shape": per_shape,
        "per_shape_uniform_random": per_shape_random,
        "vendor_int8_gemm_same_shapes_uniform_random": vendor,
        "vendor_note": "torch._int_mm (hipBLASLt's tuned assembly, int32 output, no epilogue) on the same shapes and fill: a reference point for what the chip sustains, never on the product path",
    }


def host_us_per_op(device: torch.device, calls: int = 1000) -> dict:
    """Host cost of every operator of the seam on both routes: `calls` eager launches on tiny tensors (the kernels themselves are
    ~2 us), wall time per call with the device drained before and after. "cpp" = torch.ops.fastforward_amd.* with the C++ dispatch-key
    kernels of libffq_torch.so (dispatcher -> C++ -> C ABI; what ops.* and the quantizer modules take), "python" = the Python body
    of the same operator (Python -> ctypes -> C ABI; what runs when the extension is absent)."""
    o = torch.ops.fastforward_amd
    x = torch.randn(256, device=device, dtype=torch.bfloat16)
    x2 = torch.randn(16, 256, device=device, dtype=torch.bfloat16)
    scale, scale16 = torch.tensor([0.05], device=device), torch.rand(16, device=device) * 0.05 + 0.01
    lo, hi = torch.full((1,), float("inf"), dtype=torch.bfloat16, device=device), torch.full((1,), float("-inf"), dtype=torch.bfloat16, device=device)
    s_out, o_out, flags = torch.empty(1, device=device), torch.empty(1, device=device), torch.zeros(1, dtype=torch.int32, device=device)
    xq = torch.randint(-128, 128, (16, 256), device=device, dtype=torch.int8)
    wq = torch.randint(-128, 128, (128, 256), device=device, dtype=torch.int8)
    sw = torch.rand(128, device=device) * 1e-2 + 1e-3
    ox = torch.tensor([3.0], device=device)
    bf, i8 = torch.bfloat16, torch.int8
    routes = {
        "quantize_by_tile": (lambda: o.quantize_by_tile(x, scale, [256], 8.0, i8, None), lambda: ops.quantize_by_tile(x, scale, (256,), 8, i8)),
        "dequantize_by_tile": (lambda: o.dequantize_by_tile(xq, scale16, [1, 256], None, bf), lambda: ops.dequantize_by_tile(xq, scale16, (1, 256), None, bf)),
        "quantize_dynamic_by_tile": (lambda: o.quantize_dynamic_by_tile(x2, [1, 256], 8.0, False, True, i8), lambda: ops.quantize_dynamic_by_tile(x2, (1, 256), 8, False, True, i8)),
        "quantize_by_tile_backward": (lambda: o.quantize_by_tile_backward(x2, x2, scale16, [1, 256], 8.0, None), lambda: ops.quantize_by_tile_backward(x2, x2, scale16, (1, 256), 8.0, None)),
        "running_minmax_step": (lambda: o.running_minmax_step(x, [256], lo, hi, flags, 8.0, False, True, s_out, o_out),
                                lambda: ops._running_minmax_step(x, (256,), lo, hi, flags, 8.0, False, True, s_out, o_out)),
        "linear_w8a8": (lambda: o.linear_w8a8(xq, wq, scale, ox, sw, None, None, bf, None, None, 8.0, None, None),
                        lambda: ops._linear_w8a8(xq, wq, scale, ox, sw, None, None, bf, None, None, 8.0, None, None)),
        "bmm_w8a8": (lambda: o.bmm_w8a8(xq.view(1, 16, 256), wq.view(1, 128, 256), scale, ox, scale, None, bf, None, None, 8.0, None),
                     lambda: ops._bmm_w8a8(xq.view(1, 16, 256), wq.view(1, 128, 256), scale, ox, scale, None, bf, None, None, 8.0, None)),
        "linear_wq": (lambda: o.linear_wq(x2, wq, sw, None, 256, None, bf, 0, -1, 0), lambda: ops._linear_wq(x2, wq, sw, None, 256, None, bf, 0, -1, 0)),
    }
    out: dict = {"native_dispatch": bool(ops.NATIVE_DISPATCH), "calls": calls, "unit": "us per call, host side"}
    for name, pair in routes.items():
        row = {}
        for label, fn in zip(("cpp" if ops.NATIVE_DISPATCH else "python via the registry", "python"), pair):
            try:
                for _ in range(50):
                    fn()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(calls):
                    fn()
                torch.cuda.synchronize()
                row[label] = round((time.perf_counter() - t0) / calls * 1e6, 2)
            except Exception as e:  # noqa: BLE001  (a schema detail must not cost the bench line)
                row[label] = f"unavailable: {type(e).__name__}: {str(e)[:80]}"
        out[name] = row
    return out


def hbm_kernels(device: torch.device) -> list[dict]:
    """A1 / A2 / A4 on the headline shape of SURVEY §8(d): [14336, 4096] bf16, per-channel."""
    shape = (14336, 4096)
    n = shape[0] * shape[1]
    ws = [(torch.randn(shape, device=device) * 0.02).to(torch.bfloat16) for _ in range(6)]  # > Infinity Cache
    scale = torch.rand(shape[0], device=device) * 0.001 + 0.0005
    tile = (1, shape[1])
    qs = [ops.quantize_by_tile(w, scale, tile, 8, torch.int8) for w in ws]
    rows = []

    def add(name, kernel, needles, bytes_per_elem, fn, n=n):
        ms = event_time_ms(fn, iters=10, reps=12)
        gbs = n * bytes_per_elem / ms / 1e6
        traffic, _ = pmc_traffic(*needles, stems=("_pmc_hbm_",))  # the launches of tools/hbm_probe.py: this shape
        rows.append({"op": name, "kernel": kernel, "bound": "hbm", "bytes_per_elem": bytes_per_elem, "algorithmic_bytes": n * bytes_per_elem,
                     "ms": round(ms, 5), "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": traffic})

    add("quantize per-channel bf16->int8", "quantize_stream_kernel<bf16,i8,ROWS>", ("quantize_stream_kernel<ffq::bf16_t, signed char, 1,", "false"), 3,
        lambda r: ops.quantize_by_tile(ws[r % 6], scale, tile, 8, torch.int8))
    add("quantize per-channel bf16->bf16 (reference default container)", "quantize_stream_kernel<bf16,bf16,ROWS>", ("quantize_stream_kernel<ffq::bf16_t, ffq::bf16_t, 1,", "false"), 4,
        lambda r: ops.quantize_by_tile(ws[r % 6], scale, tile, 8, torch.bfloat16))
    add("dequantize per-channel int8->bf16", "dequantize_stream_kernel<i8,bf16,ROWS>", ("dequantize_stream_kernel<signed char, ffq::bf16_t, 1,", "false"), 3,
        lambda r: ops.dequantize_by_tile(qs[r % 6], scale, tile, None, torch.bfloat16))
    add("running min/max per-channel bf16", "minmax_rows_kernel<bf16>", ("minmax_rows_kernel<ffq::bf16_t",), 2, lambda r: ops.minmax_by_tile(ws[r % 6], tile))
    # A3, per-token dynamic quantization of an activation [8, 2048, 4096] (asymmetric: min, max, A5, A1 in ONE launch, 2 R + 1 W)
    acts = [torch.randn(8, 2048, 4096, device=device, dtype=torch.bfloat16) for _ in range(5)]  # 5 x 134 MB > Infinity Cache
    add("dynamic quantize per-token bf16->int8 [8,2048,4096] (A4 + A5 + A1, one launch)", "quantize_dynamic_rows_kernel<bf16,i8,16,64,4> (one wave per row)", ("quantize_dynamic_rows_kernel<ffq::bf16_t, signed char",), 3,
        lambda r: ops.quantize_dynamic_by_tile(acts[r % 5], (1, 1, 4096), 8, False, True, torch.int8), n=8 * 2048 * 4096)
    del acts
    # producer-fused A1 on the same number of elements ([14336, 4096] read as 14336 rows of 4096)
    s1, o1 = torch.tensor([0.03], device=device), torch.tensor([3.0], device=device)
    gamma = torch.ones(shape[1], device=device, dtype=torch.bfloat16)
    add("residual add + RMSNorm + quantize (bf16, bf16 -> bf16 sum, int8 codes)", "add_rmsnorm_quantize_kernel<1,4>", ("add_rmsnorm_quantize_kernel<1, 4>",), 7,
        lambda r: ops.add_rmsnorm_quantize(ws[r % 6], ws[(r + 1) % 6], gamma, 1e-5, [(s1, o1)]))
    add("SiLU(gate) * up + quantize (bf16, bf16 -> int8 codes)", "silu_mul_quantize_table_kernel", ("silu_mul_quantize_table_kernel",), 5,
        lambda r: ops.silu_mul_quantize(ws[r % 6], ws[(r + 1) % 6], [(s1, o1)]))
    # W4 group-128: A1 + A7 and A7 + A2 in one pass each (2.5 B/elem), and the backward of fake quantization (6 B/elem)
    g4 = torch.rand(n // 128, device=device) * 0.002 + 0.002
    packed = [ops.quantize_pack_int4(w, g4, (1, 128), None, block=128) for w in ws]
    add("W4 group-128 quantize + pack (bf16 -> nibbles)", "quantize_pack_int4_kernel", ("quantize_pack_int4_kernel",), 2.5,
        lambda r: ops.quantize_pack_int4(ws[r % 6], g4, (1, 128), None, block=128))
    add("W4 group-128 unpack + dequantize (nibbles -> bf16)", "unpack_dequantize_int4_kernel", ("unpack_dequantize_int4_kernel",), 2.5,
        lambda r: ops.unpack_dequantize_int4(packed[r % 6], g4, shape, (1, 128), None, block=128))
    add("quantize_by_tile_backward per-channel (bf16 data + grad -> bf16 dinput, fp32 dscale)", "quantize_backward_kernel", ("quantize_backward_kernel",), 6,
        lambda r: ops.quantize_by_tile_backward(ws[r % 6], ws[(r + 1) % 6], scale, tile, 8.0, None))
    return rows


def attention_kernel(config: llama.LlamaConfig, batch: int, seq_len: int, device: torch.device) -> dict | None:
    """The second MFMA-bound launch of the step: causal attention + the o_proj input quantizer (csrc/ffq_attention.hip),
    timed on the forward's shape with N(0,1) operands; flops = 4 * B * H * S^2 * D / 2 (the causal half), peak = dense bf16."""
    if not llama.attention_kernel_covers(config, seq_len, torch.bfloat16):
        return None
    h, hk, d = config.num_heads, config.num_kv_heads, config.head_dim
    qs = [torch.randn(batch, seq_len, h * d, device=device, dtype=torch.bfloat16) for _ in range(2)]
    ks = [torch.randn(batch, seq_len, hk * d, device=device, dtype=torch.bfloat16) for _ in range(2)]
    vs = [torch.randn(batch, seq_len, hk * d, device=device, dtype=torch.bfloat16) for _ in range(2)]
    sc, of = torch.tensor([0.03], device=device), torch.tensor([-3.0], device=device)
    ms = event_time_ms(lambda r: ops.attention(qs[r % 2], ks[r % 2], vs[r % 2], d, causal=True, quantizer=(sc, of), want_context=False), iters=10, reps=8)
    flops = 4.0 * batch * h * seq_len * seq_len * d / 2
    import torch.nn.functional as F
    ms_sdpa = event_time_ms(lambda r: F.scaled_dot_product_attention(
        qs[r % 2].view(batch, seq_len, h, d).transpose(1, 2), ks[r % 2].view(batch, seq_len, hk, d).transpose(1, 2),
        vs[r % 2].view(batch, seq_len, hk, d).transpose(1, 2), is_causal=True, enable_gqa=h != hk), iters=5, reps=4)
    return {"op": "causal GQA attention + o_proj input quantizer (bf16 q/k/v -> int8 codes), one launch per layer", "kernel": "attention_fwd_kernel<true>",
            "bound": "mfma", "shape": {"batch": batch, "seq_len": seq_len, "q_heads": h, "kv_heads": hk, "head_dim": d},
            "algorithmic_flops": flops, "ms": round(ms, 4), "achieved": round(flops / ms / 1e9, 1), "peak": BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(flops / ms / 1e9 / BF16_PEAK_TFLOPS, 4),
            "torch_sdpa_same_shape": {"ms": round(ms_sdpa, 4), "TFLOP/s": round(flops / ms_sdpa / 1e9, 1), "note": "attention only (AOTriton flash); the o_proj quantizer is a separate pass there"}}


def cpu_baseline(config: llama.LlamaConfig, budget_s: float = 20.0) -> dict:
    """The reference's eager chain on this box's host cores: the 7 W8A8 linears of ONE decoder layer
    (quantize x, re-quantize W, dequantize both, bf16 F.linear) on a token sample sized to ~budget_s."""
    sys.path.insert(0, str(ROOT / "oracle"))
    import eager_chain

    torch.manual_seed(1236)
    h, i, kv = config.hidden_size, config.intermediate_size, config.num_kv_heads * config.head_dim
    shapes = [(h, h), (kv, h), (kv, h), (h, h), (i, h), (i, h), (h, i)]
    weights = [(torch.randn(n, k) * 0.02).to(torch.bfloat16) for n, k in shapes]
    params = []
    for w in weights:
        lo, hi = eager_chain.minmax(w, (1, w.shape[1]))
        params.append(eager_chain.parameters_for_range(lo, hi, 8, True, True)[0])

    def layer(tokens: int) -> float:
        t0 = time.perf_counter()
        for w, ws in zip(weights, params):
            x = torch.randn(tokens, w.shape[1]).to(torch.bfloat16)
            eager_chain.linear_w8a8(x, w, torch.tensor([0.03]), torch.tensor([3.0]), ws, torch.zeros_like(ws), 8)
        return time.perf_counter() - t0

    layer(8)  # page in
    probe_tokens = 64
    probe = layer(probe_tokens)
    # time(tokens) ~ fixed (weight quant/dequant) + slope * tokens; estimate with a second probe
    probe2 = layer(2 * probe_tokens)
    slope = max((probe2 - probe) / probe_tokens, 1e-9)
    fixed = max(probe - slope * probe_tokens, 0.0)
    tokens = int(min(8192, max(128, (budget_s - fixed) / slope)))
    tokens = 1 << (tokens.bit_length() - 1)
    seconds = layer(tokens)
    while seconds < 10.0 and tokens < 8192:  # the probes over-estimate the slope: grow the sample to >= 10 s of CPU work
        tokens *= 2
        seconds = layer(tokens)
    # the per-op CPU numbers behind BASELINE.md §3's table, on the headline weight shape (all shapes: profiles/r03_micro.md)
    w = weights[4]  # gate_proj [intermediate, hidden]
    tile = (1, w.shape[1])
    codes = eager_chain.quantize(w, params[4], tile, 8, torch.int8)
    per_op = {}
    for name, fn in (("A1 quantize bf16->int8", lambda: eager_chain.quantize(w, params[4], tile, 8, torch.int8)),
                     ("A2 dequantize int8->bf16", lambda: eager_chain.dequantize(codes, params[4], tile, None, torch.bfloat16)),
                     ("A4 min/max", lambda: eager_chain.minmax(w, tile))):
        fn()
        t0 = time.perf_counter()
        fn()
        per_op[name] = round((time.perf_counter() - t0) * 1e3, 2)
    return {
        "value": round(tokens / (seconds * config.num_layers), 3),
        "unit": "tokens/s",
        "cores": torch.get_num_threads(),
        "kind": "port",
        "per_op_ms": {"shape": list(w.shape), "granularity": "per output channel", **per_op},
        "sample": f"eager ATen chain (oracle/eager_chain.py) of the 7 W8A8 linears of ONE decoder layer at Llama-3-8B shapes on {tokens} tokens: {seconds:.2f} s, scaled by {config.num_layers} layers (attention/norms excluded, so this flatters the CPU)",
        "host_cpus": os.cpu_count(),
    }


def relaunch_under_torchrun(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes (one per GPU) as CHILDREN through
    torch.distributed.run and hand back their exit code. The ranks are NEW child processes — never an exec of this process —
    so whatever this process has or has not done with the GPU is irrelevant to them."""
    import socket
    import subprocess

    have = torch.cuda.device_count()
    if have < n and os.environ.get("FFQ_DIST_BACKEND") != "gloo":  # gloo: the documented several-ranks-on-one-GPU dry run of the control flow
        print(f"bench.py: --gpus {n} but this machine exposes {have} GPU(s); refusing to report a {n}-GPU number from fewer devices", file=sys.stderr)
        return 2
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           str(pathlib.Path(__file__).resolve())] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this driver
    return subprocess.run(cmd, env=env).returncode


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8, help="sequences per GPU per step")
    ap.add_argument("--seq-len", type=int, default=2048)
    ap.add_argument("--calib-seqs", type=int, default=None, help="calibration sequences per GPU; default: BASELINE's 512 in total, i.e. 512 / N per GPU (512 on one GPU, 64 each on 8)")
    ap.add_argument("--model", choices=["llama3-8b", "llama3-70b", "tiny"], default="llama3-8b")
    ap.add_argument("--layers", type=int, default=None, help="override the number of decoder layers (smoke runs of the 70B shapes)")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a hipGraph")
    ap.add_argument("--module-graph", action="store_true", help="run the reference-shaped module graph (one quantizer call per linear input, "
                    "eager RMSNorm / rotary / SiLU) instead of llama.FusedForward (A1 fused into those producers)")
    ap.add_argument("--cache-weight-codes", action="store_true", help="keep int8 weight codes across steps (NOT the headline: the reference re-quantizes)")
    ap.add_argument("--no-side-measurements", action="store_true", help="skip roofline / cpu_baseline legs")
    ap.add_argument("--qkv-three-launches", action="store_true", help="A/B arm: q_proj / k_proj / v_proj as three launches of the int8 GEMM (round 5) instead of one")
    ap.add_argument("--batch-rowsums", action="store_true", help="A/B: weight row sums from the batched weight-quantization launch instead of one rowsum_i8 launch per linear (measured slower: llama.FusedForward)")
    ap.add_argument("--layer-batched-weights", action="store_true", help="A/B: all seven weights of a layer re-quantized by one launch at the top of the layer (round 4's schedule) instead of group by group right before their GEMMs")
    ap.add_argument("--fuse-rowsums", action="store_true", help="A/B: every weight re-quantized right before its GEMM by the one-pass codes + row sums kernel (no batched launch, no rowsum_i8 launches)")
    ap.add_argument("--force-dist", action="store_true", help="create the process group even for one rank: the range all-reduce and the cross-rank "
                    "check then run through the collective backend (RCCL with one rank executes the same all_reduce(MIN) an 8-GPU run issues)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and int(os.environ.get("WORLD_SIZE", "1")) == 1 and "RANK" not in os.environ:
        raise SystemExit(relaunch_under_torchrun(args.gpus))  # N fresh ranks; never a silent 1-GPU run
    if args.calib_seqs is None:
        args.calib_seqs = max(args.batch, 512 // args.gpus)

    rank, local_rank, world = ffd.init_process_group_from_env(force=args.force_dist)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU (python bench.py --gpus N does it itself)")
    assert torch.cuda.is_available(), "bench.py needs the MI355X; there is no CPU path"
    local_rank %= torch.cuda.device_count()  # identity with one GPU per rank; lets several ranks share a GPU in a gloo dry run
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    config = {"llama3-8b": llama.LlamaConfig.llama3_8b, "llama3-70b": llama.LlamaConfig.llama3_70b, "tiny": llama.LlamaConfig.tiny}[args.model]()
    if args.layers is not None:
        config = dataclasses.replace(config, num_layers=args.layers)

    # synthetic model + data (seeds per SURVEY §8d: 1234 + config index; ranks hold identical replicas)
    model = llama.build_model(config, device, torch.bfloat16, seed=1234 + 2)
    llama.quantize_llama(model, w_bits=8, a_bits=8, quantized_dtype=torch.int8)
    gen = torch.Generator(device=device).manual_seed(4321 + rank)
    batch = torch.randint(0, config.vocab_size, (args.batch, args.seq_len), device=device, generator=gen)

    # calibration: RunningMinMax on this rank's share, then ONE all-reduce of the activation ranges
    calib_steps = max(1, args.calib_seqs // args.batch)
    calib = [torch.randint(0, config.vocab_size, (args.batch, args.seq_len), device=device, generator=gen) for _ in range(calib_steps)]
    # untimed pass over the first batch: code objects load, the allocator grows (0.3-0.4 s, once per process). Then every
    # quantizer is reset, so the timed calibration starts from uninitialised ranges exactly like a first calibration:
    # the estimators begin at (+inf, -inf) in the DATA dtype and every step takes the in-place A4 kernel (an estimator
    # seeded from a previous fp32 range would merge through torch.min / torch.max instead; ADVICE r1, bench.py:329).
    ffd.calibrate_sharded(model, calib[:1], disable_quantization=False, fused=not args.module_graph)
    for _, quantizer in ff.nn.named_quantizers(model):
        quantizer.reset_parameters()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    payload = ffd.calibrate_sharded(model, calib, disable_quantization=False, fused=not args.module_graph)
    torch.cuda.synchronize()
    calib_s = time.perf_counter() - t0
    del calib
    # self-check of the sharded calibration (N > 1): every rank must hold bit-identical parameters after the one all-reduce
    ranges_identical, ranks_seen = ffd.ranges_agree_across_ranks(model)
    exchange = dict(ffd.last_exchange)

    fused = None if args.module_graph else llama.FusedForward(model, cache_weight_codes=args.cache_weight_codes, batch_rowsums=args.batch_rowsums, fuse_rowsums=args.fuse_rowsums, just_in_time_weights=not args.layer_batched_weights, qkv_one_launch=not args.qkv_three_launches)

    def forward():
        if fused is not None:
            return fused(batch, logits=True)
        with torch.no_grad(), ff.strict_quantization(False):
            return model(batch, logits=True)

    for _ in range(max(args.warmup, 1)):
        forward()
    torch.cuda.synchronize()
    graph = None
    if not args.no_graph:
        graph = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.cuda.graph(graph, stream=side):
            static_out = forward()
        torch.cuda.current_stream().wait_stream(side)
        graph.replay()
        torch.cuda.synchronize()

    def step():
        if graph is not None:
            graph.replay()
        else:
            forward()

    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())

    # the drop-in path beside the headline: what ff.quantize_model() gives with NO harness — the module graph itself, every
    # quantizer its own module call, the dispatcher's int8 linear; its Llama modules run RMSNorm / SiLU*up / rotary / attention as
    # one-pass kernels wherever the slots between them are untouched stubs (llama.py), or, inside llama.eager_modules(), as the
    # reference helpers' eager ATen chains. Same arithmetic, same batch, hipGraph-replayed; rank 0 only, after the headline region
    module_graph = None
    if fused is not None and rank == 0 and not args.no_side_measurements:
        mg_steps = max(2, min(args.steps, 5))

        def replayed(fn) -> float:
            """Seconds per call of `fn`, replayed from a hipGraph."""
            for _ in range(2):
                fn()
            torch.cuda.synchronize()
            g, side = torch.cuda.CUDAGraph(), torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side), torch.cuda.graph(g, stream=side):
                out = fn()
            torch.cuda.current_stream().wait_stream(side)
            g.replay()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(mg_steps):
                g.replay()
            torch.cuda.synchronize()
            seconds = (time.perf_counter() - t1) / mg_steps
            del g, out
            return seconds

        def module_forward():
            with torch.no_grad(), ff.strict_quantization(False):
                return model(batch, logits=True)

        def eager_forward():
            with llama.eager_modules():
                return module_forward()

        tokens = args.batch * args.seq_len
        s_mod = replayed(module_forward)
        module_graph = {"value": round(tokens / s_mod, 1), "unit": "tokens/s", "n_gpus": 1, "steps": mg_steps, "ms_per_step": round(s_mod * 1e3, 3),
                        "what": "model(batch) on the module graph ff.quantize_model() built: one quantizer call per linear input and weight, the dispatcher's int8 linear, the quantized Llama modules' own forwards (one-pass RMSNorm / rotary / attention kernels between untouched stub slots, the MLP's gate + up + SiLU*up + down_proj input quantizer as one launch of the int8 GEMM, o_proj's input quantizer in the attention launch) — no harness"}
        s_eager = replayed(eager_forward)
        module_graph["eager_producer_chains"] = {
            "value": round(tokens / s_eager, 1), "unit": "tokens/s", "ms_per_step": round(s_eager * 1e3, 3),
            "what": "the same module graph inside llama.eager_modules(): RMSNorm / rotary / SiLU / SDPA as the eager ATen chains of the reference's helper modules"}
        # every quantizer still runs its own forward; additionally the residual adds ride in the RMSNorm launches (llama.FusedProducersForward)
        producers = llama.FusedProducersForward(model)
        s_prod = replayed(lambda: producers(batch, logits=True))
        module_graph["producers_fused_quantizers_untouched"] = {
            "value": round(tokens / s_prod, 1), "unit": "tokens/s", "ms_per_step": round(s_prod * 1e3, 3),
            "what": "llama.FusedProducersForward: quantizer modules called as they are, RMSNorm / rotary / SiLU*up / attention as one-pass kernels"}

    tokens_per_step = args.batch * args.seq_len * world
    result = {
        "metric": "tokens/sec Llama-3-8B W8A8 quantized fwd; quant/dequant GB/s vs HBM peak",
        "value": round(tokens_per_step * args.steps / elapsed, 1),
        "unit": "tokens/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "int8",
        "dtype_note": "int8 codes x int8 codes -> int32 on the matrix cores, fp32 scale / zero-point epilogue; everything between the quantized linears in bf16 (fp32 math inside RMSNorm / SiLU / softmax), as the reference recipe",
        "data": "synthetic",
        "config": {
            "workload": f"{args.model} shapes, W8 per-channel symmetric + A8 per-tensor asymmetric on the 7 linears x {config.num_layers} layers "
                        f"(BASELINE.json configs[2] forward), " + ("int8 weight codes cached across steps" if args.cache_weight_codes else "weights re-quantized every step as in the reference"),
            "global_batch": args.batch * world,
            "seq_len": args.seq_len,
            "tokens_per_step": tokens_per_step,
            "parallelism": f"dp{world} (batch-sharded replicas, no data-path collective)",
            "quantizers": llama.count_quantizers(model),
            "launch": "hipGraph replay" if graph is not None else "eager",
            "forward": "module graph (reference-shaped)" if fused is None else "llama.FusedForward (A1 fused into RMSNorm / SiLU*up / attention, rotary in place)",
        },
        "world_size": torch.distributed.get_world_size() if world > 1 else 1,
        "collective_backend": (torch.distributed.get_backend() + (" (RCCL over xGMI)" if torch.distributed.get_backend() == "nccl" else "")) if torch.distributed.is_initialized() else None,
        "module_graph_drop_in": module_graph,
        "calibration": {"sequences_per_gpu": calib_steps * args.batch, "sequences_total": calib_steps * args.batch * world, "seconds": round(calib_s, 3),
                        "sequences_per_s_all_gpus": round(calib_steps * args.batch * world / calib_s, 2),
                        "allreduce_floats": payload, "collective": "1 x all_reduce(MIN) over RCCL" if world > 1 else ("1 x all_reduce(MIN) through a one-rank group (--force-dist)" if torch.distributed.is_initialized() else "none (1 GPU)"),
                        # self-validation of the sharded path (N > 1): ranks that took part, wall time of the one collective, and
                        # whether every rank ended with bit-identical quantizer parameters (two extra all-reduces on the fingerprint)
                        "ranks_seen": ranks_seen, "ranges_identical_across_ranks": ranges_identical,
                        "all_reduce_us": round(float(exchange["seconds"]) * 1e6, 1) if exchange else None,
                        "all_reduce_backend": exchange.get("backend") if exchange else None},
    }
    if rank == 0 and not args.no_side_measurements:
        del graph
        torch.cuda.empty_cache()
        result["roofline"] = gemm_roofline(config, args.batch * args.seq_len, device, fused, batch)
        result["hbm_kernels"] = hbm_kernels(device)
        if fused is not None:
            result["attention_kernel"] = attention_kernel(config, args.batch, args.seq_len, device)
        result["host_us_per_op"] = host_us_per_op(device)
        if world == 1:
            result["cpu_baseline"] = cpu_baseline(config)
        # the figures a reader needs first, repeated compactly at the END of the line: a log viewer that keeps only the last few kB
        # of stdout still shows them (the calibration rate was cut off the record's tail)
        roof = result["roofline"]
        result["summary"] = {
            "tokens_per_s": result["value"], "ms_per_step": result["ms_per_step"],
            "calibration_sequences_per_s": result["calibration"]["sequences_per_s_all_gpus"],
            "module_graph_drop_in_tokens_per_s": None if not module_graph else module_graph.get("value"),
            "roofline_frac": roof["frac"], "same_box_probe_TOPs": roof["same_box_probe_TOPs"], "frac_of_same_box_probe": roof["frac_of_same_box_probe"],
            "hbm_kernels_frac": {k["op"][:48]: k.get("frac") for k in result["hbm_kernels"] if isinstance(k, dict) and "op" in k},
        }
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    if rank == 0:
        # the ONE JSON line goes out last: RCCL prints a version banner through C stdio, which would otherwise be flushed behind it
        import ctypes

        ctypes.CDLL(None).fflush(None)
        print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
