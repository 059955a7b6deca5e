"""Kernel microbenchmarks on the Llama-3-8B shapes of SURVEY §8(d): achieved GB/s per kernel.

Times each hot-path kernel with HIP events on torch's current stream (the stream the C ABI launches
on), median over `--iters` launches after warm-up, on buffers rotated through > 256 MiB so that the
Infinity Cache does not flatter the number. Algorithmic bytes per element follow SURVEY §8(d).
"""

from __future__ import annotations

import argparse
import json
import statistics
import sys
import pathlib

import torch

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))

import fastforward_amd as ff  # noqa: E402

from fastforward_amd import ops  # noqa: E402

HBM_PEAK_GBS = 8000.0


def time_ms(fn, iters: int, warmup: int = 3, reps: int = 16) -> float:
    """Median time of ONE call of fn. `reps` calls are captured into a hipGraph and replayed, so the
    host-side launch path (Python, ctypes, allocator) is outside the timed region: what remains is
    kernel time plus the ~1.5 us kernel-boundary cost."""
    for w in range(warmup):
        fn(w)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    stream = torch.cuda.Stream()
    stream.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(stream):
        with torch.cuda.graph(graph, stream=stream):
            for r in range(reps):
                fn(r)
    torch.cuda.current_stream().wait_stream(stream)
    graph.replay()
    torch.cuda.synchronize()
    times = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        graph.replay()
        b.record()
        b.synchronize()
        times.append(a.elapsed_time(b) / reps)
    return statistics.median(times)


def rotate(make, nbytes_each: int, min_total: int = 768 << 20):
    n = max(2, min(16, -(-min_total // max(nbytes_each, 1))))
    return [make() for _ in range(n)]


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--out", type=str, default="")
    ap.add_argument("--quick", action="store_true")
    args = ap.parse_args()
    dev = "cuda"
    torch.manual_seed(1234)
    results = []

    def record(name, shape, bytes_per_elem, ms):
        numel = 1
        for s in shape:
            numel *= s
        gbs = numel * bytes_per_elem / ms / 1e6
        row = {"kernel": name, "shape": list(shape), "ms": round(ms, 5), "GB/s": round(gbs, 1), "frac_of_8TB/s": round(gbs / HBM_PEAK_GBS, 4)}
        results.append(row)
        print(json.dumps(row), flush=True)

    weight_shapes = [(14336, 4096)] if args.quick else [(4096, 4096), (1024, 4096), (14336, 4096), (4096, 14336)]
    for shape in weight_shapes:
        ws = rotate(lambda: (torch.randn(shape, device=dev) * 0.02).to(torch.bfloat16), shape[0] * shape[1] * 2)
        scale = torch.rand(shape[0], device=dev) * 0.001 + 0.0005
        tile = (1, shape[1])
        # A1 per-channel bf16 -> int8 (3 B/elem) and bf16 -> bf16 container (4 B/elem)
        record("quantize perchannel bf16->i8", shape, 3, time_ms(lambda i: ops.quantize_by_tile(ws[i % len(ws)], scale, tile, 8, torch.int8), args.iters))
        record("quantize perchannel bf16->bf16", shape, 4, time_ms(lambda i: ops.quantize_by_tile(ws[i % len(ws)], scale, tile, 8, torch.bfloat16), args.iters))
        qs = [ops.quantize_by_tile(w, scale, tile, 8, torch.int8) for w in ws]
        # A2 int8 -> bf16 (3 B/elem)
        record("dequantize perchannel i8->bf16", shape, 3, time_ms(lambda i: ops.dequantize_by_tile(qs[i % len(qs)], scale, tile, None, torch.bfloat16), args.iters))
        # A4 row-wise min/max (2 B/elem)
        record("minmax perchannel bf16", shape, 2, time_ms(lambda i: ops.minmax_by_tile(ws[i % len(ws)], tile), args.iters))
        # group-128 W4: quantize to int8 codes (3 B/elem)
        g = shape[1] // 128
        gscale = torch.rand(shape[0] * g, device=dev) * 0.01 + 0.005
        record("quantize group128 bf16->i8 (4b)", shape, 3, time_ms(lambda i: ops.quantize_by_tile(ws[i % len(ws)], gscale, (1, 128), 4, torch.int8), args.iters))
        record("pack int4 (i8 codes -> nibbles)", shape, 1.5, time_ms(lambda i: ops.pack_int4(qs[i % len(qs)], block=128), args.iters))
        del ws, qs

    act_shapes = [(8, 2048, 4096)] if args.quick else [(8, 2048, 4096), (8, 2048, 14336)]
    for shape in act_shapes:
        nbytes = shape[0] * shape[1] * shape[2] * 2
        xs = rotate(lambda: torch.randn(shape, device=dev, dtype=torch.bfloat16), nbytes)
        scale, offset = torch.tensor([0.03], device=dev), torch.tensor([3.0], device=dev)
        record("quantize pertensor bf16->i8", shape, 3, time_ms(lambda i: ops.quantize_by_tile(xs[i % len(xs)], scale, shape, 8, torch.int8, offset), args.iters))
        record("quantize pertensor bf16->bf16", shape, 4, time_ms(lambda i: ops.quantize_by_tile(xs[i % len(xs)], scale, shape, 8, torch.bfloat16, offset), args.iters))
        record("minmax pertensor bf16", shape, 2, time_ms(lambda i: ops.minmax_by_tile(xs[i % len(xs)], shape), args.iters))
        qx = ops.quantize_by_tile(xs[0], scale, shape, 8, torch.int8, offset)
        record("dequantize pertensor i8->bf16", shape, 3, time_ms(lambda i: ops.dequantize_by_tile(qx, scale, shape, offset, torch.bfloat16), args.iters))
        # calibration step: reduce, set range, quantize with the new range (5 B/elem, 2 dependent passes)
        quantizer = ff.nn.LinearQuantizer(8, symmetric=False, quantized_dtype=torch.int8, device=dev)
        with ff.estimate_ranges(torch.nn.ModuleList([quantizer]), ff.range_setting.running_minmax, sync_free=True):
            record("calibration step (minmax+range+quantize)", shape, 5, time_ms(lambda i: quantizer(xs[i % len(xs)]), args.iters))
        del xs

    # A6: W8A8 linear, T = 2048 tokens, the four Llama-3-8B GEMM shapes; TOPS on 2*M*N*K
    for (n, k) in ([(14336, 4096)] if args.quick else [(4096, 4096), (1024, 4096), (14336, 4096), (4096, 14336)]):
        m = 2048
        xq = torch.randint(-128, 128, (m, k), device=dev, dtype=torch.int8)
        wq = torch.randint(-128, 128, (n, k), device=dev, dtype=torch.int8)
        sx, ox = torch.tensor([0.02], device=dev), torch.tensor([4.0], device=dev)
        sw = torch.rand(n, device=dev) * 0.001 + 0.0005
        ms = time_ms(lambda i: ops.linear_w8a8(xq, wq, sx, ox, sw, None, out_dtype=torch.bfloat16), args.iters)
        row = {"kernel": "linear w8a8", "shape": [m, n, k], "ms": round(ms, 5), "TOPS": round(2 * m * n * k / ms / 1e9, 1)}
        results.append(row)
        print(json.dumps(row), flush=True)
        xb, wb = torch.randn(m, k, device=dev, dtype=torch.bfloat16), torch.randn(n, k, device=dev, dtype=torch.bfloat16)
        ms = time_ms(lambda i: torch.nn.functional.linear(xb, wb), args.iters)
        row = {"kernel": "torch bf16 linear (hipBLASLt, for scale)", "shape": [m, n, k], "ms": round(ms, 5), "TFLOPS": round(2 * m * n * k / ms / 1e9, 1)}
        results.append(row)
        print(json.dumps(row), flush=True)

    if args.out:
        pathlib.Path(args.out).parent.mkdir(parents=True, exist_ok=True)
        pathlib.Path(args.out).write_text(json.dumps(results, indent=1))


if __name__ == "__main__":
    main()
