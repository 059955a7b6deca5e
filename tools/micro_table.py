"""BASELINE.md §3's microbenchmark table: A1 / A2 / A4 (and, round 5, A3 per row) on each of the 7 Llama-3-8B weight shapes (per output channel) and the two
activation shapes (per tensor), GPU kernel time from a rocprofv3 kernel trace beside the CPU eager chain's time for the same op.

    rocprofv3 --kernel-trace --output-format csv -d DIR -o micro -- python3 tools/micro_table.py --probe PLAN.json
    python3 tools/micro_table.py --cpu CPU.json
    python3 tools/micro_table.py --merge DIR/.../micro_kernel_trace.csv PLAN.json CPU.json profiles/r03_micro.md

--probe launches, for every (op, shape), WARM + N kernels back to back in a fixed order and writes that order as PLAN.json; --merge
walks the trace in dispatch order, so each row's average is over exactly that row's launches (rocprofv3's own --stats groups by
kernel name, which mixes the shapes). Algorithmic bytes per element: SURVEY 8(d) (A1 bf16 -> int8: 3, A2 int8 -> bf16: 3, A4: 2).
The CPU column is oracle/eager_chain.py (the reference's unfused ATen sequence, _quantizer_impl.py:154-169,181-190, minmax.py:227-237)
on this box's host cores — the baseline, not the target.
"""

from __future__ import annotations

import csv
import json
import pathlib
import statistics
import sys
import time

import torch

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

WEIGHTS = [("q_proj", 4096, 4096), ("k_proj", 1024, 4096), ("v_proj", 1024, 4096), ("o_proj", 4096, 4096),
           ("gate_proj", 14336, 4096), ("up_proj", 14336, 4096), ("down_proj", 4096, 14336)]
ACTIVATIONS = [("hidden [8,2048,4096]", (8, 2048, 4096)), ("mlp [8,2048,14336]", (8, 2048, 14336))]
WARM, N = 3, 20
HBM_PEAK_GBS = 8000.0


def cases():
    for name, n, k in WEIGHTS:
        yield f"{name} [{n},{k}]", (n, k), (1, k)
    for name, shape in ACTIVATIONS:
        yield name, shape, shape


def probe(plan_path: str) -> None:
    from fastforward_amd import ops

    dev = "cuda"
    plan = []
    marker = torch.arange(7, device=dev)
    for label, shape, tile in cases():
        torch.manual_seed(len(label))
        per_tensor = tile == shape
        xs = [(torch.randn(shape, device=dev) * (1.0 if per_tensor else 0.02)).to(torch.bfloat16) for _ in range(3)]  # rotate: > 256 MiB for the big ones
        lo, hi = ops.minmax_by_tile(xs[0], tile)
        scale, offset = ops.parameters_for_range(lo, hi, 8, not per_tensor, True)
        offset = offset if per_tensor else None
        codes = [ops.quantize_by_tile(x, scale, tile, 8, torch.int8, offset) for x in xs]
        torch.cuda.synchronize()
        row_tile = tuple([1] * (len(shape) - 1) + [shape[-1]])  # A3: one parameter pair per row (per output channel / per token)
        for op, fn in (("A1 quantize bf16->int8", lambda r: ops.quantize_by_tile(xs[r % 3], scale, tile, 8, torch.int8, offset)),
                       ("A2 dequantize int8->bf16", lambda r: ops.dequantize_by_tile(codes[r % 3], scale, tile, offset, torch.bfloat16)),
                       ("A4 min/max", lambda r: ops.minmax_by_tile(xs[r % 3], tile)),
                       ("A3 dynamic quantize per row bf16->int8", lambda r: ops.quantize_dynamic_by_tile(xs[r % 3], row_tile, 8, False, True, torch.int8))):
            marker.flip(0)  # a kernel nothing else here launches: marks the start of a measured row in the trace
            for r in range(WARM + N):
                fn(r)
            torch.cuda.synchronize()
            numel = 1
            for s in shape:
                numel *= s
            plan.append({"label": label, "op": op, "numel": numel, "launches": WARM + N, "bytes_per_elem": 2 if op.startswith("A4") else 3})
        del xs, codes
    pathlib.Path(plan_path).write_text(json.dumps(plan))
    print(f"probe: {len(plan)} rows")


def cpu(out_path: str) -> None:
    sys.path.insert(0, str(ROOT / "oracle"))
    import eager_chain

    rows = {}
    for label, shape, tile in cases():
        torch.manual_seed(len(label))
        per_tensor = tile == shape
        x = (torch.randn(shape) * (1.0 if per_tensor else 0.02)).to(torch.bfloat16)
        lo, hi = eager_chain.minmax(x, tile)
        scale, offset = eager_chain.parameters_for_range(lo, hi, 8, not per_tensor, True)
        codes = eager_chain.quantize(x, scale, tile, 8, torch.int8, offset)
        row_tile = tuple([1] * (len(shape) - 1) + [shape[-1]])

        def dynamic():  # quantize_dynamic_by_tile_impl (_quantizer_impl.py:243-285): min, max, parameters_for_range, round(offset), quantize
            lo_r, hi_r = eager_chain.minmax(x, row_tile)
            s_r, o_r = eager_chain.parameters_for_range(lo_r, hi_r, 8, False, True)
            return eager_chain.quantize(x, s_r, row_tile, 8, torch.int8, torch.round(o_r))

        for op, fn in (("A1 quantize bf16->int8", lambda: eager_chain.quantize(x, scale, tile, 8, torch.int8, offset)),
                       ("A2 dequantize int8->bf16", lambda: eager_chain.dequantize(codes, scale, tile, offset, torch.bfloat16)),
                       ("A4 min/max", lambda: eager_chain.minmax(x, tile)),
                       ("A3 dynamic quantize per row bf16->int8", dynamic)):
            fn()
            times = []
            for _ in range(3):
                t0 = time.perf_counter()
                fn()
                times.append(time.perf_counter() - t0)
            rows[f"{label}|{op}"] = round(statistics.median(times) * 1e3, 3)
            print(label, op, rows[f"{label}|{op}"], "ms", flush=True)
    import os
    pathlib.Path(out_path).write_text(json.dumps({"threads": torch.get_num_threads(), "host_cpus": os.cpu_count(), "ms": rows}))


def merge(trace_csv: str, plan_path: str, cpu_path: str, out_md: str) -> None:
    plan = json.loads(pathlib.Path(plan_path).read_text())
    cpu_ms = json.loads(pathlib.Path(cpu_path).read_text()) if pathlib.Path(cpu_path).exists() else {"ms": {}, "threads": None, "host_cpus": None}
    with open(trace_csv) as f:
        rows = list(csv.DictReader(f))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    key = {"A1": ("ffq::quantize_",), "A2": ("ffq::dequantize_",), "A4": ("ffq::minmax_",), "A3": ("ffq::quantize_dynamic",)}  # NB "dequantize_" contains "quantize_"
    # the measured launches of each plan row: the LAST N groups of its kernels before the next row's kernels begin
    pos = 0
    tag = pathlib.Path(out_md).name.split("_")[0]  # profiles/rNN_micro.md
    out = [f"# {tag} — A1 / A2 / A4 per shape: rocprofv3 kernel trace (GPU) beside the CPU eager chain", "",
           f"`rocprofv3 --kernel-trace -- python3 tools/micro_table.py --probe` on one MI355X: per row {N} launches after {WARM} warm-up launches, inputs rotated over 3 tensors; "
           "avg = mean kernel duration from the trace (A4 = every kernel of the call summed: one launch where the last block to arrive finishes, else partial + finalize; A3 = quantize_dynamic_by_tile with one parameter pair per ROW of the tensor, asymmetric: one launch). Algorithmic bytes per element: A1 3, A2 3, A4 2, A3 3 (SURVEY 8(d)); "
           f"peak 8000 GB/s. CPU column: `oracle/eager_chain.py` (the reference's unfused ATen chain) with {cpu_ms.get('threads')} torch threads on {cpu_ms.get('host_cpus')} host CPUs, median of 3.", "",
           "| tensor | op | kernels per launch | GPU avg us | GB/s | frac of 8 TB/s | CPU eager ms | GPU / CPU |", "|---|---|---:|---:|---:|---:|---:|---:|"]
    for row in plan:
        want = key[row["op"][:2]]
        mine = []
        while pos < len(rows) and "flip" not in rows[pos]["Kernel_Name"]:
            pos += 1  # the previous row's tail, the next tensor's set-up launches
        pos += 1      # the marker itself
        while pos < len(rows) and (any(k in rows[pos]["Kernel_Name"] for k in want) or (row["op"].startswith("A4") and "finalize" in rows[pos]["Kernel_Name"])):
            mine.append(rows[pos])
            pos += 1
        per_launch = max(1, round(len(mine) / row["launches"]))
        groups = [mine[i:i + per_launch] for i in range(0, len(mine) - per_launch + 1, per_launch)][-N:]
        if not groups:
            out.append(f"| {row['label']} | {row['op']} | - | (no launches found in the trace) | | | | |")
            continue
        us = statistics.mean(sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in g) / 1e3 for g in groups)
        gbs = row["numel"] * row["bytes_per_elem"] / us / 1e3
        c = cpu_ms["ms"].get(f"{row['label']}|{row['op']}")
        out.append(f"| {row['label']} | {row['op']} | {per_launch} | {us:.2f} | {gbs:.0f} | {gbs / HBM_PEAK_GBS:.3f} | {c if c is not None else '-'} | "
                   f"{(c * 1e3 / us):.0f}x |" if c is not None else f"| {row['label']} | {row['op']} | {per_launch} | {us:.2f} | {gbs:.0f} | {gbs / HBM_PEAK_GBS:.3f} | - | - |")
    pathlib.Path(out_md).write_text("\n".join(out) + "\n")
    print("\n".join(out[4:]))


if __name__ == "__main__":
    mode = sys.argv[1]
    if mode == "--probe":
        probe(sys.argv[2])
    elif mode == "--cpu":
        cpu(sys.argv[2])
    else:
        merge(*sys.argv[2:6])
