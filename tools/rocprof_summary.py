"""Summarise a rocprofv3 run (rocpd sqlite `*_results.db`, or `*_kernel_stats.csv`) as a small table.

    python tools/rocprof_summary.py gpurun_out/prof_bench/bench_results.db profiles/r01_bench_kernel_stats.md "title"

Kernel names are shortened (template arguments of torch's elementwise kernels run to kilobytes).
"""

from __future__ import annotations

import csv
import pathlib
import re
import sqlite3
import sys


def short(name: str, width: int = 110) -> str:
    name = re.sub(r"\s+", " ", name)
    name = name.replace("void ", "")
    m = re.match(r"at::native::(?:\(anonymous namespace\)::)?(\w+)<.*?(\w+(?:Functor|_kernel_cuda|kernel_impl|Ops)\w*)", name)
    if m and len(name) > width:
        name = f"at::native::{m.group(1)}<...{m.group(2)}...>"
    return name if len(name) <= width else name[: width - 3] + "..."


def rows_from_db(path: str):
    db = sqlite3.connect(path)
    return [(n, int(c), float(t), float(a), float(p)) for n, c, t, a, p in db.execute("select name, total_calls, total_duration, average, percentage from top_kernels")]


def rows_from_csv(path: str):
    out = []
    with open(path) as f:
        for r in csv.DictReader(f):
            out.append((r["Name"], int(r["Calls"]), float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
    return out


def main() -> None:
    src, dst = sys.argv[1], pathlib.Path(sys.argv[2])
    title = sys.argv[3] if len(sys.argv) > 3 else src
    rows = rows_from_db(src) if src.endswith(".db") else rows_from_csv(src)
    rows.sort(key=lambda r: -r[2])
    total = sum(r[2] for r in rows)
    ours = sum(r[2] for r in rows if "ffq::" in r[0])
    lines = [f"# {title}", "", f"source: `{src}` (rocprofv3 --kernel-trace --stats); durations in microseconds", "",
             f"total kernel time {total / 1e3:.1f} ms over {sum(r[1] for r in rows)} dispatches; `ffq::` kernels {ours / 1e3:.1f} ms ({100 * ours / total:.1f} %)", "",
             "| kernel | calls | total us | avg us | % |", "|---|---:|---:|---:|---:|"]
    for name, calls, tot, avg, pct in rows[:40]:
        lines.append(f"| `{short(name)}` | {calls} | {tot:.0f} | {avg:.2f} | {pct:.2f} |")
    dst.parent.mkdir(parents=True, exist_ok=True)
    dst.write_text("\n".join(lines) + "\n")
    print(f"wrote {dst} ({len(rows)} kernels)")


if __name__ == "__main__":
    main()
