"""Checks the K-loops of the persistent GEMM kernels in a hipcc -S device assembly file: inside every MFMA cluster (the
instructions between `s_setprio 1` and `s_setprio 0`) no run of more than MAX_RUN non-MFMA vector instructions may sit between
two MFMAs, and no scratch (spill) instruction may sit inside the MFMA span. A sched_group_barrier pattern that names an
instruction kind the cluster does not contain is silently dropped by hipcc and the conversion work lands in one lump with the
matrix pipe idle — round 3 shipped exactly that for a whole round; this check reads what the compiler actually emitted.

usage: python tools/asm_cluster_check.py file.s [name-filter] [max-run]      (exit code 1 on a violation)
       python tools/asm_cluster_check.py --build csrc/ffq_wlinear.hip [name-filter] [max-run]
"""
from __future__ import annotations

import pathlib
import re
import subprocess
import sys
import tempfile

ROOT = pathlib.Path(__file__).resolve().parent.parent
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-fast-math",
         f"-I{ROOT / 'include'}", "-S", "--cuda-device-only"]


def build_assembly(source: pathlib.Path) -> str:
    """gfx950 assembly of one kernel source. Cached under the system's temp directory, keyed by the contents of the source, of every
    header beside it and of the flags: the CPU test suite asks for the same file from several test modules (a compile is 20-70 s)."""
    import hashlib

    source = pathlib.Path(source)
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for f in [source, *sorted(source.parent.glob("*.h")), source.parent.parent.parent / "include" / "ffq.h"]:
        h.update(f.name.encode())
        h.update(f.read_bytes())
    cache = pathlib.Path(tempfile.gettempdir()) / "ffq_isa_cache"
    cached = cache / f"{source.stem}-{h.hexdigest()[:24]}.s"
    if cached.exists():
        return cached.read_text()
    with tempfile.TemporaryDirectory() as tmp:
        out = pathlib.Path(tmp) / "kernel.s"
        subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, str(source), "-o", str(out)], check=True, capture_output=True)
        text = out.read_text()
    try:
        cache.mkdir(exist_ok=True)
        partial = cached.with_suffix(f".{__import__('os').getpid()}.tmp")
        partial.write_text(text)
        partial.replace(cached)
    except OSError:
        pass  # a read-only temp directory: no cache
    return text


def clusters_of(text: str, name_filter: str = ""):
    """(kernel name, list of clusters, scratch instructions inside the MFMA span); a cluster = list of instruction mnemonics."""
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)\n\s*\.end_amdhsa_kernel", text, re.S | re.M):
        name, body = m.group(1), m.group(2)
        if name_filter not in name:
            continue
        lines = [l.strip() for l in body.split("\n")]
        code = [l for l in lines if l and not l.startswith((";", ".", "//")) and not l.endswith(":")]
        mfma = [i for i, l in enumerate(code) if l.startswith("v_mfma")]
        if not mfma:
            continue
        scratch = sum(1 for l in code[mfma[0]:mfma[-1] + 1] if l.startswith("scratch_"))
        clusters, current = [], None
        for l in code:
            if l.startswith("s_setprio 1"):
                current = []
            elif l.startswith("s_setprio 0"):
                if current is not None and any(x.startswith("v_mfma") for x in current):
                    clusters.append(current)
                current = None
            elif current is not None:
                current.append(l.split()[0])
        yield name, clusters, scratch


def longest_vector_run(cluster: list[str]) -> int:
    """Longest run of VALU / LDS instructions between the first and the last MFMA of a cluster."""
    idx = [i for i, op in enumerate(cluster) if op.startswith("v_mfma")]
    best = run = 0
    for op in cluster[idx[0]:idx[-1] + 1]:
        if op.startswith("v_mfma"):
            run = 0
        elif op.startswith(("v_", "ds_")):
            run += 1
            best = max(best, run)
    return best


def check(text: str, name_filter: str = "", max_run: int = 6) -> list[str]:
    problems = []
    for name, clusters, scratch in clusters_of(text, name_filter):
        worst = max((longest_vector_run(c) for c in clusters), default=0)
        if worst > max_run:
            problems.append(f"{name}: {worst} vector instructions in a row between two MFMAs of a cluster (limit {max_run})")
        if scratch:
            problems.append(f"{name}: {scratch} scratch instructions inside the MFMA span")
    return problems


def main() -> None:
    args = sys.argv[1:]
    if args and args[0] == "--build":
        text = build_assembly(ROOT / "fastforward_amd" / args[1] if not pathlib.Path(args[1]).exists() else pathlib.Path(args[1]))
        args = args[2:]
    else:
        text = pathlib.Path(args[0]).read_text()
        args = args[1:]
    flt = args[0] if args else ""
    max_run = int(args[1]) if len(args) > 1 else 6
    for name, clusters, scratch in clusters_of(text, flt):
        runs = [longest_vector_run(c) for c in clusters]
        print(f"{name[:80]:80s} clusters {len(clusters):3d} longest vector run per cluster {runs} scratch in span {scratch}")
    problems = check(text, flt, max_run)
    for p in problems:
        print("PROBLEM:", p)
    sys.exit(1 if problems else 0)


if __name__ == "__main__":
    main()
