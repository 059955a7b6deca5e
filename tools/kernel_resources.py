"""Per-kernel register / spill / scratch / LDS usage of the device code inside a built library (or a hipcc --save-temps .s file).

A shared library built by hipcc carries one clang offload bundle per translation unit in its `.hip_fatbin` section; each
bundle holds the gfx950 code object whose AMDGPU metadata note lists, per kernel, `.vgpr_count`, `.agpr_count`,
`.sgpr_count`, `.vgpr_spill_count`, `.sgpr_spill_count`, `.private_segment_fixed_size` (scratch) and
`.group_segment_fixed_size` (static LDS). This tool extracts them with nothing but Python + `llvm-readelf`.

usage: python tools/kernel_resources.py [library.so | file.s] [name-filter]
"""
from __future__ import annotations

import pathlib
import re
import shutil
import struct
import subprocess
import sys
import tempfile

ROOT = pathlib.Path(__file__).resolve().parent.parent
DEFAULT_LIBRARY = ROOT / "fastforward_amd" / "csrc" / "libffq_hip.so"
_MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
_FIELDS = ("vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size", "group_segment_fixed_size")


def readelf() -> str | None:
    for candidate in ("/opt/rocm/lib/llvm/bin/llvm-readelf", shutil.which("llvm-readelf") or ""):
        if candidate and pathlib.Path(candidate).exists():
            return candidate
    return None


def code_objects(library: pathlib.Path) -> list[bytes]:
    """The gfx950 ELF images of every offload bundle in `library`."""
    blob = library.read_bytes()
    images, at = [], 0
    while (at := blob.find(_MAGIC, at)) >= 0:
        (count,) = struct.unpack_from("<Q", blob, at + len(_MAGIC))
        cursor = at + len(_MAGIC) + 8
        for _ in range(count):
            offset, size, triple_size = struct.unpack_from("<QQQ", blob, cursor)
            triple = blob[cursor + 24:cursor + 24 + triple_size].decode()
            cursor += 24 + triple_size
            if "gfx950" in triple and size:
                images.append(blob[at + offset:at + offset + size])
        at = cursor
    return images


def _parse(text: str) -> list[dict[str, object]]:
    kernels = []
    for block in re.findall(r"- \.agpr_count:.*?(?=\n\s*- \.agpr_count:|\namdhsa\.target|\Z)", text, re.S):
        entry: dict[str, object] = {}
        name = re.search(r"\.name:\s+(\S+)", block)
        if not name:
            continue
        entry["name"] = name.group(1)
        for field in _FIELDS:
            m = re.search(rf"\.{field}:\s+(\d+)", block)
            entry[field] = int(m.group(1)) if m else 0
        kernels.append(entry)
    return kernels


def kernel_resources(source: pathlib.Path = DEFAULT_LIBRARY) -> list[dict[str, object]]:
    """One dict per kernel: name (mangled) and the integer fields of `_FIELDS`."""
    source = pathlib.Path(source)
    if source.suffix == ".s":
        return _parse(source.read_text())
    tool = readelf()
    if tool is None:
        raise RuntimeError("llvm-readelf not found")
    kernels: list[dict[str, object]] = []
    with tempfile.TemporaryDirectory() as tmp:
        for i, image in enumerate(code_objects(source)):
            path = pathlib.Path(tmp) / f"co{i}.elf"
            path.write_bytes(image)
            notes = subprocess.run([tool, "--notes", str(path)], capture_output=True, text=True, check=True).stdout
            kernels += _parse(notes)
    return kernels


def demangle(names: list[str]) -> list[str]:
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    return [n.replace("ffq::", "").replace("void ", "") for n in out]


def main() -> None:
    args = [a for a in sys.argv[1:]]
    source = pathlib.Path(args[0]) if args and pathlib.Path(args[0]).exists() else DEFAULT_LIBRARY
    flt = args[-1] if args and not pathlib.Path(args[-1]).exists() else ""
    kernels = [k for k in kernel_resources(source) if flt in str(k["name"])]
    for k, dem in zip(kernels, demangle([str(k["name"]) for k in kernels])):
        dem = re.sub(r"\(ffq::\w+Args, int\)|\(\w+Args, int\)", "", dem)
        print(f"{dem[:100]:100s} vgpr {k['vgpr_count']:4d} agpr {k['agpr_count']:4d} sgpr {k['sgpr_count']:4d} spill {k['vgpr_spill_count']:3d}/{k['sgpr_spill_count']:3d} "
              f"scratch {k['private_segment_fixed_size']:5d} lds {k['group_segment_fixed_size']}")
    print(f"{len(kernels)} kernels; {sum(1 for k in kernels if k['vgpr_spill_count'])} with spilled VGPRs")


if __name__ == "__main__":
    main()
