"""pytest configuration: the `gpu` marker, the oracle loader and the fixture files.

`-m "not gpu"` runs on a machine without a GPU: the oracle against the golden vectors, the host
logic (driven through the oracle, which exports the same C ABI on host pointers), and the symbol
table of libffq_hip.so. `-m gpu` runs the parity tests proper on an MI355X, through the C ABI of the
HIP library. The oracle is only ever loaded from here (tests), __graft_entry__.smoke() and
bench.py's cpu_baseline leg — never by the package.
"""

from __future__ import annotations

import contextlib
import os
import pathlib
import subprocess
import sys

import pytest
import torch

ROOT = pathlib.Path(__file__).resolve().parent.parent
GOLDEN = ROOT / "tests" / "golden"
ORACLE_SO = ROOT / "oracle" / "_build" / "libffq_oracle.so"
HIP_SO = ROOT / "fastforward_amd" / "csrc" / "libffq_hip.so"

sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(GOLDEN))


def pytest_configure(config: pytest.Config) -> None:
    config.addinivalue_line("markers", "gpu: needs an MI355X (run with -m gpu on the GPU box)")


def load_oracle():
    from fastforward_amd._cabi import FFQLibrary

    override = os.environ.get("FFQ_ORACLE_SO")  # e.g. a sanitizer build of the oracle (oracle/sanitize.sh): the checker checked
    if override:
        return FFQLibrary(pathlib.Path(override))
    if not ORACLE_SO.exists() or ORACLE_SO.stat().st_mtime < (ROOT / "oracle" / "ffq_oracle.c").stat().st_mtime:
        subprocess.run(["make", "-C", str(ROOT / "oracle")], check=True, capture_output=True)
    return FFQLibrary(ORACLE_SO)


def use_backend(lib):
    """Temporarily make `lib` the library fastforward_amd.ops calls into (test-only; oracle/inject.py)."""
    sys.path.insert(0, str(ROOT / "oracle"))
    import inject

    return inject.use_library(lib)


@pytest.fixture(scope="session")
def oracle_lib():
    return load_oracle()


@pytest.fixture()
def oracle_backend(oracle_lib):
    """Host-logic tests on CPU tensors: the oracle stands in for the HIP library."""
    with use_backend(oracle_lib):
        yield oracle_lib


@pytest.fixture(scope="session")
def hip_lib():
    from fastforward_amd._cabi import FFQLibrary

    return FFQLibrary(HIP_SO)


@pytest.fixture()
def hip_backend(hip_lib):
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    with use_backend(hip_lib):
        yield hip_lib


def golden(name: str):
    return torch.load(GOLDEN / name, weights_only=True)


@pytest.fixture(autouse=True)
def _restore_flags():
    """Flags are process-global; keep tests independent."""
    from fastforward_amd import flags

    saved = dict(flags._VALUES)
    yield
    flags._VALUES.update(saved)
