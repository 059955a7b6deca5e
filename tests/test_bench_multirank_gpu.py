"""`python bench.py --gpus 2` end to end, as the driver starts it: a FRESH process that launches two ranks itself. One GPU is
enough for the control flow (FFQ_DIST_BACKEND=gloo lets two ranks share it; RCCL wants one device per rank): rendezvous,
batch-sharded calibration, the ONE all-reduce of the activation ranges, the cross-rank check of the resulting parameters,
max-over-ranks timing, one JSON line from rank 0. The 8-GPU curve itself is the driver's to measure."""

import json
import os
import pathlib
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = pathlib.Path(__file__).resolve().parent.parent


def _bench(*args: str, backend: str | None = "gloo") -> dict:
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for key in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "FFQ_DIST_BACKEND"):
        env.pop(key, None)
    if backend is not None:
        env["FFQ_DIST_BACKEND"] = backend
    proc = subprocess.run([sys.executable, str(ROOT / "bench.py"), *args], capture_output=True, text=True, env=env, timeout=900, cwd=ROOT)
    assert proc.returncode == 0, proc.stderr[-3000:]
    lines = [line for line in proc.stdout.splitlines() if line.startswith("{")]
    assert len(lines) == 1, f"exactly one JSON line expected from rank 0, got {len(lines)}"
    return json.loads(lines[0])


def test_two_ranks_tiny_model():
    line = _bench("--gpus", "2", "--model", "tiny", "--steps", "2", "--warmup", "1", "--batch", "2", "--seq-len", "64", "--calib-seqs", "4", "--no-side-measurements")
    assert line["n_gpus"] == 2 and line["world_size"] == 2 and line["scaling"] == "weak"
    cal = line["calibration"]
    assert cal["ranks_seen"] == 2 and cal["ranges_identical_across_ranks"] is True
    assert cal["allreduce_floats"] == 2 * 14 + 1 and cal["all_reduce_us"] > 0 and cal["all_reduce_backend"] == "gloo"
    assert line["config"]["tokens_per_step"] == 2 * 2 * 64 and line["value"] > 0


def test_two_ranks_llama3_70b_shapes_two_layers():
    """BASELINE configs[4]'s recipe at its real widths (8192 / 28672, 64 + 8 heads of 128), two decoder layers per rank."""
    line = _bench("--gpus", "2", "--model", "llama3-70b", "--layers", "2", "--steps", "1", "--warmup", "1", "--batch", "1", "--seq-len", "256",
                  "--calib-seqs", "2", "--no-side-measurements")
    cal = line["calibration"]
    assert line["n_gpus"] == 2 and cal["ranks_seen"] == 2 and cal["ranges_identical_across_ranks"] is True
    assert cal["allreduce_floats"] == 2 * 14 + 1 and line["value"] > 0


def test_one_rank_through_rccl():
    """RCCL itself, once: a one-rank "nccl" group (bench.py --gpus 1 --force-dist) sends the calibrated model's REAL range buffer
    through all_reduce(MIN) on fp32 and the cross-rank check through MIN / MAX / SUM on int32 — the collectives and the backend an
    8-GPU run uses, executed on the one GPU this box has; the scaling curve itself is the driver's to measure."""
    line = _bench("--gpus", "1", "--force-dist", "--model", "tiny", "--steps", "2", "--warmup", "1", "--batch", "2", "--seq-len", "64", "--calib-seqs", "4",
                  "--no-side-measurements", backend=None)
    assert line["n_gpus"] == 1 and line["world_size"] == 1 and line["collective_backend"].startswith("nccl")
    cal = line["calibration"]
    assert cal["all_reduce_backend"] == "nccl" and cal["all_reduce_us"] > 0 and cal["allreduce_floats"] == 2 * 14 + 1
    assert cal["ranks_seen"] == 1 and cal["ranges_identical_across_ranks"] is True and line["value"] > 0
