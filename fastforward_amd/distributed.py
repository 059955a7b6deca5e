"""Batch-sharded calibration over the GPUs of one node: one process per GPU, ONE all-reduce.

The reference has no multi-device code; its calibration loop is a plain Python loop
over batches (docs/examples/quick_start_quantize_llms.nb.py:193,255). That loop shards naturally:

  * every rank holds a full replica of the model and the quantizers and runs RunningMinMax
    (A4) over ITS share of the calibration batches, without touching the host
    (``sync_free=True``);
  * weight-quantizer ranges depend only on the replicated weights -> identical on every rank,
    nothing to exchange;
  * activation-quantizer ranges are partial: running min / running max are associative and
    commutative, so the global range is min over ranks of the mins and max over ranks of the maxes.
    All activation quantizers are packed into one fp32 buffer ``[mins | -maxes | -flags]`` and
    reduced by ONE ``all_reduce(MIN)`` — RCCL over xGMI with the "nccl" backend on ROCm. For
    Llama-3-8B that is 224 quantizers -> 449 floats (1.8 KB); for 70B 1121 floats (4.5 KB). The
    message is latency-bound, so link bandwidth and ring-vs-tree are irrelevant;
  * every rank then runs A5 (range -> scale/offset, including the GLOBAL one-sided decision and the
    Inf check) on the reduced range and obtains bit-identical parameters.

Exactness: with ``disable_quantization=True`` (ranges collected on an un-quantized forward,
reference range_setting/common.py:236-238) the sharded result equals a sequential single-process
calibration over all batches bit-for-bit. With quantize-while-calibrating (the reference default)
each quantizer's input depends on the running ranges upstream, so the result is order-dependent
even on one device; sharding is then statistically equivalent, not bit-equal.

Inference after calibration is pure data parallelism: replicas, no collective.
"""

from __future__ import annotations

import os
import time

from typing import Iterable, Sequence

import torch
import torch.distributed as dist

import fastforward_amd as ff

from fastforward_amd import ops
from fastforward_amd.nn.quantizer import Quantizer
from fastforward_amd.range_setting.minmax import RunningMinMaxEstimator


def init_process_group_from_env(backend: str | None = None, force: bool = False) -> tuple[int, int, int]:
    """(rank, local_rank, world_size) from RANK / LOCAL_RANK / WORLD_SIZE; initialises the default
    group with "nccl" (= RCCL) when a HIP device is present, else "gloo". A single process needs no group; ``force`` creates
    the one-rank group anyway, so that the collectives of this module run through the backend (RCCL executes the very
    all_reduce(MIN) kernel an 8-GPU run issues — what a one-GPU box can prove about the multi-GPU path)."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:  # FFQ_DIST_BACKEND=gloo: several ranks on one GPU (a control-flow check; RCCL wants one device per rank)
            backend = os.environ.get("FFQ_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


# what the most recent range exchange of this process looked like (bench.py reports it): wall time of the all_reduce itself
# between two device drains, payload, ranks, backend
last_exchange: dict[str, object] = {}


def shard(items: Sequence, rank: int, world: int) -> list:
    """Round-robin share of `items` for `rank` (equal counts when len(items) % world == 0)."""
    return [item for i, item in enumerate(items) if i % world == rank]


def _activation_estimators(model: torch.nn.Module) -> list[tuple[Quantizer, RunningMinMaxEstimator]]:
    """(quantizer, estimator) for every activation quantizer under calibration, in module order —
    the same order on every rank because every rank holds the same model."""
    pairs = []
    for _, quantizer in ff.nn.named_quantizers(model):
        meta = quantizer.quant_metadata
        if meta is not None and meta.parameter_quantizer:
            continue  # weights are replicated: their ranges are already identical everywhere
        for fn in quantizer.overrides:
            if isinstance(fn, RunningMinMaxEstimator):
                pairs.append((quantizer, fn))
    return pairs


def _range_entries(quantizer: Quantizer, estimator: RunningMinMaxEstimator) -> int:
    """Number of (min, max) pairs this quantizer contributes to the exchange — a function of the MODEL, not of what this
    rank happened to see, so that every rank builds a buffer of the same length: the estimator's own count when it saw
    data, else the quantizer's materialised parameter count, else 1 for a per-tensor quantizer."""
    if estimator.min is not None:
        return estimator.min.numel()
    scale = getattr(quantizer, "scale", None)
    if isinstance(scale, torch.Tensor) and not isinstance(scale, torch.nn.parameter.UninitializedTensorMixin):
        return scale.numel()
    if getattr(quantizer, "per_tensor", False):
        return 1
    raise RuntimeError(
        "all_reduce_ranges: a rank saw no data for a non-per-tensor activation quantizer that was never initialised, so the "
        "size of its range is unknown on that rank; give every rank at least one calibration batch"
    )


def all_reduce_ranges(model: torch.nn.Module, group: dist.ProcessGroup | None = None) -> int:
    """Reduce the running (min, max) of all activation quantizers across ranks with one collective
    and set the global range on every quantizer. Returns the number of floats exchanged.

    The buffer is laid out from the model's quantizer list, identical on all ranks. An estimator that saw no data on
    this rank (fewer batches than ranks, or a quantizer no batch reached) contributes the neutral elements +inf / -inf;
    a quantizer that NO rank reached keeps its uninitialised range."""
    pairs = _activation_estimators(model)
    if not pairs:
        return 0
    device = next((e.min.device for _, e in pairs if e.min is not None), None)
    if device is None:
        device = next((p.device for p in model.parameters()), torch.device("cpu"))
    counts = [_range_entries(q, e) for q, e in pairs]
    inf = float("inf")
    mins = [e.min.detach().reshape(-1).to(torch.float32) if e.min is not None else torch.full((k,), inf, device=device) for (_, e), k in zip(pairs, counts)]
    maxs = [e.max.detach().reshape(-1).to(torch.float32) if e.max is not None else torch.full((k,), -inf, device=device) for (_, e), k in zip(pairs, counts)]
    statuses = torch.stack([e.status.to(device) if e.status is not None else torch.zeros(1, dtype=torch.int32, device=device) for _, e in pairs])
    any_inf = (statuses & ops.FLAG_INF).max().to(torch.float32).reshape(1)  # 1.0 if any quantizer saw +-Inf
    packed = torch.cat(mins + [-m for m in maxs] + [-any_inf])
    if dist.is_initialized():  # a one-rank group (init_process_group_from_env(force=True)) runs the collective too
        timed = packed.is_cuda
        if timed:  # the caller reads a flag off the result right below anyway: one more drain costs nothing and gives the time
            torch.cuda.synchronize(packed.device)
        t0 = time.perf_counter()
        dist.all_reduce(packed, op=dist.ReduceOp.MIN, group=group)  # THE collective of this path
        if timed:
            torch.cuda.synchronize(packed.device)
        last_exchange.update(seconds=time.perf_counter() - t0, floats=int(packed.numel()), world_size=dist.get_world_size(group),
                             backend=dist.get_backend(group))
    n = sum(counts)
    lo_all, hi_all, any_inf_anywhere = packed[:n], -packed[n : 2 * n], bool(-packed[-1].item() > 0)
    if any_inf_anywhere:
        raise NotImplementedError("Infinite")  # reference range_setting/minmax.py:233-234, any rank
    unseen = (lo_all == inf) & (hi_all == -inf)  # the neutral elements survived: no rank saw data for that entry
    any_unseen = bool(unseen.any()) if any(e.min is None for _, e in pairs) else False
    at = 0
    for (quantizer, estimator), k in zip(pairs, counts):
        lo, hi = lo_all[at : at + k], hi_all[at : at + k]
        at += k
        if estimator.min is None and any_unseen and bool(unseen[at - k : at].all()):
            continue
        dtype = estimator.min.dtype if estimator.min is not None else torch.float32
        estimator.min = lo.to(dtype).contiguous()
        estimator.max = hi.to(dtype).contiguous()
        quantizer.quantization_range = (estimator.min, estimator.max)  # A5 on the global range
    return packed.numel()


def calibrate_sharded(
    model: torch.nn.Module,
    local_batches: Iterable[torch.Tensor],
    disable_quantization: bool = True,
    group: dist.ProcessGroup | None = None,
    fused: bool = False,
) -> int:
    """RunningMinMax calibration of `model` on this rank's batches followed by the range all-reduce.

    `local_batches` is this rank's share (see :func:`shard`). Returns the all-reduce payload size in
    floats. Nothing waits for the device until the single flag read after the collective.
    """
    forward = None
    if fused:  # Llama harness only: the producers between the quantizers as one-pass kernels
        from fastforward_amd.llama import FusedCalibrationForward

        forward = FusedCalibrationForward(model)
    with torch.no_grad(), ff.strict_quantization(False):
        with ff.estimate_ranges(model, ff.range_setting.running_minmax, sync_free=True, disable_quantization=disable_quantization):
            for batch in local_batches:
                if forward is not None:
                    forward(batch)
                else:
                    model(batch, logits=False) if _accepts_logits(model) else model(batch)
            _estimate_unseen_weight_ranges(model)
            return all_reduce_ranges(model, group)


def _estimate_unseen_weight_ranges(model: torch.nn.Module) -> None:
    """A rank whose share held no batch never ran its weight quantizers. Their ranges depend on the replicated weights
    only, so run each one once on its own weight (one estimator step of the same data every other rank saw: the same
    running min / max) — the rank then holds the parameters everyone else holds without any exchange."""
    for module in model.modules():
        quantizer, weight = getattr(module, "weight_quantizer", None), getattr(module, "weight", None)
        if not isinstance(quantizer, Quantizer) or not isinstance(weight, torch.Tensor):
            continue
        if any(isinstance(fn, RunningMinMaxEstimator) and fn.min is None for fn in quantizer.overrides):
            quantizer(weight)


def _accepts_logits(model: torch.nn.Module) -> bool:
    from fastforward_amd.llama import LlamaModel

    return isinstance(model, LlamaModel)


def ranges_fingerprint(model: torch.nn.Module) -> torch.Tensor:
    """All quantizer parameters flattened into one fp32 vector (for cross-rank equality checks)."""
    parts = []
    for _, quantizer in ff.nn.named_quantizers(model):
        for name in ("scale", "offset"):
            t = getattr(quantizer, name, None)
            if isinstance(t, torch.Tensor) and not isinstance(t, torch.nn.parameter.UninitializedTensorMixin):
                parts.append(t.detach().reshape(-1).to(torch.float32))
    return torch.cat(parts) if parts else torch.zeros(0)


def ranges_agree_across_ranks(model: torch.nn.Module, group: dist.ProcessGroup | None = None) -> tuple[bool, int]:
    """(every rank holds bit-identical quantizer parameters, number of ranks that took part). Two more small collectives on
    the fingerprint vector (element-wise MIN and MAX over ranks agree exactly where all ranks agree) and a SUM of ones;
    a self-check for multi-GPU runs, not part of the calibration path."""
    fingerprint = ranges_fingerprint(model)
    if not dist.is_initialized():
        return True, 1
    bits = fingerprint.view(torch.int32).clone()  # compare bit patterns: NaNs and signed zeros included
    lo, hi = bits.clone(), bits.clone()
    ones = torch.ones(1, dtype=torch.int32, device=bits.device)
    dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
    dist.all_reduce(ones, op=dist.ReduceOp.SUM, group=group)
    return bool(torch.equal(lo, hi)), int(ones.item())
