"""The N>1 path on CPU: 2 processes, gloo backend, the oracle standing in for the HIP library.

Checks the claims of fastforward_amd/distributed.py: one all-reduce of [mins | -maxes | -flag],
bit-identical quantizer parameters on every rank, and — with disable_quantization=True — equality
with a sequential single-process calibration over all batches.
"""

import socket
import sys
import traceback

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _build(disable_quantization_ranges=None):
    from fastforward_amd import llama

    cfg = llama.LlamaConfig(hidden_size=64, intermediate_size=160, num_layers=2, num_heads=4, num_kv_heads=2, vocab_size=97)
    model = llama.build_model(cfg, "cpu", torch.float32, seed=7, std=0.2)
    llama.quantize_llama(model, quantized_dtype=None)
    return cfg, model


def _batches(cfg, n=6):
    g = torch.Generator().manual_seed(11)
    return [torch.randint(0, cfg.vocab_size, (2, 16), generator=g) for _ in range(n)]


def _worker(rank: int, world: int, port: int, mode: str, out_queue) -> None:
    try:
        sys.path.insert(0, str(ROOT))
        sys.path.insert(0, str(ROOT / "tests"))
        torch.set_num_threads(1)
        import fastforward_amd as ff

        from conftest import load_oracle, use_backend
        from fastforward_amd import distributed as ffd

        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
        with use_backend(load_oracle()):
            cfg, model = _build()
            batches = _batches(cfg, n=1 if mode == "starved" else 6)  # starved: rank 1 gets no batch at all
            if mode == "inf" and rank == 1:
                with torch.no_grad():  # makes the input of down_proj infinite on this rank only
                    model.layers[0].mlp.up_proj.weight[0, 0] = float("inf")
            try:
                payload = ffd.calibrate_sharded(model, ffd.shard(batches, rank, world), disable_quantization=(mode != "quantized"))
                error = None
            except NotImplementedError as e:
                payload, error = -1, str(e)
            fp = ffd.ranges_fingerprint(model)
            gathered = [torch.zeros_like(fp) for _ in range(world)]
            dist.all_gather(gathered, fp)
            result = {"rank": rank, "payload": payload, "error": error, "same_on_all_ranks": all(torch.equal(g, gathered[0]) for g in gathered)}
            if rank == 0 and mode in ("exact", "starved"):
                _, sequential = _build()
                from fastforward_amd import llama

                llama.calibrate(sequential, batches, sync_free=True, disable_quantization=True)
                result["equals_sequential"] = torch.equal(ffd.ranges_fingerprint(sequential), fp)
                ids = batches[0]
                with torch.no_grad(), ff.strict_quantization(False):
                    result["forward_equal"] = torch.equal(model(ids), sequential(ids))
            out_queue.put(result)
        dist.barrier()
        dist.destroy_process_group()
    except Exception:  # pragma: no cover
        out_queue.put({"rank": rank, "exception": traceback.format_exc()})


def _run(mode: str):
    ctx = mp.get_context("spawn")
    queue = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, mode, queue)) for r in range(2)]
    for p in procs:
        p.start()
    results = [queue.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for r in results:
        assert "exception" not in r, r["exception"]
    return sorted(results, key=lambda r: r["rank"])


@pytest.mark.timeout(300)
def test_sharded_calibration_equals_sequential_with_float_forward():
    r0, r1 = _run("exact")
    # 2 layers x 7 per-tensor activation quantizers: 14 mins + 14 maxes + 1 flag word
    assert r0["payload"] == r1["payload"] == 29
    assert r0["same_on_all_ranks"] and r1["same_on_all_ranks"]
    assert r0["equals_sequential"] and r0["forward_equal"]


@pytest.mark.timeout(300)
def test_sharded_calibration_with_quantized_forward_agrees_across_ranks():
    r0, r1 = _run("quantized")
    assert r0["error"] is None and r0["same_on_all_ranks"] and r1["same_on_all_ranks"]


@pytest.mark.timeout(300)
def test_infinite_activation_on_one_rank_raises_on_every_rank():
    r0, r1 = _run("inf")
    assert r0["error"] == "Infinite" and r1["error"] == "Infinite"


@pytest.mark.timeout(300)
def test_rank_without_any_batch_still_takes_part_in_the_exchange():
    """Fewer batches than ranks: the starved rank contributes the neutral elements (+inf, -inf) in a buffer of the same
    length, receives the global ranges and ends with the same parameters as a sequential run (distributed.py:84)."""
    r0, r1 = _run("starved")
    assert r0["error"] is None and r1["error"] is None
    assert r0["payload"] == r1["payload"] == 29
    assert r0["same_on_all_ranks"] and r1["same_on_all_ranks"]
    assert r0["equals_sequential"] and r0["forward_equal"]
