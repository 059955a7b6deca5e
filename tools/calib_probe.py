"""Wall time of RunningMinMax calibration steps (Llama-3-8B shapes, B=8, S=2048) — under rocprofv3 --kernel-trace --stats the
kernel-time sum next to it says how much of a step the host spends launching."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fastforward_amd as ff
from fastforward_amd import llama
dev = "cuda"
cfg = llama.LlamaConfig.llama3_8b()
model = llama.build_model(cfg, dev, torch.bfloat16, seed=1236)
llama.quantize_llama(model, w_bits=8, a_bits=8, quantized_dtype=torch.int8)
gen = torch.Generator(device=dev).manual_seed(77)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
batches = [torch.randint(0, cfg.vocab_size, (8, 2048), device=dev, generator=gen) for _ in range(steps)]
fwd = llama.FusedProducersForward(model)
from fastforward_amd import distributed as ffd
torch.cuda.synchronize(); t_enter = time.perf_counter()
with torch.no_grad(), ff.strict_quantization(False):
    with ff.estimate_ranges(model, ff.range_setting.running_minmax, sync_free=True):
        torch.cuda.synchronize(); print(f"enter estimate_ranges: {1e3*(time.perf_counter()-t_enter):.1f} ms", flush=True)
        for i, ids in enumerate(batches):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            fwd(ids)
            t1 = time.perf_counter()          # host done enqueueing
            torch.cuda.synchronize(); t2 = time.perf_counter()
            print(f"step {i}: host enqueue {1e3*(t1-t0):.1f} ms, wall {1e3*(t2-t0):.1f} ms", flush=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = ffd.all_reduce_ranges(model)
        torch.cuda.synchronize(); print(f"all_reduce_ranges ({n} floats): {1e3*(time.perf_counter()-t0):.1f} ms", flush=True)
        t_exit = time.perf_counter()
torch.cuda.synchronize(); print(f"exit estimate_ranges: {1e3*(time.perf_counter()-t_exit):.1f} ms", flush=True)
