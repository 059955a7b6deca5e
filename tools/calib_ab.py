"""Calibration step time with each estimation shortcut switched off, in ONE process (boxes differ by several percent, runs of
one box do not): Llama-3-8B shapes, LAYERS decoder layers, batch 8 x 2048, RunningMinMax through distributed.calibrate_sharded.
usage: python tools/calib_ab.py [layers=8] [steps=6] [module]   ("module": the reference-shaped module graph instead of the fused forward)"""
import contextlib, pathlib, sys, time, dataclasses
import torch
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import fastforward_amd as ff
from fastforward_amd import llama, distributed as ffd
from fastforward_amd.quantization.affine._memo import RECENT

layers = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
FUSED = not (len(sys.argv) > 3 and sys.argv[3] == "module")
cfg = dataclasses.replace(llama.LlamaConfig.llama3_8b(), num_layers=layers)
model = llama.build_model(cfg, "cuda", torch.bfloat16, seed=1)
llama.quantize_llama(model, 8, 8, torch.int8)
gen = torch.Generator(device="cuda").manual_seed(3)
calib = [torch.randint(0, cfg.vocab_size, (8, 2048), device="cuda", generator=gen) for _ in range(steps)]


@contextlib.contextmanager
def patched(*names):
    saved = []
    def swap(obj, attr, new):
        saved.append((obj, attr, getattr(obj, attr)))
        setattr(obj, attr, new)
    if "no_weight_one_pass" in names:
        swap(ff.nn.LinearQuantizer, "update_range_and_quantize", lambda self, *a, **k: None)
    if "no_either_or" in names:
        swap(ff.ops, "mlp_gate_up_w8a8_estimating", lambda *a, **k: None)
    if "no_gated" in names:
        swap(ff.ops, "mlp_gate_up_w8a8_estimating", lambda *a, **k: None)
        swap(ff.ops, "linear_w8a8_gated", lambda *a, **k: None)
    if "no_undecided_siblings" in names:
        swap(RECENT, "earlier_for", lambda *a, **k: None)  # k / v / up quantize unconditionally
    if "no_product_extrema" in names:
        swap(RECENT, "remember_extrema", lambda data, pair: None)  # (also un-shares the siblings' reduction)
    try:
        yield
    finally:
        for obj, attr, old in reversed(saved):
            setattr(obj, attr, old)


def reset():
    # a timed pass starts from uninitialised ranges, like a first calibration (bench.py): estimators seeded from a quantizer's
    # existing fp32 range merge through torch.min / torch.max instead of the in-place kernels
    for _, q in ff.nn.named_quantizers(model):
        q.reset_parameters()


def run(*names):
    with patched(*names):
        reset()
        ffd.calibrate_sharded(model, calib[:1], disable_quantization=False, fused=FUSED)
        reset()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ffd.calibrate_sharded(model, calib, disable_quantization=False, fused=FUSED)
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


arms = [(), ("no_undecided_siblings",), ("no_either_or",), ("no_weight_one_pass",), ("no_gated",), ("no_product_extrema",), ("no_weight_one_pass", "no_gated")]
best = {a: 1e9 for a in arms}
for rnd in range(3):
    for a in arms:
        best[a] = min(best[a], run(*a))
for a in arms:
    print(f"{' + '.join(a) or 'all shortcuts':60s} {best[a]:8.2f} ms / step of {layers} layers = {best[a] / layers:6.3f} ms per layer", flush=True)

# host-bound or device-bound? kernel time of one calibration pass against its wall time
reset()
ffd.calibrate_sharded(model, calib[:1], disable_quantization=False, fused=FUSED)
reset()
torch.cuda.synchronize()
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA, torch.profiler.ProfilerActivity.CPU]) as prof:
    t0 = time.perf_counter()
    ffd.calibrate_sharded(model, calib, disable_quantization=False, fused=FUSED)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
dev = sum(e.device_time_total for e in prof.key_averages() if e.device_type == torch.autograd.DeviceType.CUDA) / 1e3
print(f"one pass of {steps} steps: wall {wall * 1e3:.1f} ms (under the profiler), kernel time {dev:.1f} ms", flush=True)
