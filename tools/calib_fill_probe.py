"""Where do the at::fill_ launches of a calibration step come from? torch.profiler with stacks on a 2-layer model."""
import pathlib, sys, dataclasses, collections
import torch
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import fastforward_amd as ff
from fastforward_amd import llama, distributed as ffd

cfg = dataclasses.replace(llama.LlamaConfig.llama3_8b(), num_layers=2)
model = llama.build_model(cfg, "cuda", torch.bfloat16, seed=1)
llama.quantize_llama(model, 8, 8, torch.int8)
calib = [torch.randint(0, cfg.vocab_size, (2, 512), device="cuda") for _ in range(3)]
ffd.calibrate_sharded(model, calib[:1], disable_quantization=False, fused=True)
for _, q in ff.nn.named_quantizers(model):
    q.reset_parameters()
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU], with_stack=True) as prof:
    ffd.calibrate_sharded(model, calib, disable_quantization=False, fused=True)
count = collections.Counter()
for e in prof.events():
    if e.name in ("aten::fill_", "aten::zero_", "aten::zeros", "aten::full", "aten::copy_", "aten::clone", "aten::to", "aten::_to_copy", "aten::neg", "aten::any", "aten::cat"):
        frames = [f for f in (e.stack or []) if "fastforward_amd" in f or "tools/" in f][:3]
        count[(e.name, " <- ".join(frames))] += 1
for (name, where), n in sorted(count.items(), key=lambda kv: -kv[1])[:40]:
    print(n, name, where)
