"""One weight-only forward of BASELINE config 2 (W8 per channel) or 4 (W4 group-128, packed nibbles) for a rocprofv3 kernel trace:
which GEMM kernels the decoder linears run. usage: rocprofv3 --kernel-trace --stats ... -- python3 tools/cfg_trace.py cfg2|cfg4 [batch]
The only vendor GEMM (Cijk_*) left in the trace is lm_head (float in the recipe, quick-start :145): one launch per forward."""
import pathlib, sys
import torch
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import fastforward_amd as ff
from fastforward_amd import llama

which = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 8
forwards = 3
dev = torch.device("cuda", 0)
config = llama.LlamaConfig.llama3_8b()
model = llama.build_model(config, dev, torch.bfloat16, seed=1234 + (1 if which == "cfg2" else 3))
if which == "cfg2":
    llama.quantize_llama(model, w_bits=8, a_bits=None, quantized_dtype=torch.int8)
    runner = llama.FusedProducersForward(model)
else:
    llama.quantize_llama(model, w_bits=4, a_bits=None, quantized_dtype=torch.int8, weight_granularity=ff.PerBlock(1, 128, 0))
    runner = None
gen = torch.Generator(device=dev).manual_seed(7)
llama.calibrate(model, [torch.randint(0, config.vocab_size, (1, 256), device=dev, generator=gen)])
if runner is None:
    runner = llama.FusedProducersForward(model, weight_storage="packed")
ids = torch.randint(0, config.vocab_size, (batch, 2048), device=dev, generator=gen)
for _ in range(forwards):
    out = runner(ids, logits=True)
torch.cuda.synchronize()
print(f"{which}: {forwards} forwards of B={batch}, S=2048 done; logits {tuple(out.shape)}")
