"""A/B of FusedForward switches on one box (Llama-3-8B W8A8, B=8, S=2048, hipGraph replay): ms per forward."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fastforward_amd as ff
from fastforward_amd import llama
from bench_configs import timed_forward
dev = "cuda"
cfg = llama.LlamaConfig.llama3_8b()
model = llama.build_model(cfg, dev, torch.bfloat16, seed=1236)
llama.quantize_llama(model, w_bits=8, a_bits=8, quantized_dtype=torch.int8)
gen = torch.Generator(device=dev).manual_seed(4321)
batch = torch.randint(0, cfg.vocab_size, (8, 2048), device=dev, generator=gen)
llama.calibrate(model, [torch.randint(0, cfg.vocab_size, (8, 2048), device=dev, generator=gen) for _ in range(2)], fused=True)
variants = {"default": {}, "fuse_residual=True": {"fuse_residual": True}, "fuse_rowsums=True": {"fuse_rowsums": True}, "fuse_attention=False": {"fuse_attention": False}, "fuse_mlp=False": {"fuse_mlp": False}}
for rep in range(2):
    for name, kw in variants.items():
        f = llama.FusedForward(model, **kw)
        s = timed_forward(lambda: f(batch, logits=True), steps=10, warmup=2)
        print(f"{name:24s} {s*1e3:8.2f} ms  {16384/s:9.0f} tokens/s", flush=True)
