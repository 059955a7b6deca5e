"""q's rotary embedding inside the attention launch against the stand-alone pass, one process: the headline forward
(llama.FusedForward, hipGraph replay, Llama-3-8B shapes, B=8, S=2048), interleaved rounds; the two settings give the same logits bits.
usage: python tools/rope_ab.py [steps=10]"""
import pathlib, sys, time
import torch
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import fastforward_amd as ff
from fastforward_amd import llama, distributed as ffd

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
cfg = llama.LlamaConfig.llama3_8b()
model = llama.build_model(cfg, "cuda", torch.bfloat16, seed=1236)
llama.quantize_llama(model, 8, 8, torch.int8)
gen = torch.Generator(device="cuda").manual_seed(4321)
batch = torch.randint(0, cfg.vocab_size, (8, 2048), device="cuda", generator=gen)
ffd.calibrate_sharded(model, [batch], disable_quantization=False, fused=True)


def graph_of(flag):
    llama.FUSE_Q_ROPE = flag
    fwd = llama.FusedForward(model)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fwd(batch)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            out = fwd(batch)
    torch.cuda.current_stream().wait_stream(side)
    return g, out


graphs = {flag: graph_of(flag) for flag in (True, False)}
for g, _ in graphs.values():
    g.replay()
torch.cuda.synchronize()
assert torch.equal(graphs[True][1], graphs[False][1]), "the two settings differ"
best = {True: 1e9, False: 1e9}
for rnd in range(3):
    for flag in (True, False):
        g = graphs[flag][0]
        g.replay(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            g.replay()
        torch.cuda.synchronize()
        best[flag] = min(best[flag], (time.perf_counter() - t0) / steps * 1e3)
for flag in (True, False):
    print(f"q rotated {'inside the attention launch' if flag else 'by the rotary kernel      '}: {best[flag]:8.3f} ms / forward = {8 * 2048 / best[flag]:7.1f} k tokens/s", flush=True)
