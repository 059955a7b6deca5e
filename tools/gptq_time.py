"""GPTQ on one Llama-3-8B-shaped linear ([4096, 4096], 4-bit per-channel): one-launch-per-block kernel vs the reference's column loop."""
import pathlib, sys, time
import torch
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import fastforward_amd as ff
from fastforward_amd.quantization.gptq import gptq

dev = "cuda"
torch.manual_seed(0)
n_out, n_in = (int(v) for v in (sys.argv[1:3] if len(sys.argv) > 2 else (4096, 4096)))
acts = [((torch.randn(4, 512, n_in, device=dev),), {}) for _ in range(2)]
for fused in (True, True, False, True, False):  # the first pass warms hipSOLVER / hipBLASLt up
    layer = torch.nn.Linear(n_in, n_out, bias=False, device=dev)
    ff.quantize_model(layer)
    layer.weight_quantizer = ff.nn.LinearQuantizer(4, granularity=ff.PerChannel(0), symmetric=False, device=dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.no_grad(), ff.strict_quantization(False):
        gptq(layer, acts, fused=fused)
    torch.cuda.synchronize()
    print(f"fused={fused}: {time.perf_counter() - t0:.3f} s for [{n_out}, {n_in}]")
