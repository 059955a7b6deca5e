"""How much of an int8 GEMM's time depends on WHERE its operands are when it starts (Infinity Cache or HBM): the gate+up launch of
Llama-3-8B at 16384 tokens behind different predecessors, timed by HIP events around the launch itself (its two weight row-sum
launches included), sequences replayed from a hipGraph.
  A  x codes written (RMSNorm + quantize), gate/up re-quantized, GEMM                       (the forward's order)
  B  gate/up re-quantized, x codes written, GEMM
  C  A + one read pass over the x codes right before the GEMM
  D  A with the weights NOT re-quantized in between (codes of an earlier launch: cold)       + 1 GB of unrelated traffic before
usage: python tools/mall_probe.py"""
import pathlib, statistics, sys
import torch
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from fastforward_amd import ops

dev = "cuda"
T, K, N = 16384, 4096, 14336
torch.manual_seed(0)
hidden = torch.randn(T, K, device=dev, dtype=torch.bfloat16)
delta = torch.randn(T, K, device=dev, dtype=torch.bfloat16)
gamma = torch.ones(K, device=dev, dtype=torch.bfloat16)
wg = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
wu = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
sg = wg.float().abs().amax(1) / 127
su = wu.float().abs().amax(1) / 127
sx, ox = torch.tensor([0.03], device=dev), torch.tensor([3.0], device=dev)
so, oo = torch.tensor([0.02], device=dev), torch.tensor([-4.0], device=dev)
junk = torch.empty(1 << 30, dtype=torch.uint8, device=dev)


def producer():
    _, _, (codes,) = ops.add_rmsnorm_quantize(hidden, delta, gamma, 1e-5, [(sx, ox)])
    return codes


def weights():
    return ops.quantize_rows_batch([wg, wu], [sg, su], [None, None], 8)


def gemm(x, g, u, events):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    out = ops.mlp_gate_up_w8a8(x, g, u, sx, ox, sg, su, so, oo, 8)
    b.record()
    events.append((a, b))
    return out


def variant(name):
    ev = []
    g0, u0 = weights()
    for _ in range(6):
        if name == "A":
            x = producer(); g, u = weights(); gemm(x, g, u, ev)
        elif name == "B":
            g, u = weights(); x = producer(); gemm(x, g, u, ev)
        elif name == "C":
            x = producer(); g, u = weights(); ops.minmax_by_tile(x.view(torch.bfloat16), (T, K // 2)); gemm(x, g, u, ev)
        else:
            x = producer(); junk.fill_(1); gemm(x, g0, u0, ev)
    torch.cuda.synchronize()
    return statistics.median(a.elapsed_time(b) for a, b in ev[2:]) * 1e3


for rep in range(3):
    print("  ".join(f"{n}: {variant(n):7.1f} us" for n in "ABCD"), flush=True)
