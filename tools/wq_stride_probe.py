"""Does the row pitch of the operands matter? The bf16-image weight-only GEMM at T tokens on [N, K] for several K around 14336 and 4096
(us per 64-deep super-step per tile round): a pitch that is a multiple of 4 KiB may alias every row of a tile onto few L2 channels.
usage: python tools/wq_stride_probe.py [T]"""
import pathlib, sys
import torch
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from fastforward_amd import ops
from bench import event_time_ms

T = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
dev = "cuda"
torch.manual_seed(0)
for n, ks in ((4096, (14336, 14400, 14464, 14592, 12288, 16384)), (4096, (4096, 4160, 4224, 4352)), (14336, (4096, 4160, 4224))):
    for k in ks:
        x = torch.randn(T, k, device=dev, dtype=torch.bfloat16)
        codes = torch.randint(-127, 128, (n, k), device=dev, dtype=torch.int8)
        s = torch.rand(n, device=dev) * 1e-3 + 5e-4
        w = ops.dequantize_by_tile(codes, s, (1, k), None, torch.bfloat16)
        ms = event_time_ms(lambda r: ops.linear_wq(x, codes, s, None, two_pass=True), iters=5, reps=4)
        mv = event_time_ms(lambda r: torch.nn.functional.linear(x, w), iters=5, reps=4)
        tiles = -(-T // 256) * -(-n // 256)
        rounds = -(-tiles // 256)
        steps = k // 64
        f = 2.0 * T * n * k
        print(f"N={n:5d} K={k:5d} (pitch {2 * k:6d} B = {2 * k / 4096:.3f} x 4 KiB): ours {ms * 1e3:8.1f} us {f / ms / 1e9:6.0f} TF, {ms * 1e3 / (rounds * steps):.3f} us/step | vendor {mv * 1e3:8.1f} us {f / mv / 1e9:6.0f} TF", flush=True)
        del x, codes, w
