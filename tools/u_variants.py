"""Interleaved A/B of FFQ_STREAM_U (chunks per lane) for quantize bf16->bf16 / bf16->i8 and dequantize i8->bf16."""
import os, sys, pathlib, statistics, torch
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from fastforward_amd import ops
shape=(14336,4096); n=shape[0]*shape[1]
ws=[(torch.randn(shape,device="cuda")*0.02).to(torch.bfloat16) for _ in range(6)]
scale=torch.rand(shape[0],device="cuda")*0.001+0.0005
tile=(1,shape[1])
qs=[ops.quantize_by_tile(w,scale,tile,8,torch.int8) for w in ws]
xs=[torch.randn(8,2048,4096,device="cuda",dtype=torch.bfloat16) for _ in range(4)]
s1,o1=torch.tensor([0.03],device="cuda"),torch.tensor([3.0],device="cuda")
cases={
 "quant  pc bf16->bf16 (4B)": (4, lambda r: ops.quantize_by_tile(ws[r%6],scale,tile,8,torch.bfloat16), n),
 "quant  pc bf16->i8   (3B)": (3, lambda r: ops.quantize_by_tile(ws[r%6],scale,tile,8,torch.int8), n),
 "dequant pc i8->bf16  (3B)": (3, lambda r: ops.dequantize_by_tile(qs[r%6],scale,tile,None,torch.bfloat16), n),
 "quant  pt bf16->i8   (3B)": (3, lambda r: ops.quantize_by_tile(xs[r%4],s1,xs[0].shape,8,torch.int8,o1), xs[0].numel()),
}
for name,(bpe,fn,numel) in cases.items():
    graphs={}
    for u in (1,2,4):
        os.environ["FFQ_STREAM_U"]=str(u)
        for r in range(2): fn(r)
        torch.cuda.synchronize()
        g=torch.cuda.CUDAGraph(); side=torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.cuda.graph(g, stream=side):
            for r in range(12): fn(r)
        torch.cuda.current_stream().wait_stream(side)
        graphs[u]=g
    times={u:[] for u in graphs}
    for rnd in range(10):
        for u,g in graphs.items():
            a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
            a.record(); g.replay(); b.record(); b.synchronize()
            times[u].append(a.elapsed_time(b)/12)
    print(name, "  ".join(f"U={u}: {statistics.median(t)*1e3:6.2f} us {numel*bpe/statistics.median(t)/1e6:5.0f} GB/s" for u,t in times.items()))
