"""Time torch SDPA variants on the Llama-3-8B attention shape (B=8, S=2048, 32 q heads, 8 kv heads, D=128), causal."""
import time, torch, torch.nn.functional as F
dev="cuda"; b,s,hq,hk,d=8,2048,32,8,128
q=torch.randn(b,s,hq*d,device=dev,dtype=torch.bfloat16); k=torch.randn(b,s,hk*d,device=dev,dtype=torch.bfloat16); v=torch.randn(b,s,hk*d,device=dev,dtype=torch.bfloat16)
def t(fn,n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/n*1e3
flops=4*b*hq*s*s*d/2
def run(name, fn):
    try:
        ms=t(fn); print(f"{name:60s} {ms:.3f} ms  {flops/ms/1e9:.0f} TFLOP/s")
    except Exception as e:
        print(f"{name:60s} FAILED {type(e).__name__}: {str(e)[:100]}")
qv=q.view(b,s,hq,d).transpose(1,2); kv_=k.view(b,s,hk,d).transpose(1,2); vv=v.view(b,s,hk,d).transpose(1,2)
run("transposed views, enable_gqa", lambda: F.scaled_dot_product_attention(qv,kv_,vv,is_causal=True,enable_gqa=True))
qc,kc,vc=qv.contiguous(),kv_.contiguous(),vv.contiguous()
run("contiguous [B,H,S,D], enable_gqa", lambda: F.scaled_dot_product_attention(qc,kc,vc,is_causal=True,enable_gqa=True))
ke=kc.repeat_interleave(4,dim=1); ve=vc.repeat_interleave(4,dim=1)
run("contiguous, kv expanded to 32 heads", lambda: F.scaled_dot_product_attention(qc,ke,ve,is_causal=True))
print("preferred fa lib api:", hasattr(torch.backends.cuda,"preferred_rocm_fa_library"))
try:
    print("current:", torch.backends.cuda.preferred_rocm_fa_library())
    torch.backends.cuda.preferred_rocm_fa_library("ck")
    print("after set:", torch.backends.cuda.preferred_rocm_fa_library())
    run("ck: transposed views, enable_gqa", lambda: F.scaled_dot_product_attention(qv,kv_,vv,is_causal=True,enable_gqa=True))
    run("ck: contiguous, kv expanded", lambda: F.scaled_dot_product_attention(qc,ke,ve,is_causal=True))
except Exception as e:
    print("ck not available:", type(e).__name__, str(e)[:200])
from torch.nn.attention import sdpa_kernel, SDPBackend
for be in (SDPBackend.FLASH_ATTENTION, SDPBackend.EFFICIENT_ATTENTION, SDPBackend.MATH):
    with sdpa_kernel(be):
        run(f"backend {be.name}: transposed, gqa", lambda: F.scaled_dot_product_attention(qv,kv_,vv,is_causal=True,enable_gqa=True))
