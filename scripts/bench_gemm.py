"""GEMM micro-benchmark per tile variant (ADVGRPO_GEMM_FORCE=<id> python scripts/bench_gemm.py)."""
import os, sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adv_grpo_amd import ops
def bench(M,N,K,iters=20, epi=False):
    a=torch.randn(M,K,device='cuda').to(torch.bfloat16); w=torch.randn(N,K,device='cuda').to(torch.bfloat16)
    out=torch.empty(M,N,dtype=torch.bfloat16,device='cuda')
    kw={}
    if epi:
        kw=dict(bias=torch.randn(N,device='cuda').to(torch.bfloat16), gate=torch.randn(16,N,device='cuda').to(torch.bfloat16), gate_rows=(M+15)//16, residual=out)
    for _ in range(3): ops.gemm(a,w,out=out,**kw)
    s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(iters): ops.gemm(a,w,out=out,**kw)
    e.record(); torch.cuda.synchronize()
    ms=s.elapsed_time(e)/iters
    return 2*M*N*K/ms/1e9
shapes=[(4096,4096,4096),(8192,8192,8192),(16384,1536,1536),(16384,4608,1536),(16384,6144,1536),(16384,1536,6144),(3280,4608,1536),(3280,1536,1536),(3280,6144,1536),(3280,1536,6144)]
print("variant", os.environ.get("ADVGRPO_GEMM_FORCE","auto"), " ".join(f"{bench(*s):7.0f}" for s in shapes), "| epi:", " ".join(f"{bench(*s,epi=True):7.0f}" for s in shapes[2:6]))
