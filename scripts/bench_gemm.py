import sys, torch
sys.path.insert(0, '.')
from adv_grpo_amd import ops
def bench(M,N,K,iters=20):
    a=torch.randn(M,K,device='cuda').to(torch.bfloat16); w=torch.randn(N,K,device='cuda').to(torch.bfloat16)
    out=torch.empty(M,N,dtype=torch.bfloat16,device='cuda')
    for _ in range(3): ops.gemm(a,w,out=out)
    s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(iters): ops.gemm(a,w,out=out)
    e.record(); torch.cuda.synchronize()
    ms=s.elapsed_time(e)/iters
    # torch (hipBLASLt) for context only
    for _ in range(3): torch.matmul(a,w.T)
    torch.cuda.synchronize(); s.record()
    for _ in range(iters): torch.matmul(a,w.T)
    e.record(); torch.cuda.synchronize()
    ms2=s.elapsed_time(e)/iters
    print(f"M={M} N={N} K={K}: {ms*1e3:.1f} us  {2*M*N*K/ms/1e9:.1f} TFLOP/s   [hipblaslt {2*M*N*K/ms2/1e9:.1f}]")
for shp in [(4096,4096,4096),(8192,8192,8192),(16384,1536,1536),(16384,4608,1536),(16384,6144,1536),(16384,1536,6144),(3280,4608,1536),(3280,1536,1536)]:
    bench(*shp)
