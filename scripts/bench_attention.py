import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adv_grpo_amd import ops
def bench(B,H,S,iters=20):
    D=64
    qkv=torch.randn(B,S,3*H*D,device='cuda').to(torch.bfloat16)
    q,k,v=qkv[...,:H*D],qkv[...,H*D:2*H*D],qkv[...,2*H*D:]
    out=torch.empty(B,S,H*D,dtype=torch.bfloat16,device='cuda')
    for _ in range(3): ops.attention(q,k,v,H,out=out)
    s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(iters): ops.attention(q,k,v,H,out=out)
    e.record(); torch.cuda.synchronize()
    ms=s.elapsed_time(e)/iters
    fl=4*B*H*S*S*D
    print(f"B={B} H={H} S={S}: {ms*1e3:.1f} us  {fl/ms/1e9:.1f} TFLOP/s")
bench(16,24,1229); bench(16,24,1024); bench(8,12,1370); bench(16,24,4301)
