"""Head-dim-64 attention forward at the MMDiT / DINOv2 shapes: min and median of R repeats of N back-to-back launches each (single 20-launch
bursts scatter by +-5 % on one box; scripts/ab.sh alternates two builds).  Usage: bench_attention.py [N [R]]"""
import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adv_grpo_amd import ops
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
R = int(sys.argv[2]) if len(sys.argv) > 2 else 7
def bench(B,H,S):
    D=64
    qkv=torch.randn(B,S,3*H*D,device='cuda').to(torch.bfloat16)
    q,k,v=qkv[...,:H*D],qkv[...,H*D:2*H*D],qkv[...,2*H*D:]
    out=torch.empty(B,S,H*D,dtype=torch.bfloat16,device='cuda')
    for _ in range(10): ops.attention(q,k,v,H,out=out)
    ts=[]
    for _ in range(R):
        s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); s.record()
        for _ in range(N): ops.attention(q,k,v,H,out=out)
        e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e)/N)
    ts.sort()
    ms=ts[0]
    fl=4*B*H*S*S*D
    print(f"B={B} H={H} S={S}: min {ms*1e3:.1f} us  median {ts[len(ts)//2]*1e3:.1f} us  {fl/ms/1e9:.1f} TFLOP/s (at min)")
bench(16,24,1229); bench(16,24,1024); bench(8,12,1370); bench(16,24,4301)
