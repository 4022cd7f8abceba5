import sys, torch, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adv_grpo_amd import synthetic
from adv_grpo_amd.mmdit import SD3Transformer2DModel
from oracle.mmdit import MMDiTConfig   # config dataclass only
cfg = MMDiTConfig()
W = synthetic.mmdit_weights(cfg, 1234)
model = SD3Transformer2DModel(W, cfg, "cuda"); del W
B = 16
lat = torch.randn(B,16,64,64,device='cuda').to(torch.bfloat16); t = torch.full((B,), 500.0, device='cuda')
ctx = torch.randn(B,205,4096,device='cuda').to(torch.bfloat16); pooled = torch.randn(B,2048,device='cuda').to(torch.bfloat16)
for _ in range(2): model(lat,t,ctx,pooled)
torch.cuda.synchronize(); t0=time.time()
n=5
for _ in range(n): model(lat,t,ctx,pooled)
torch.cuda.synchronize(); dt=(time.time()-t0)/n
print(f"MMDiT fwd B={B}: {dt*1e3:.2f} ms  -> {2.219*B/dt/1e3:.3f} PFLOP/s effective ({2.219*B/dt/1e3/2.5*100:.1f}% of 2.5 PF)")
