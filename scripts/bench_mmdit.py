import sys, torch, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adv_grpo_amd import synthetic
from adv_grpo_amd.mmdit import SD3Transformer2DModel
from adv_grpo_amd.model_configs import MMDiTConfig
# usage: bench_mmdit.py [medium|large] [latent side, default 64]
kind = sys.argv[1] if len(sys.argv) > 1 else "medium"
side = int(sys.argv[2]) if len(sys.argv) > 2 else 64
cfg = MMDiTConfig() if kind == "medium" else MMDiTConfig(num_layers=38, num_heads=38, pos_embed_max_size=192, dual_attention_layers=())
W = synthetic.mmdit_weights(cfg, 1234)
model = SD3Transformer2DModel(W, cfg, "cuda"); del W
B = 16 if side <= 64 else 8
lat = torch.randn(B,16,side,side,device='cuda').to(torch.bfloat16); t = torch.full((B,), 500.0, device='cuda')
ctx = torch.randn(B,205,4096,device='cuda').to(torch.bfloat16); pooled = torch.randn(B,2048,device='cuda').to(torch.bfloat16)
for _ in range(2): model(lat,t,ctx,pooled)
torch.cuda.synchronize(); t0=time.time()
n=5
for _ in range(n): model(lat,t,ctx,pooled)
torch.cuda.synchronize(); dt=(time.time()-t0)/n
D, L, L2, Nt, Ni = cfg.dim, cfg.num_layers, len(cfg.dual_attention_layers), 205, (side // 2) ** 2
S = Ni + Nt
tf = (2 * (12 * D * D * S * L - 9 * D * D * Nt + 4 * D * D * Ni * L2) + 4 * D * (S * S * L + Ni * Ni * L2)) / 1e12
print(f"MMDiT {kind} {side * 8}^2 fwd B={B}: {dt*1e3:.2f} ms  ({tf:.2f} TFLOP/sample) -> {tf*B/dt/1e3:.3f} PFLOP/s effective ({tf*B/dt/1e3/2.5*100:.1f}% of 2.5 PF)")
