import sys, os, torch, faulthandler
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_gpu_train import _setup
from adv_grpo_amd.mmdit_train import SD3TransformerLoRA
from adv_grpo_amd.model_configs import MMDiTConfig
cfg = MMDiTConfig()
W, lora, lat, t, ctx, pooled, g = _setup(cfg, 31, B=16, hw=64, Nt=205)
model = SD3TransformerLoRA(W, cfg, "cuda", lora_state=lora)
torch.cuda.synchronize(); print("built", flush=True)
v, saved = model.forward_train(lat.cuda(), t.cuda(), ctx.cuda(), pooled.cuda())
torch.cuda.synchronize(); print("fwd ok", flush=True)
dv = torch.randn(v.shape, generator=g).to(torch.bfloat16)
model.backward(saved, dv.cuda())
torch.cuda.synchronize(); print("bwd ok", flush=True)
