"""Is the LoRA gradient of one micro-step bitwise repeatable?  (same model, same inputs, twice; side-stream wgrad on / off)"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from adv_grpo_amd import synthetic
from adv_grpo_amd.mmdit_train import SD3TransformerLoRA
from adv_grpo_amd.model_configs import MMDiTConfig
full = len(sys.argv) > 1 and sys.argv[1] == "full"
mcfg = MMDiTConfig() if full else MMDiTConfig(num_layers=2, num_heads=4, joint_attention_dim=256, pooled_projection_dim=128, pos_embed_max_size=96, dual_attention_layers=(0,))
with synthetic.on_device("cuda"):
    tr = SD3TransformerLoRA(synthetic.mmdit_weights(mcfg, 1234), mcfg, "cuda", seed=1)
tr.params[tr.params == 0] = 0.01      # B != 0 so that every adapter gradient is exercised
tr.refresh() if hasattr(tr, "refresh") else None
B, hw, Nt = (16, 64, 205) if full else (4, 32, 21)
g = torch.Generator(device="cuda").manual_seed(0)
lat = torch.randn(B, 16, hw, hw, device="cuda", generator=g).to(torch.bfloat16)
t = torch.full((B,), 913.0, device="cuda")
ctx = torch.randn(B, Nt, mcfg.joint_attention_dim, device="cuda", generator=g).to(torch.bfloat16)
pooled = torch.randn(B, mcfg.pooled_projection_dim, device="cuda", generator=g).to(torch.bfloat16)
dv = torch.randn(B, 16, hw, hw, device="cuda", generator=g).to(torch.bfloat16)
for overlap in (True, False):
    tr.overlap_wgrad = overlap
    res = []
    for rep in range(3):
        tr.grads.zero_()
        v, saved = tr.forward_train(lat, t, ctx, pooled)
        tr.backward(saved, dv)
        torch.cuda.synchronize()
        res.append((v.clone(), tr.grads.clone()))
    for i in (1, 2):
        nd = int((res[0][1] != res[i][1]).sum())
        print(f"overlap_wgrad={overlap} rep {i}: forward equal {torch.equal(res[0][0], res[i][0])}; gradient elements differing {nd} of {res[0][1].numel()}"
              f" max diff {(res[0][1] - res[i][1]).abs().max().item():.3e} (|g| max {res[0][1].abs().max().item():.3e})")
        if nd:
            idx = (res[0][1] != res[i][1]).nonzero().flatten()
            print("   first differing flat indices", idx[:8].tolist(), "last", idx[-3:].tolist())
